#!/usr/bin/env python3
"""Timeline (HIP events) of the chunked H2D uploads on the side stream and of the chunked compute on the main stream, alone and
together: does either slow the other down?  python tools/upload_timeline.py [chunk]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
import bench
from sslam_amd.harness import StreamingSequence, chunk_bounds
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 167
n = 613
dev = torch.device("cuda", 0)
pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device=dev)
imgs, toks = bench.synth_sequence(n, 0, n, 480, 640, 28, dev, seed=1234)
pin = imgs.cpu().pin_memory()
side = torch.cuda.Stream()
devbuf = torch.empty_like(imgs)
bounds = chunk_bounds(n, chunk, 16)
E = lambda: torch.cuda.Event(enable_timing=True)


def run(copy, compute, late=False):
    seq = StreamingSequence(pipe, (1,))
    seq.reset(capacity=n)
    cur = torch.cuda.current_stream()
    t0 = E(); t0.record(cur)
    side.wait_stream(cur)
    cev, kev = [], []
    if copy:
        with torch.cuda.stream(side):
            for a, b in bounds:
                s, e = E(), E()
                s.record(side)
                devbuf[a:b].copy_(pin[a:b], non_blocking=True)
                e.record(side)
                cev.append((s, e))
    if compute:
        for i, (a, b) in enumerate(bounds):
            if copy and not late:
                cur.wait_event(cev[i][1])
            s, e = E(), E()
            s.record(cur)
            seq.push(toks[a:b], devbuf[a:b], images_ready=cev[i][1] if (copy and late) else None)
            e.record(cur)
            kev.append((s, e))
    cur.wait_stream(side)
    torch.cuda.synchronize()
    return ([(t0.elapsed_time(s), t0.elapsed_time(e)) for s, e in cev], [(t0.elapsed_time(s), t0.elapsed_time(e)) for s, e in kev])


from sslam_amd.harness import run_frames
for label, kw in (("copies only", (True, False)), ("compute only", (False, True)), ("both", (True, True)),
                  ("both, pixels awaited in front of A9 only", (True, True, True))):
    run(*kw); run(*kw)
    c, k = run(*kw)
    print(label)
    for i, (a, b) in enumerate(bounds):
        row = f"  chunk {i} [{a:3d},{b:3d})"
        if c:
            row += f"   copy {c[i][0]:7.3f} -> {c[i][1]:7.3f} ({c[i][1] - c[i][0]:6.3f} ms, {(b - a) * 921600 / (c[i][1] - c[i][0]) / 1e6:5.1f} GB/s)"
        if k:
            row += f"   compute {k[i][0]:7.3f} -> {k[i][1]:7.3f} ({k[i][1] - k[i][0]:6.3f} ms)"
        print(row, flush=True)

for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_frames(pipe, n, 480, 640, spacings=(1,), tokens=toks, pinned_source=pin, chunk=chunk)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"run_frames: host returns after {(t1 - t0) * 1e3:7.3f} ms, GPU done after {(t2 - t0) * 1e3:7.3f} ms")

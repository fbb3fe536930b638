#!/usr/bin/env python3
"""Host-resident frames -> matches (sslam_amd.harness.run_frames): frames/s against the chunk size of the overlapped H2D feed,
beside the resident pass and the raw H2D rate.  python tools/upload_sweep.py [frames]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
import bench
from sslam_amd.harness import run_frames
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 613
dev = torch.device("cuda", 0)
cfg = ExtractorConfig()
pipe = SequencePipeline(cfg, synth.selector_state(0), synth.refiner_state(0), device=dev)
imgs, toks = bench.synth_sequence(n, 0, n, 480, 640, 28, dev, seed=1234)
pin = imgs.cpu().pin_memory()


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: pipe.run(imgs, toks))
print(f"resident pass: {t * 1e3:7.3f} ms = {n / t:9.1f} frames/s", flush=True)
probe = torch.empty_like(imgs)
t = timed(lambda: probe.copy_(pin, non_blocking=True))
print(f"raw H2D:       {t * 1e3:7.3f} ms = {pin.numel() / t / 1e9:6.1f} GB/s = {n / t:9.1f} frames/s", flush=True)
for kw, label in (({}, "whole-sequence device buffer"), ({"max_bytes": 0, "ring": 3}, "ring of 3 chunk slots")):
    for chunk in (83, 128, 167, 205, 256, 307, 613):
        for pre in (False, True):
            t = timed(lambda: run_frames(pipe, n, 480, 640, spacings=(1,), tokens=toks, pinned_source=pin, chunk=chunk, preprocess_too=pre, feeder_kw=kw), reps=8)
            print(f"{label}: chunk {chunk:4d} A0 {str(pre):>5s}: {t * 1e3:7.3f} ms = {n / t:9.1f} frames/s", flush=True)
# host-side cost alone: the same chunked pass over frames that are already resident (no feeder)
from sslam_amd.harness import StreamingSequence, chunk_bounds
for chunk in (49, 167):
    def chunked():
        seq = StreamingSequence(pipe, (1,))
        seq.reset(capacity=n)
        for a, b in chunk_bounds(n, chunk, 16):
            seq.push(toks[a:b], imgs[a:b])
    t = timed(chunked)
    print(f"resident, chunked {chunk:4d}: {t * 1e3:7.3f} ms = {n / t:9.1f} frames/s", flush=True)

# diagnosis: all H2D copies enqueued up front by the main thread (no feeder thread), compute follows chunk by chunk
side = torch.cuda.Stream()
devbuf = torch.empty_like(imgs)
for chunk in (49, 83, 167):
    bounds = chunk_bounds(n, chunk, 16)

    def upfront(compute=True, copy=True):
        seq = StreamingSequence(pipe, (1,))
        seq.reset(capacity=n)
        evs = []
        cur = torch.cuda.current_stream()
        if copy:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for a, b in bounds:
                    devbuf[a:b].copy_(pin[a:b], non_blocking=True)
                    e = torch.cuda.Event()
                    e.record(side)
                    evs.append(e)
        if compute:
            for i, (a, b) in enumerate(bounds):
                if copy:
                    cur.wait_event(evs[i])
                seq.push(toks[a:b], devbuf[a:b])
        if copy:
            cur.wait_stream(side)
    for label, kw in (("copies + compute", {}), ("copies only", {"compute": False}), ("compute only", {"copy": False})):
        t = timed(lambda: upfront(**kw))
        print(f"up-front, chunk {chunk:4d}, {label:17s}: {t * 1e3:7.3f} ms = {n / t:9.1f} frames/s", flush=True)

#!/usr/bin/env python3
"""Latency of the authored path at small batches (the drop-in scripts call it frame by frame): per-stage HIP-event times
and the wall time of extract(n frames) + match(n-1 pairs [+ 1 against the previous batch])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from sslam_amd import lib
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline

cfg = ExtractorConfig()
pipe = SequencePipeline(cfg, synth.selector_state(0), synth.refiner_state(0))
toks = torch.from_numpy(synth.token_sequence(17, 28)).cuda()
imgs = torch.from_numpy(synth.image_sequence(17)).cuda()
for n in (1, 2, 4, 8, 16):
    def step():
        ex = pipe.extract(toks[1:1 + n], imgs[1:1 + n])
        prev = pipe.extract(toks[:1], imgs[:1]) if n == 1 else None
        if n == 1:
            d = torch.cat([prev["descriptors"], ex["descriptors"]]); s = torch.cat([prev["scores"], ex["scores"]]); it = torch.cat([prev["intensity"], ex["intensity"]])
        else:
            d, s, it = ex["descriptors"], ex["scores"], ex["intensity"]
        return pipe.match(d, s, it)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 50
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # device-only time of one extract
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): pipe.extract(toks[1:1 + n], imgs[1:1 + n])
    b.record(); torch.cuda.synchronize()
    print(f"batch {n:2d}: step wall {dt*1e3:7.3f} ms ({n/dt:8.1f} frames/s)   extract device {a.elapsed_time(b)/20:7.3f} ms", flush=True)

#!/usr/bin/env python3
"""A7 (exact descriptor MLP, x_in entry) against the number of 32-row workgroups: whole rounds of 768 slots vs partial ones."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
from sslam_amd.pipeline import PackedRefiner

ref = PackedRefiner(synth.refiner_state(0), "cuda")
for wgs in (768, 1536, 768 * 12, 768 * 12 + 128, 768 * 12 + 256, 9578, 768 * 12 + 512, 768 * 13):
    rows = wgs * 32
    x = torch.randn(rows, 384, device="cuda")
    for _ in range(3):
        lib.refine(x, ref.packed, ref.n_blocks)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.refine(x, ref.packed, ref.n_blocks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print(f"workgroups {wgs:6d} ({wgs / 768:6.2f} rounds): {dt:7.3f} ms  ({rows * 1572864 / dt / 1e9:6.1f} TF)", flush=True)

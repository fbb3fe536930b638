#!/usr/bin/env python3
"""A6+A7 (exact gather + descriptor MLP) on the bench workload's shape (613 frames x 500 keypoints, 28 x 28 grid): time and
TFLOP/s of the fused entry and of the x_in entry.  python tools/refine_bench.py [path/to/libsslam_hip.so]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
if len(sys.argv) > 1:
    lib.SO_PATH = os.path.abspath(sys.argv[1])
from sslam_amd.pipeline import PackedRefiner

F, K, G = 613, 500, 28
ref = PackedRefiner(synth.refiner_state(0), "cuda")
torch.manual_seed(0)
feat = torch.randn(F, G, G, 384, device="cuda")
kp = torch.rand(F, K, 2, device="cuda") * (G - 1)
x = torch.randn(F * K, 384, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


d1 = lib.gather_refine(feat, kp, ref.packed, ref.n_blocks)
t1 = timed(lambda: lib.gather_refine(feat, kp, ref.packed, ref.n_blocks))
t2 = timed(lambda: lib.refine(x, ref.packed, ref.n_blocks))
fl = F * K * 1572864
print(f"{os.path.basename(lib.SO_PATH)}: gather+MLP {t1:6.3f} ms ({fl / t1 / 1e9:6.1f} TF)   MLP (x_in) {t2:6.3f} ms ({fl / t2 / 1e9:6.1f} TF)"
      f"   checksum {float(d1.double().sum()):.9f}", flush=True)

#!/usr/bin/env python3
"""A6+A7 in the bf16 throughput mode on the bench workload's shape (613 frames x 500 keypoints, 28 x 28 grid): time and
TFLOP/s of the fused and the x_in entry.  python tools/refine_bf16_bench.py [libsslam_hip.so variant] [frac]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
if len(sys.argv) > 1:                      # a variant build of the library (experiments)
    lib.SO_PATH = os.path.abspath(sys.argv[1])
from sslam_amd.pipeline import PackedRefiner

F, K, G = 613, 500, 28
ref = PackedRefiner(synth.refiner_state(0), "cuda", bf16=True)
torch.manual_seed(0)
feat = torch.randn(F, G, G, 384, device="cuda")
# integer cells, as the selector delivers them (grid_sample's coordinate round trip makes ~1/3 of them fractional by one ulp);
# argv[2] == "frac": uniformly fractional coordinates (all four taps of every keypoint have weight)
kp = torch.randint(0, G, (F, K, 2), device="cuda").float()
if len(sys.argv) > 2 and sys.argv[2] == "frac":
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


fl = F * K * 1572864
t1 = timed(lambda: lib.gather_refine_bf16(feat, kp, ref.packed_bf16, ref.n_blocks))
t2 = timed(lambda: lib.refine_bf16(x, ref.packed_bf16, ref.n_blocks))
print(f"{os.path.basename(lib.SO_PATH)}: gather+MLP {t1:6.3f} ms ({fl / t1 / 1e9:7.1f} TF = {fl / t1 / 1e9 / 2500:.3f} of the bf16 peak)   "
      f"MLP (x_in) {t2:6.3f} ms ({fl / t2 / 1e9:7.1f} TF)", flush=True)

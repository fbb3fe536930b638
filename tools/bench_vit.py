#!/usr/bin/env python3
"""Times the HIP ViT-S/16 forward alone (random weights): frames/s and effective TFLOP/s for several chunk sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from sslam_amd import lib
if os.environ.get("SSLAM_BENCH_SO"):            # a variant build of the library (experiments)
    lib.SO_PATH = os.path.abspath(os.environ["SSLAM_BENCH_SO"])
from sslam_amd.vit import DinoV3ViT
from sslam_amd.vit_hip import HipViT, HipViTF32

size = int(sys.argv[1]) if len(sys.argv) > 1 else 448
torch.manual_seed(0)
vit = DinoV3ViT().cuda().eval()
F32 = os.environ.get("SSLAM_BENCH_VIT", "bf16") == "fp32"       # the fp32-operand ViT (csrc/vit_f32.hip)
hv = HipViTF32(vit) if F32 else HipViT(vit)
T = 5 + (size // 16) ** 2
flop = 12 * (T * 384 * 1152 * 2 + 2 * 6 * T * T * 64 * 2 + T * 384 * 384 * 2 + 2 * T * 384 * 1536 * 2) + (T - 5) * 768 * 384 * 2
CH = int(os.environ.get("SSLAM_BENCH_CHUNK", "0"))      # frames per launch group (default: the whole batch in one group)
chunks = [int(a) for a in sys.argv[2:]] or [8, 16, 32, 64, 128]
for n in chunks:
    x = torch.randn(n, 3, size, size, device="cuda")
    for _ in range(2):
        hv.forward_features(x, chunk=CH or n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        hv.forward_features(x, chunk=CH or n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"size {size} chunk {n:4d}: {dt*1e3:8.3f} ms  {n/dt:9.1f} frames/s  {n*flop/dt/1e12:7.1f} TFLOP/s", flush=True)

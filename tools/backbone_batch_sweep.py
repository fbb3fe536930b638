#!/usr/bin/env python3
"""DinoBackbone.forward (the drop-in class) at the batch sizes the reference's callers use (B = 1 per frame; train.py: 4 / 8):
ms per call for vit_precision = fp32 (default, HIP fp32 kernels), bf16 (HIP), eager (the module's own torch forward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from sslam_amd import lib
if os.environ.get("SSLAM_BENCH_SO"):            # a variant build of the library (experiments)
    lib.SO_PATH = os.path.abspath(os.environ["SSLAM_BENCH_SO"])
from models.dino_backbone import DinoBackbone
from sslam_amd.vit import DinoV3ViT
torch.manual_seed(0)
vit = DinoV3ViT().cuda().eval()
bbs = {p: DinoBackbone(input_size=448, dino=vit, vit_precision=p).cuda().eval() for p in ("fp32", "bf16", "eager")}
for b in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]:
    x = torch.randn(b, 3, 448, 448, device="cuda")
    row = []
    for p, bb in bbs.items():
        with torch.no_grad():
            for _ in range(3):
                bb(x)
            torch.cuda.synchronize()
            reps = 20 if b <= 8 else 8
            t0 = time.perf_counter()
            for _ in range(reps):
                bb(x)
            torch.cuda.synchronize()
        row.append(f"{p} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms")
    print(f"B = {b:3d}: " + "   ".join(row), flush=True)

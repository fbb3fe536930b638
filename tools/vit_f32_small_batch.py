#!/usr/bin/env python3
"""HipViTF32.forward_features at the reference's own batch sizes (B = 1 per frame: visualize_matches_sequence.py:72-74; B = 4:
train.py:300-302): ms per call with the key-split attention (the default for <= 8 frames) and with the one-pass form
(SSLAM_VIT_F32_NO_KEY_SPLIT=1), and their token agreement.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from sslam_amd import lib
from sslam_amd.vit import DinoV3ViT
from sslam_amd.vit_hip import HipViTF32
torch.manual_seed(0)
vit = DinoV3ViT().cuda().eval()
hv = HipViTF32(vit)
for b in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 9, 16]:
    x = torch.randn(b, 3, 448, 448, device="cuda")
    res, toks = [], []
    for off in ((0,) if os.environ.get("SSLAM_SMALL_ONLY_DEFAULT") else (0, 1)):
        with lib.knobs(SSLAM_VIT_F32_NO_KEY_SPLIT=off), torch.no_grad():
            for _ in range(3):
                t = hv.forward_features(x)
            torch.cuda.synchronize()
            reps = 30
            t0 = time.perf_counter()
            for _ in range(reps):
                hv.forward_features(x)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / reps * 1e3)
            toks.append(t.clone())
    if len(res) == 1:
        print(f"B = {b:3d}: default {res[0]:7.3f} ms", flush=True)
        continue
    rel = float((toks[0] - toks[1]).norm() / toks[1].norm())
    print(f"B = {b:3d}: default {res[0]:7.3f} ms   one-pass attention {res[1]:7.3f} ms   tokens rel {rel:.2e}", flush=True)

#!/usr/bin/env python3
"""Does running two ViT launch groups on two streams (each kernel's ramp and tail filled by the other group's kernels) beat one
stream?  python tools/vit_streams.py [frames_per_group] [groups]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from sslam_amd import lib
from sslam_amd.vit import DinoV3ViT
from sslam_amd.vit_hip import HipViT

per = int(sys.argv[1]) if len(sys.argv) > 1 else 82
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
hv = HipViT(DinoV3ViT().cuda().eval())
NS = 4
imgs = [torch.randn(per, 3, 448, 448, device="cuda") for _ in range(NS)]
outs = [torch.empty(per, 789, 384, device="cuda") for _ in range(NS)]
ws = [torch.empty(lib.vit_workspace_bytes(per, 448), dtype=torch.uint8, device="cuda") for _ in range(NS)]
hv.forward_features(imgs[0], chunk=per)          # rope tables, warm-up
streams = [torch.cuda.Stream() for _ in range(NS)]


def run(n_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for g in range(groups):
        s = g % n_streams
        with torch.cuda.stream(streams[s]):
            lib.vit_forward(imgs[s], hv.w, ws[s], out=outs[s])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for n_streams in (1, 2, 3, 4, 1, 2, 3, 4):
    run(n_streams)
    dt = run(n_streams)
    print(f"{n_streams} stream(s): {groups} groups of {per} frames in {dt * 1e3:8.2f} ms = {groups * per / dt:9.1f} frames/s", flush=True)

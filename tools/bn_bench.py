#!/usr/bin/env python3
"""A2 (token BatchNorm, per-frame statistics) at the three BASELINE grids: time and algorithmic TB/s (tokens read once +
features written once) of the register-resident forms against the three-sweep kernel.  python tools/bn_bench.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from sslam_amd import lib
if os.environ.get("SSLAM_BENCH_SO"):            # a variant build of the library (experiments)
    lib.SO_PATH = os.path.abspath(os.environ["SSLAM_BENCH_SO"])

dev = "cuda"
ones, zeros = torch.ones(384, device=dev), torch.zeros(384, device=dev)
for name, g, n in (("fr1_desk_613 (G = 28)", 28, 613), ("fr2_desk_1024kp (G = 40)", 40, 512), ("synthetic_2048kp (G = 60)", 60, 128)):
    tok = torch.randn(n, 5 + g * g, 384, device=dev)
    out = torch.empty(n, g * g, 384, device=dev)
    row = []
    for form, label in ((0, "register-resident"), (1, "three sweeps")):
        with lib.knobs(SSLAM_BN_FORM=form):
            for _ in range(3):
                lib.bn_tokens(tok, 5, 1, ones, zeros, zeros, ones, True, 1e-5, out=out, want_stats=False)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            reps = 20
            for _ in range(reps):
                lib.bn_tokens(tok, 5, 1, ones, zeros, zeros, ones, True, 1e-5, out=out, want_stats=False)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
        byts = 2 * n * g * g * 384 * 4
        row.append(f"{label}: {ms:6.3f} ms = {byts / ms / 1e9:5.2f} TB/s")
    print(f"{name:28s} {n:4d} frames   " + "   ".join(row), flush=True)

#!/usr/bin/env python3
"""A3 (exact, throughput form) against the number of 128-cell tiles: whole rounds of 512 workgroup slots vs partial ones, with and
without the 32-cell tail tiles (SSLAM_CONV_TAIL).  G = 16 so that a frame is exactly two tiles.  python tools/conv_rounds.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
from sslam_amd.pipeline import PackedSelector

lib.lib().sslam_test_set_knob(b"SSLAM_CONV_LATENCY_ROWS", 0, 0)
sel = PackedSelector(synth.selector_state(0), "cuda")
for tiles in (512, 1024, 3072, 3584, 3584 + 86, 3584 + 170, 3584 + 256, 3584 + 342, 4096):
    n = tiles // 2
    feat = torch.randn(n, 16, 16, 384, device="cuda")
    out = torch.empty(n, 16, 16, device="cuda")
    row = []
    for tail in ("0", "512"):
        lib.lib().sslam_test_set_knob(b"SSLAM_CONV_TAIL", int(tail), 0)
        for _ in range(3):
            lib.selector_saliency(feat, sel.w1p, sel.b1, sel.w2, sel.b2, sel.hidden, out=out)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.selector_saliency(feat, sel.w1p, sel.b1, sel.w2, sel.b2, sel.hidden, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        row.append(f"tail={tail:>3s}: {dt:7.3f} ms ({tiles * 128 * 256 * 3456 * 2 / dt / 1e9:6.1f} TF)")
    print(f"tiles {tiles:5d} ({tiles / 512:5.2f} rounds): " + "   ".join(row), flush=True)

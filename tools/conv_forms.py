#!/usr/bin/env python3
"""A3 (exact conv) time per launch shape and frame count: where the latency forms stop paying.  python tools/conv_forms.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
from sslam_amd.pipeline import PackedSelector

sel = PackedSelector(synth.selector_state(0), "cuda")
FORMS = {"lat2": {"SSLAM_CONV_LATENCY_ROWS": 1 << 30, "SSLAM_CONV_LAT2_ROWS": 1 << 30},
         "lat1": {"SSLAM_CONV_LATENCY_ROWS": 1 << 30, "SSLAM_CONV_LAT2_ROWS": 0},
         "thr": {"SSLAM_CONV_LATENCY_ROWS": 0, "SSLAM_CONV_LAT2_ROWS": 0}}
for n in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 256, 400):
    feat = torch.randn(n, 28, 28, 384, device="cuda")
    out = torch.empty(n, 28, 28, device="cuda")
    row = []
    for name, env in FORMS.items():
        with lib.knobs(**env):
            for _ in range(3):
                lib.selector_saliency(feat, sel.w1p, sel.b1, sel.w2, sel.b2, sel.hidden, out=out)
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.selector_saliency(feat, sel.w1p, sel.b1, sel.w2, sel.b2, sel.hidden, out=out)
            torch.cuda.synchronize()
        row.append(f"{name} {(time.perf_counter() - t0) / reps * 1e3:7.3f} ms")
    print(f"frames {n:4d}: " + "   ".join(row), flush=True)

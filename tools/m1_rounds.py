#!/usr/bin/env python3
"""M1 against the number of pairs around the bench's 612 (K = 500: 4 query blocks per pair, 512 workgroup slots per round): what a
persistent (pair, query-block) loop could remove is the partial last round - 612 pairs are 4.78 rounds.  Also K = 512 (no padded
queries / candidates) at the same pair count: what exact 4 x 125 blocking could remove.  python tools/m1_rounds.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from sslam_amd import lib


def timed(K, pairs):
    d = torch.nn.functional.normalize(torch.randn(pairs + 1, K, 128, device="cuda"), dim=-1)
    ws = torch.empty(lib.workspace_bytes(1, 28, K, pairs), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        lib.sim_argmax(d[:-1], K * 128, K, d[1:], K * 128, K, pairs, workspace=ws)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        lib.sim_argmax(d[:-1], K * 128, K, d[1:], K * 128, K, pairs, workspace=ws)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 30


for K in (500, 512):
    for pairs in (512, 576, 612, 640, 704, 768):
        ms = timed(K, pairs)
        wgs = pairs * ((K + 127) // 128)
        print(f"K {K}  pairs {pairs:4d}  workgroups {wgs:5d} = {wgs / 512:5.2f} rounds   {ms:6.3f} ms   {ms / pairs * 1e3:6.3f} us per pair   "
              f"{pairs * K * K * 128 * 2 / ms / 1e9:6.1f} TFLOP/s", flush=True)

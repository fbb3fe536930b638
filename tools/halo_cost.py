#!/usr/bin/env python3
"""What extracting the boundary frame as its own launch group costs a rank (the 'early halo' of shard.py): one extraction of
613 frames against 1 + 612 frames, tokens in, on one GPU.  python tools/halo_cost.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline

dev = torch.device("cuda")
pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device=dev)
n = 613
toks = torch.randn(n, 789, 384, device=dev) * 3 + 0.5
imgs = torch.randint(0, 255, (n, 480, 640, 3), device=dev, dtype=torch.uint8)
out = pipe.alloc_extract(n, True)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def whole():
    pipe.extract(toks, imgs, out=out)


def split():
    pipe.extract(toks[:1], imgs[:1], out={k: v[:1] for k, v in out.items()})
    pipe.extract(toks[1:], imgs[1:], out={k: v[1:] for k, v in out.items()})


t1, t2 = timed(whole), timed(split)
print(f"extract 613 frames: {t1:.3f} ms;  1 + 612 frames: {t2:.3f} ms  (+{t2 - t1:.3f} ms)")

#!/usr/bin/env python3
"""Does an HBM-bound stage hide under the MFMA-bound conv when it runs on a second stream?  613 frames: A0 (preprocess) serial in front
of A2 + A3, against A0 on a side stream beside them; likewise A2 of a second half beside A3 of the first."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, synth
from sslam_amd import lib
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
dev = torch.device("cuda", 0)
n = 613
pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device=dev)
imgs = torch.from_numpy(synth.image_sequence(8)).to(dev).repeat(77, 1, 1, 1)[:n].contiguous()
toks = torch.from_numpy(synth.token_sequence(8, 28)).to(dev).repeat(77, 1, 1)[:n].contiguous()
s = pipe.selector
side = torch.cuda.Stream(dev)
sal = torch.empty((n, 28, 28), device=dev)
ws = pipe.workspace(n, 0)

def serial():
    v = pipe.preprocess(imgs)
    f = pipe.features(toks)
    lib.selector_saliency(f, s.w1p, s.b1, s.w2, s.b2, s.hidden, out=sal, workspace=ws)

def beside():
    cur = torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    f = pipe.features(toks)
    with torch.cuda.stream(side):
        v = pipe.preprocess(imgs)
        v.record_stream(side)
    lib.selector_saliency(f, s.w1p, s.b1, s.w2, s.b2, s.hidden, out=sal, workspace=ws)
    cur.wait_stream(side)

def conv_only():
    f = pipe.features(toks)
    lib.selector_saliency(f, s.w1p, s.b1, s.w2, s.b2, s.hidden, out=sal, workspace=ws)

for name, fn in (("A0 ; A2 ; A3 serial", serial), ("A2 ; (A3 || A0 on a side stream)", beside), ("A2 ; A3 alone", conv_only), ("A0 alone", lambda: pipe.preprocess(imgs))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms")

#!/usr/bin/env python3
"""Does the exact path of the first half of a sequence hide under the bf16 ViT of the second half (two streams)?  The bf16 ViT
leaves most of the matrix pipe's issue slots empty (0.28 of the spec peak); the exact stages are fp32-MFMA-bound.
613 frames: (a) serial: ViT(all) -> extract(all) -> match; (b) ViT(h1) -> [extract(h1) on a side stream || ViT(h2)] -> extract(h2) -> match.
Two pipeline objects so that the halves do not share stage buffers.  python tools/vit_overlap_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench, synth
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
from sslam_amd.vit import DinoV3ViT
dev = torch.device("cuda")
n, h, w, size, K = bench.WORKLOADS["fr1_desk_613"]
cfg = ExtractorConfig(input_size=size, num_keypoints=K)
ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
torch.manual_seed(0)
vit = DinoV3ViT().to(dev).eval()
pa = SequencePipeline(cfg, ssd, rsd, device=dev, vit=vit)
pb = SequencePipeline(cfg, ssd, rsd, device=dev)
imgs, _ = bench.synth_sequence(n, 0, n, h, w, size // 16, dev, seed=1234)
side = torch.cuda.Stream(dev)
half = 5 * 82          # whole ViT launch groups in the first part


def serial():
    return pa.run(imgs)


def overlapped():
    cur = torch.cuda.current_stream(dev)
    out = pa.alloc_extract(n, True)
    tok = torch.empty((n, 5 + cfg.grid ** 2, 384), dtype=torch.float32, device=dev)
    pa.tokens_from_images(imgs[:half], out=tok[:half])
    ev = cur.record_event()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        pb.extract(tok[:half], imgs[:half], out={k: v[:half] for k, v in out.items()})
    pa.tokens_from_images(imgs[half:], out=tok[half:])
    pa.extract(tok[half:], imgs[half:], out={k: v[half:] for k, v in out.items()})
    cur.wait_stream(side)
    out.update(pa.match(out["descriptors"], out["scores"], out["intensity"]))
    return out


res = {}
for name, fn in (("serial", serial), ("overlapped", overlapped)):
    for _ in range(2):
        o = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        o = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    res[name] = o
    print(f"{name:11s}: {dt * 1e3:7.2f} ms  {n / dt:8.1f} frames/s", flush=True)
print("equal:", all(torch.equal(res["serial"][k], res["overlapped"][k]) for k in ("idx", "descriptors", "matches", "match_count")))

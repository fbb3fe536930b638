#!/usr/bin/env python3
"""What a bf16 RESIDUAL STREAM would cost the ViT in accuracy, measured BEFORE building it.

The HIP ViT (csrc/vit.hip) keeps the residual stream in fp32; its one HBM-bound kernel is o_proj's fp32 read-modify-write of
that stream (57 us of a 469 us layer at 82 frames) and the fused MLP reads it twice.  Stored as bf16 the stream would take
~95 MB per layer and launch off those two kernels (-27 us per layer by their measured byte rates, i.e. +5-6 % ViT frames/s).
This script emulates the numerics in torch on the GPU, on the bench workload's frames, with the same random weights:

  fp32      the eager fp32 definition (the reference's numerics)
  bf16op    every GEMM operand rounded to bf16, fp32 accumulation, fp32 residual / LayerNorm / softmax  (= what vit.hip computes;
            its agreement with fp32 is also measured on the real kernels by bench.py: this row validates the emulation)
  bf16op+r  the same with the residual stream rounded to bf16 after the patch embedding and after every residual add

and reports, against fp32: relative error of the tokens, keypoint-set agreement per frame, match agreement as (cell, cell)
pairs - the bars of VERDICT r3 item 7 are >= 0.995 / 0.99.  python tools/vit_bf16_residual_emulation.py [frames]"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
import bench
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
from sslam_amd.vit import DinoV3ViT

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")
torch.manual_seed(0)
vit = DinoV3ViT().to(dev).eval()
pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device=dev)
imgs, _ = bench.synth_sequence(n, 0, n, 480, 640, 28, dev, seed=1234)


def rb(t):
    return t.to(torch.bfloat16).float()


def forward(x, bf16_ops: bool, bf16_res: bool):
    op = rb if bf16_ops else (lambda t: t)
    res = rb if bf16_res else (lambda t: t)
    B, _, H, W = x.shape
    gh, gw = H // 16, W // 16
    t = F.conv2d(op(x), op(vit.patch_embed.weight), vit.patch_embed.bias, stride=16).flatten(2).transpose(1, 2)
    t = res(torch.cat([vit.cls_token.expand(B, -1, -1), vit.register_tokens.expand(B, -1, -1), t], 1))
    cos, sin = vit.rope_tables(gh, gw, x.device)
    hd, nh = 64, 6
    for b in vit.blocks:
        h = op(b.norm1(t))
        q = F.linear(h, op(b.q_proj.weight), b.q_proj.bias).view(B, -1, nh, hd).transpose(1, 2)
        k = F.linear(h, op(b.k_proj.weight)).view(B, -1, nh, hd).transpose(1, 2)
        v = F.linear(h, op(b.v_proj.weight), b.v_proj.bias).view(B, -1, nh, hd).transpose(1, 2)

        def rope(u):
            pre, pat = u[:, :, :5], u[:, :, 5:]
            rot = torch.cat((-pat[..., hd // 2:], pat[..., :hd // 2]), -1)
            return torch.cat((pre, pat * cos + rot * sin), 2)
        q, k, v = op(rope(q)), op(rope(k)), op(v)
        att = torch.softmax((q @ k.transpose(2, 3)) * hd ** -0.5, -1)
        o = (op(att) @ v).transpose(1, 2).reshape(B, -1, 384)
        t = res(t + F.linear(op(o), op(b.o_proj.weight), b.o_proj.bias) * b.ls1)
        hid = F.gelu(F.linear(op(b.norm2(t)), op(b.up_proj.weight), b.up_proj.bias))
        t = res(t + F.linear(op(hid), op(b.down_proj.weight), b.down_proj.bias) * b.ls2)
    return vit.norm(t)


def tokens(mode):
    out = torch.empty((n, 789, 384), device=dev)
    with torch.no_grad():
        for a in range(0, n, 16):
            out[a:a + 16] = forward(pipe.preprocess(imgs[a:a + 16]), mode != "fp32", mode == "bf16op+r")
    return out


def cells(o, p):
    c = int(o["match_count"][p])
    m = o["matches"][p, :c].cpu().numpy()
    i1, i2 = o["idx"][p].cpu().numpy(), o["idx"][p + 1].cpu().numpy()
    return set(zip(i1[m[:, 0]].tolist(), i2[m[:, 1]].tolist()))


ref_t = tokens("fp32")
ref = {k: v.clone() for k, v in pipe.run(imgs, ref_t).items()}
print(f"{n} frames of the bench workload (640 x 480, G = 28, K = 500), random DINOv3-architecture weights, seed 0")
for mode in ("bf16op", "bf16op+r"):
    t = tokens(mode)
    o = pipe.run(imgs, t)
    rel = float((t - ref_t).norm() / ref_t.norm())
    kp = float(np.mean([np.intersect1d(a, b).size / np.unique(b).size for a, b in zip(o["idx"].cpu().numpy(), ref["idx"].cpu().numpy())]))
    hit = tot = 0
    for p in range(n - 1):
        a, b = cells(o, p), cells(ref, p)
        hit += len(a & b)
        tot += len(b)
    print(f"{mode:9s} vs fp32: tokens rel err {rel:.3e}   keypoint-set agreement {kp:.4f}   match agreement {hit / max(tot, 1):.4f}")

"""Checker of the bf16 THROUGHPUT mode (BASELINE.json configs[1] "bf16 conv stack").

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: imported by tests/ (and nothing else), never by the product package.

The mode's definition = the reference's algorithm (saliency CNN: semantic-slam/models/keypoint_selector.py:45-67; descriptor
MLP: semantic-slam/models/descriptor_refiner.py:58-126) with every GEMM operand rounded to bf16 (round-to-nearest-even) and
the products accumulated exactly (float64 here, fp32 on the GPU): what separates a correct kernel from this restatement is
fp32 accumulation-order noise only.  It is pinned against the exact oracle / the reference's own golden outputs by
tests/test_oracle_golden.py::test_bf16_mode_checker_tracks_the_exact_oracle (a drift bound: bf16 operands move the saliency
by < 3e-2 and keep descriptor cosines > 0.999) - the bf16 mode is not bit-comparable to the fp32 reference by construction.
"""
from __future__ import annotations

import numpy as np


def bf16_round(a):
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def saliency_bf16_ref(feat, sd):
    """conv3x3 (bf16 operands, float64 accumulate) + ReLU + conv1x1 + sigmoid (keypoint_selector.py:45-67)."""
    n, g, _, c = feat.shape
    x = np.zeros((n, g + 2, g + 2, c), np.float64)
    x[:, 1:-1, 1:-1] = bf16_round(feat)
    w = bf16_round(sd["conv.0.weight"]).astype(np.float64)          # (hs, c, 3, 3)
    hid = np.zeros((n, g, g, w.shape[0]), np.float64) + sd["conv.0.bias"].astype(np.float64)
    for ky in range(3):
        for kx in range(3):
            hid += x[:, ky:ky + g, kx:kx + g] @ w[:, :, ky, kx].T
    hid = np.maximum(hid.astype(np.float32), 0).astype(np.float64)
    logit = hid @ sd["conv.2.weight"].reshape(-1).astype(np.float64) + float(sd["conv.2.bias"].reshape(-1)[0])
    return (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)


def refine_bf16_ref(x, sd, n_blocks=2):
    """The bf16 kernel's formulation in float64: bf16 GEMM operands, LayerNorm folded into the next GEMM
    (descriptor_refiner.py:58-126 algebraically; refine_bf16.hip header)."""
    f8 = np.float64

    def lin(a, W, b):
        return bf16_round(a.astype(np.float32)).astype(f8) @ bf16_round(W).astype(f8).T + b.astype(f8)

    def ln_lin(a, gam, bet, W, b):
        a32 = a.astype(np.float32).astype(f8)
        mean = a32.mean(-1, keepdims=True)
        var = np.maximum((a32 * a32).mean(-1, keepdims=True) - mean * mean, 0)
        rstd = 1.0 / np.sqrt(var + 1e-5)
        wg = bf16_round((W * gam[None, :]).astype(np.float32)).astype(f8)
        c = b.astype(f8) + W.astype(f8) @ bet.astype(f8)
        return rstd * (bf16_round(a.astype(np.float32)).astype(f8) @ wg.T - mean * wg.sum(1)[None, :]) + c[None, :]

    X = np.maximum(lin(x, sd["input_proj.weight"], sd["input_proj.bias"]), 0)
    for i in range(n_blocks):
        p = f"residual_blocks.{i}."
        h = np.maximum(ln_lin(X, sd[p + "norm1.weight"], sd[p + "norm1.bias"], sd[p + "fc1.weight"], sd[p + "fc1.bias"]), 0)
        X = np.maximum(ln_lin(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "fc2.weight"], sd[p + "fc2.bias"]) + X, 0)
    o = lin(X, sd["output_proj.weight"], sd["output_proj.bias"])
    return (o / np.maximum(np.sqrt((o * o).sum(-1, keepdims=True)), 1e-12)).astype(np.float32)

"""Deterministic synthetic inputs shared by the golden-vector generator, the parity tests and bench.py.

Everything here is numpy-only (no torch, no reference code) so that the same bytes can be regenerated on the
GPU box, where /root/reference does not exist.  The recipes follow SURVEY.md §8(d):

* selector / refiner weights: same *distributions* as the reference's init rules
  (xavier-uniform gain 0.5 for the saliency CNN, keypoint_selector.py:38-43; unit-scale dense matrices and
  U(-0.1, 0.1) biases for the descriptor MLP, descriptor_refiner.py:47-56) drawn from a numpy PCG64 stream,
  with conv.2.weight scaled x8 so that saliency spans ~[0.1, 0.95];
* tokens: N(0.5, 3^2), seed 2000+i;
* images: clip(128 + 60*lowpass(N(0,1), sigma=12 px) + 25*N(0,1)), seed 1000+i.
"""
from __future__ import annotations

import numpy as np

C_FEAT = 384          # ViT-S/16 embed dim (dino_backbone.py:50)
N_PREFIX = 5          # CLS + 4 register tokens (dino_backbone.py:51,91)


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def _uniform(rng, shape, bound) -> np.ndarray:
    return ((rng.random(shape) * 2.0 - 1.0) * bound).astype(np.float32)


def _normal(rng, shape, std=1.0, mean=0.0) -> np.ndarray:
    return (rng.standard_normal(shape) * std + mean).astype(np.float32)


def selector_state(seed: int = 0, c_in: int = C_FEAT, hidden: int = 256, w2_scale: float = 8.0,
                   bias_scale: float = 0.02) -> dict:
    """state_dict of models.keypoint_selector.KeypointSelector (keys: SURVEY §8b1)."""
    rng = _rng(10_000 + seed)
    b1 = 0.5 * np.sqrt(6.0 / (c_in * 9 + hidden * 9))
    b2 = 0.5 * np.sqrt(6.0 / (hidden + 1))
    return {
        "conv.0.weight": _uniform(rng, (hidden, c_in, 3, 3), b1),
        "conv.0.bias": _uniform(rng, (hidden,), bias_scale),
        "conv.2.weight": _uniform(rng, (1, hidden, 1, 1), b2) * np.float32(w2_scale),
        "conv.2.bias": _uniform(rng, (1,), bias_scale),
    }


def refiner_state(seed: int = 0, c_in: int = C_FEAT, hidden: int = 384, d_out: int = 128,
                  n_blocks: int = 2) -> dict:
    """state_dict of models.descriptor_refiner.DescriptorRefiner (keys: SURVEY §8b1)."""
    rng = _rng(20_000 + seed)
    sd = {
        "input_proj.weight": _normal(rng, (hidden, c_in), 1.0 / np.sqrt(c_in)),
        "input_proj.bias": _uniform(rng, (hidden,), 0.1),
    }
    for i in range(n_blocks):
        for nm in ("norm1", "norm2"):
            sd[f"residual_blocks.{i}.{nm}.weight"] = (1.0 + _uniform(rng, (hidden,), 0.1)).astype(np.float32)
            sd[f"residual_blocks.{i}.{nm}.bias"] = _uniform(rng, (hidden,), 0.05)
        for nm in ("fc1", "fc2"):
            sd[f"residual_blocks.{i}.{nm}.weight"] = _normal(rng, (hidden, hidden), 1.0 / np.sqrt(hidden))
            sd[f"residual_blocks.{i}.{nm}.bias"] = _uniform(rng, (hidden,), 0.1)
    sd["output_proj.weight"] = _normal(rng, (d_out, hidden), 1.0 / np.sqrt(hidden))
    sd["output_proj.bias"] = _uniform(rng, (d_out,), 0.1)
    return sd


def tokens(frame: int, grid: int = 28, batch: int = 1) -> np.ndarray:
    """(batch, 5 + grid*grid, 384) fp32 stand-in for timm's forward_features output (dino_backbone.py:85)."""
    rng = _rng(2000 + frame)
    return _normal(rng, (batch, N_PREFIX + grid * grid, C_FEAT), 3.0, 0.5)


def token_sequence(n_frames: int, grid: int = 28, seed: int = 7, noise: float = 0.35) -> np.ndarray:
    """(n_frames, 5+grid^2, 384): windows sliding over one larger random field plus per-frame noise, so that
    consecutive frames share most of their content and mutual matches exist."""
    rng = _rng(30_000 + seed)
    pad = 6
    field = _normal(rng, (grid + 2 * pad, grid + 2 * pad, C_FEAT), 3.0, 0.5)
    out = np.empty((n_frames, N_PREFIX + grid * grid, C_FEAT), np.float32)
    ox = oy = pad
    for i in range(n_frames):
        step = rng.integers(-1, 2, size=2)
        ox = int(np.clip(ox + step[0], 0, 2 * pad))
        oy = int(np.clip(oy + step[1], 0, 2 * pad))
        win = field[oy:oy + grid, ox:ox + grid].reshape(grid * grid, C_FEAT)
        out[i, :N_PREFIX] = _normal(rng, (N_PREFIX, C_FEAT), 3.0, 0.5)
        out[i, N_PREFIX:] = win + _normal(rng, win.shape, noise)
    return out


def image(frame: int, height: int = 480, width: int = 640) -> np.ndarray:
    """(H, W, 3) uint8 smooth-plus-texture RGB frame (SURVEY §8d pixel recipe)."""
    from scipy.ndimage import gaussian_filter

    rng = _rng(1000 + frame)
    low = gaussian_filter(rng.standard_normal((height, width, 3)), sigma=(12, 12, 0), mode="wrap")
    low = low / (low.std() + 1e-9)
    img = 128.0 + 60.0 * low + 25.0 * rng.standard_normal((height, width, 3))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def image_sequence(n_frames: int, height: int = 480, width: int = 640, seed: int = 3) -> np.ndarray:
    """(n, H, W, 3) uint8: one base frame shifted by <= 4 px per step plus fresh texture noise."""
    rng = _rng(40_000 + seed)
    base = image(seed, height + 64, width + 64).astype(np.int16)
    out = np.empty((n_frames, height, width, 3), np.uint8)
    ox = oy = 32
    for i in range(n_frames):
        step = rng.integers(-4, 5, size=2)
        ox = int(np.clip(ox + step[0], 0, 64))
        oy = int(np.clip(oy + step[1], 0, 64))
        win = base[oy:oy + height, ox:ox + width] + rng.integers(-6, 7, size=(height, width, 3))
        out[i] = np.clip(win, 0, 255).astype(np.uint8)
    return out


def depth(frame: int, height: int = 480, width: int = 640) -> np.ndarray:
    """uint16 depth, U(2500, 25000) == 0.5-5 m at TUM scale 5000 (tum_dataset.py:139); loaded, never consumed."""
    return _rng(5000 + frame).integers(2500, 25000, size=(height, width), dtype=np.uint16)


def unit_descriptors(seed: int, n: int, d: int = 128, dup: int = 0) -> np.ndarray:
    """(n, d) fp32 unit-norm rows; the last `dup` rows repeat earlier rows bit-for-bit (SURVEY H2 duplicates)."""
    rng = _rng(50_000 + seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.sqrt((x.astype(np.float64) ** 2).sum(-1, keepdims=True)).astype(np.float32)
    if dup:
        src = rng.integers(0, n - dup, size=dup)
        x[n - dup:] = x[src]
    return x


def descriptor_pair(seed: int, n: int, m: int, dup: int, noise: float = 0.25):
    """Two descriptor sets with known correspondences and duplicated rows (== tests/golden/make_golden.py:pair)."""
    d1 = unit_descriptors(seed, n, 128, dup)
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    perm = (rng.permutation(max(n, m)) % n)[:m]
    d2 = d1[perm] + noise * rng.standard_normal((m, 128)).astype(np.float32) / np.sqrt(128).astype(np.float32)
    d2 = (d2 / np.linalg.norm(d2.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    if dup:
        d2[m - dup // 2:] = d2[: dup // 2]
    s1 = (0.2 + 0.8 * rng.random(n)).astype(np.float32)
    s2 = (0.2 + 0.8 * rng.random(m)).astype(np.float32)
    i1 = rng.random(n).astype(np.float32)
    i2 = rng.random(m).astype(np.float32)
    return d1, d2, s1, s2, i1, i2


def wide_map(seed: int):
    """Case `seed` of the wide selection fixture (tests/golden/select_wide.npz): a tie-free saliency map (uniform /
    a band hugging 0.5 / smooth bumps) with a random grid, K, NMS radius and percentile -> (map, K, radius, pct)."""
    rng = np.random.Generator(np.random.PCG64(60_000 + seed))
    g = int(rng.integers(6, 61))
    kind = seed % 3
    while True:
        u = rng.random((g, g))
        if kind == 0:
            m = u
        elif kind == 1:
            m = 0.5 + 0.02 * (u - 0.5)
        else:
            yy, xx = np.mgrid[0:g, 0:g] / g
            m = 0.5 + 0.4 * np.sin(6.0 * xx + seed) * np.cos(5.0 * yy - seed) + 0.05 * u
        m = m.astype(np.float32)
        if np.unique(m).size == m.size:
            break
    K = int(rng.integers(1, g * g + 1))
    radius = int(rng.integers(0, 4))
    pct = float(rng.choice([0.1, 0.25, 0.5, 0.6, 0.8]))
    return m, K, radius, pct


def wide_pair(seed: int):
    """Case `seed` of the wide matcher fixture (tests/golden/match_wide.npz)."""
    rng = np.random.Generator(np.random.PCG64(70_000 + seed))
    n, m = int(rng.integers(2, 600)), int(rng.integers(2, 600))
    dup = int(rng.integers(0, n // 4 + 1))
    d1, d2, s1, s2, i1, i2 = descriptor_pair(5000 + seed, n, m, min(dup, m // 2 * 2), noise=float(rng.choice([0.1, 0.25, 0.5])))
    kw = dict(saliency_weight=float(np.round(rng.random(), 3)), min_saliency=float(np.round(rng.random() * 0.6, 3)),
              min_descriptor_sim=float(np.round(0.3 + 0.6 * rng.random(), 3)), min_intensity=float(np.round(rng.random() * 0.4, 3)))
    if seed % 2:
        kw = dict(kw, intensity1=i1, intensity2=i2)
    return d1, d2, s1, s2, kw


def write_tum_sequence(root, n_rgb: int = 9, n_depth: int = 7, n_gt: int = 11) -> list:
    """A tiny TUM RGB-D sequence directory (rgb/, depth/, groundtruth.txt) written deterministically, with the traits
    that exercise data/tum_dataset.py:210-255: more rgb than depth files, several frames inside one whole second,
    comment / short lines in groundtruth.txt, un-normalised quaternions, ground-truth stamps not aligned with frames.
    Used by tests/golden/make_golden_tum.py (which runs the reference's TUMDataset on it) and by tests/test_tum_reader.py
    (which runs sslam_amd.tum.TUMSequence on the same bytes).  Returns the rgb file names in creation order."""
    import os

    from PIL import Image
    os.makedirs(os.path.join(root, "rgb"))
    os.makedirs(os.path.join(root, "depth"))
    rng = _rng(80_000)
    names = []
    for i in range(n_rgb):
        # creation order is deliberately NOT the sorted order
        k = (i * 4) % n_rgb
        nm = f"{1305031452 + k // 3}.{(791720 + 333000 * k) % 1000000:06d}.png"
        Image.fromarray(rng.integers(0, 255, (24, 32, 3), dtype=np.uint8)).save(os.path.join(root, "rgb", nm))
        names.append(nm)
    for i in range(n_depth):
        nm = f"{1305031452 + i // 3}.{(800000 + 333000 * i) % 1000000:06d}.png"
        Image.fromarray(rng.integers(2500, 25000, (24, 32)).astype(np.uint16)).save(os.path.join(root, "depth", nm))
    with open(os.path.join(root, "groundtruth.txt"), "w") as f:
        f.write("# ground truth trajectory\n# file: 'synthetic'\n# timestamp tx ty tz qx qy qz qw\nshort line 1 2\n\n")
        for k in range(n_gt):
            a = 0.13 * k
            q = np.array([0.1 * np.sin(a), -0.2 * np.cos(a), np.sin(a), np.cos(a)]) * (1.0 + 0.05 * k)   # not unit norm
            f.write(f"{1305031451.7 + 0.41 * k:.4f} {0.1 * k:.4f} {-0.05 * k:.4f} {1.0 + 0.01 * k:.4f} "
                    f"{q[0]:.6f} {q[1]:.6f} {q[2]:.6f} {q[3]:.6f}\n")
    return names


def write_tum_rgb_sequence(root, frames: np.ndarray, with_depth: bool = False) -> list:
    """A TUM-layout sequence directory whose rgb/ frames are `frames` ((n, H, W, 3) uint8), named by 30 Hz timestamps in
    the TUM style (files are written in a scrambled order: the reader must sort by name); groundtruth.txt with one pose per
    frame.  For the directory -> matches tests and `bench.py --tum-root` when no real sequence is on the box."""
    import os

    from PIL import Image
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    n = len(frames)
    names = [f"{1305031452 + (791720 + 33333 * i) // 1000000}.{(791720 + 33333 * i) % 1000000:06d}.png" for i in range(n)]
    assert names == sorted(names)
    for i in _rng(81_000).permutation(n):
        Image.fromarray(frames[i]).save(os.path.join(root, "rgb", names[i]), compress_level=1)
    if with_depth:
        os.makedirs(os.path.join(root, "depth"), exist_ok=True)
        for i in range(n):
            Image.fromarray(depth(i, frames.shape[1], frames.shape[2])).save(os.path.join(root, "depth", names[i]))
    with open(os.path.join(root, "groundtruth.txt"), "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for i in range(n):
            f.write(f"{1305031452.7917 + i / 30.0:.4f} {0.01 * i:.4f} 0.0000 1.0000 0.000000 0.000000 0.000000 1.000000\n")
    return names

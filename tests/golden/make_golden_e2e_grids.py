#!/usr/bin/env python3
"""End-to-end golden sequences at the larger grids of BASELINE configs[2] / configs[4], a keypoint-ORDER set at G = 60, and
the B = 4 BatchNorm case of train.py:300-302 - produced like make_golden.py by RUNNING the reference's own modules in the
authoring container (same import route: make_golden._import_reference).

  e2e_g40.npz     :  8 consecutive frames, G = 40 (input_size 640), K = 1024, 640 x 480 images; 7 spacing-1 pairs + (0,5)
  e2e_g60.npz     : 16 consecutive frames, G = 60 (input_size 960), K = 2048, 1280 x 960 images; 15 spacing-1 pairs + (0,5),(5,10)
  order_g60.npz   : 32 more G = 60 frames (independent token fields, synth.tokens(7000 + i, 60)): the reference's keypoint
                    order and scores only - the sample the keypoint-order swap rate against torch is measured on - plus,
                    per frame, how many positions of the reference's OWN keypoint list change when the same code runs with
                    one intra-op thread instead of four, or with oneDNN disabled (self_swaps_*)
  bn_tokens_b4.npz: BatchNorm1d over a batch of FOUR frames (train and eval mode), G = 28

The chain driven per frame (visualize_matches_sequence.py:69-104, B = 1, backbone NOT in eval mode - SURVEY H1):
  DinoBackbone.forward (token drop + BatchNorm1d + reshape, dino_backbone.py:91-106) -> KeypointSelector.forward (:45-67)
  -> select_keypoints (:69-207) -> extract_at_keypoints (dino_backbone.py:114-152) -> DescriptorRefiner.forward
  (descriptor_refiner.py:58-91) -> patch_to_pixel (:154-165) -> intensity lookup (Pillow default-filter resize -> "L",
  visualize_matches_sequence.py:87-95); per pair SequenceMatcher.match_with_quality with the CLI thresholds (:381-388).

Inputs are regenerated from seeds by the tests (tests/synth.py); the fixtures hold the reference's outputs only.
Keypoints are stored as flat cell indices (int16; the generator asserts kp == (idx % G, idx // G)), matches as int16.

Usage:  python tests/golden/make_golden_e2e_grids.py
"""
from __future__ import annotations

import numpy as np
import torch

import make_golden as mg
from make_golden import t
import synth

CLI = dict(saliency_weight=0.3, min_saliency=0.5, min_descriptor_sim=0.7, min_intensity=0.15)
SPECS = {
    # tag: (grid, K, frames, image h, w, extra pairs, descriptor subsample step)
    "e2e_g40": (40, 1024, 8, 480, 640, [(0, 5)], 8),
    "e2e_g60": (60, 2048, 16, 960, 1280, [(0, 5), (5, 10)], 16),
}
ORDER_FRAMES = 32
ORDER_SEED0 = 7000


def extract_frame(M, bb, sel, ref, tok, grid, K):
    bb.dino.tokens = t(tok[None])
    size = 16 * grid
    with torch.no_grad():
        f = bb(torch.zeros(1, 3, size, size))
        sal = sel(f)
        kp, sc = sel.select_keypoints(sal, num_keypoints=K)
        desc = ref(bb.extract_at_keypoints(f, kp))
        pix = bb.patch_to_pixel(kp)[0].numpy()
    kp = kp[0].numpy()
    idx = (kp[:, 1] * grid + kp[:, 0]).astype(np.int64)
    assert np.array_equal(kp, np.stack([idx % grid, idx // grid], 1).astype(np.float32))
    return dict(kp=kp, idx=idx, scores=sc[0].numpy(), desc=desc[0].numpy(), pix=pix, sal=sal[0, :, :, 0].numpy())


def gen_sequence(M, tag):
    from PIL import Image
    grid, K, n, h, w, extra, step = SPECS[tag]
    size = 16 * grid
    sel = mg.load_selector(M["sel"], 0, 256)
    ref = mg.load_refiner(M["ref"], 0)
    toks = synth.token_sequence(n, grid)
    imgs = synth.image_sequence(n, h, w)
    bb = mg.make_backbone(M["bb"], grid)
    bb.train(True)
    frames = []
    for i in range(n):
        f = extract_frame(M, bb, sel, ref, toks[i], grid, K)
        gray = np.asarray(Image.fromarray(imgs[i], "RGB").resize((size, size)).convert("L"), dtype=np.float32) / 255.0
        xs = np.clip(f["pix"][:, 0].round().astype(int), 0, size - 1)
        ys = np.clip(f["pix"][:, 1].round().astype(int), 0, size - 1)
        f["intensity"] = gray[ys, xs]
        frames.append(f)
    out = dict(grid=grid, K=K, n_frames=n, height=h, width=w, desc_step=step)
    for i, f in enumerate(frames):
        srt = np.sort(f["sal"].ravel())[::-1]
        out[f"f{i}_idx"] = f["idx"].astype(np.int16)
        out[f"f{i}_scores"] = f["scores"]
        out[f"f{i}_intensity"] = f["intensity"]
        out[f"f{i}_sal"] = f["sal"]
        out[f"f{i}_desc_sub"] = f["desc"][::step]
        out[f"f{i}_desc_sha"] = mg.sha(f["desc"])
        out[f"f{i}_min_gap"] = np.min(srt[:-1] - srt[1:])          # 0 where the reference's own saliencies tie exactly
        out[f"f{i}_n_ties"] = f["sal"].size - np.unique(f["sal"]).size
    mq = M["vms"].SequenceMatcher.match_with_quality
    pairs = [(i, i + 1) for i in range(n - 1)] + list(extra)
    for a, b in pairs:
        fa, fb = frames[a], frames[b]
        mt, q = mq(fa["desc"], fb["desc"], fa["scores"], fb["scores"], intensity1=fa["intensity"],
                   intensity2=fb["intensity"], **CLI)
        assert mt.dtype == np.int64 and q.dtype == np.float32
        out[f"pair_{a}_{b}_matches"] = mt.astype(np.int16)
        out[f"pair_{a}_{b}_quality"] = q
        out[f"pair_{a}_{b}_rowgap"], out[f"pair_{a}_{b}_colgap"] = mg.gaps(fa["desc"], fb["desc"])
    out["pairs"] = np.array(pairs, np.int32)
    mg.save(tag, **out)


def gen_order(M):
    grid, K = 60, 2048
    sel = mg.load_selector(M["sel"], 0, 256)
    ref = mg.load_refiner(M["ref"], 0)
    bb = mg.make_backbone(M["bb"], grid)
    bb.train(True)
    out = dict(grid=grid, K=K, count=ORDER_FRAMES, seed0=ORDER_SEED0)
    for i in range(ORDER_FRAMES):
        f = extract_frame(M, bb, sel, ref, synth.tokens(ORDER_SEED0 + i, grid)[0], grid, K)
        out[f"f{i}_idx"] = f["idx"].astype(np.int16)
        out[f"f{i}_scores"] = f["scores"]
        out[f"f{i}_n_ties"] = f["sal"].size - np.unique(f["sal"]).size
        # the reference against ITSELF under two settings that leave its algorithm untouched and only change the fp32
        # summation order inside torch's convolution: one intra-op thread instead of four; oneDNN disabled
        tok = synth.tokens(ORDER_SEED0 + i, grid)[0]
        torch.set_num_threads(1)
        f1 = extract_frame(M, bb, sel, ref, tok, grid, K)
        torch.set_num_threads(4)
        with torch.backends.mkldnn.flags(enabled=False):
            f2 = extract_frame(M, bb, sel, ref, tok, grid, K)
        for nm, o in (("1thread", f1), ("nomkldnn", f2)):
            assert np.array_equal(np.sort(o["idx"]), np.sort(f["idx"]))
            out[f"f{i}_self_swaps_{nm}"] = int((o["idx"] != f["idx"]).sum())
    mg.save("order_g60", **out)


def gen_bn_b4(M):
    tok = synth.tokens(50, 28, batch=4)
    out = {}
    for tag, train in [("train_b4", True), ("eval_b4", False)]:
        bb = mg.make_backbone(M["bb"], 28)
        bb.train(train)
        bb.dino.tokens = t(tok)
        with torch.no_grad():
            y = bb(torch.zeros(4, 3, 448, 448)).numpy()
        out[tag + "_sub"] = y.reshape(4, 784, 384)[:, ::41].copy()
        out[tag + "_sum64"] = y.astype(np.float64).sum(axis=(1, 2))
        out[tag + "_running_mean"] = bb.feature_norm.running_mean.numpy().copy()
        out[tag + "_running_var"] = bb.feature_norm.running_var.numpy().copy()
    mg.save("bn_tokens_b4", frame=50, grid=28, batch=4, sub_step=41, **out)


def main():
    M = mg._import_reference()
    gen_bn_b4(M)
    for tag in SPECS:
        gen_sequence(M, tag)
    gen_order(M)


if __name__ == "__main__":
    main()

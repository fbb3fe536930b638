#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz by RUNNING the reference's own code in this container.

Runs only where /root/reference exists (the authoring container).  The fixtures hold arrays only - inputs that
cannot be regenerated from tests/synth.py seeds, and the reference's outputs.  No reference source text is stored.

What is driven:
  * models.keypoint_selector.KeypointSelector          - imported as is (torch only)
  * models.descriptor_refiner.DescriptorRefiner        - imported as is (torch only)
  * models.dino_backbone.DinoBackbone                  - module imported with an *inert* placeholder for the
    absent third-party `timm`; the object is built without __init__ (which would fetch remote weights) and
    given a stand-in `dino` whose forward_features returns the synthetic tokens, so that the reference's own
    forward (token drop + BatchNorm1d + reshape), extract_at_keypoints and patch_to_pixel run unchanged.
  * visualize_matches_sequence.SequenceMatcher.match_with_quality (M1), visualize_matches.MatchVisualizer.
    find_matches (M2), train.SemanticSLAMTrainer._find_matches (M3), test_descriptor_quality.
    DescriptorQualityTester.find_mutual_nearest_neighbors (M4) - script-resident; imported with inert
    placeholders for torchvision / cv2 / wandb / timm, none of which these functions touch.
  * Pillow (third-party, importable here) for A0 / A9 resampling: torchvision's Resize on a PIL image is
    Image.resize(size, BILINEAR) and ToTensor / Normalize are one line of fp32 arithmetic each, restated here.

Usage:  python tests/golden/make_golden.py        (writes next to this file)
"""
from __future__ import annotations

import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True
REF = "/root/reference/semantic-slam"

import torch  # noqa: E402

import synth  # noqa: E402

torch.set_num_threads(4)


def _import_reference():
    for name in ["timm", "torchvision", "torchvision.transforms", "cv2", "wandb"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "test"))
    import models.dino_backbone as m_bb
    import models.keypoint_selector as m_sel
    import models.descriptor_refiner as m_ref
    import visualize_matches_sequence as m_vms
    import visualize_matches as m_vm
    mods = dict(bb=m_bb, sel=m_sel, ref=m_ref, vms=m_vms, vm=m_vm)
    try:
        # train.py / test_descriptor_quality.py import data.tum_dataset (needs torchvision.transforms attrs)
        tv = sys.modules["torchvision.transforms"]
        for attr in ["Compose", "Resize", "ToTensor", "Normalize", "ColorJitter", "GaussianBlur",
                     "RandomApply", "functional"]:
            if not hasattr(tv, attr):
                setattr(tv, attr, object)
        sys.modules["torchvision"].transforms = tv
        import train as m_train
        mods["train"] = m_train
    except Exception as e:  # pragma: no cover
        print("train.py not importable here:", type(e).__name__, e)
    try:
        import test_descriptor_quality as m_tdq
        mods["tdq"] = m_tdq
    except Exception as e:  # pragma: no cover
        print("test_descriptor_quality.py not importable here:", type(e).__name__, e)
    return mods


class _TokenDino(torch.nn.Module):
    """Stand-in for the third-party ViT: returns the tokens it was handed (dino_backbone.py:85 call site)."""

    def __init__(self):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        self.embed_dim = synth.C_FEAT
        self.tokens = None

    def forward_features(self, images):
        return self.tokens


def make_backbone(m_bb, grid: int):
    bb = m_bb.DinoBackbone.__new__(m_bb.DinoBackbone)
    torch.nn.Module.__init__(bb)
    bb.model_name = "synthetic"
    bb.input_size = grid * 16
    bb.patch_size = 16
    bb.grid_h = bb.grid_w = grid
    bb.num_patches = grid * grid
    bb.dino = _TokenDino()
    bb.embed_dim = synth.C_FEAT
    bb.n_storage_tokens = 4
    bb.feature_norm = torch.nn.BatchNorm1d(synth.C_FEAT, affine=True)
    return bb


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    out = {}
    for k, v in arrs.items():
        if isinstance(v, str):
            v = np.frombuffer(v.encode(), dtype=np.uint8)
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def load_selector(m_sel, seed=0, hidden=256):
    sel = m_sel.KeypointSelector(synth.C_FEAT, hidden).eval()
    sel.load_state_dict({k: t(v) for k, v in synth.selector_state(seed, hidden=hidden).items()})
    return sel


def load_refiner(m_ref, seed=0):
    ref = m_ref.DescriptorRefiner(synth.C_FEAT, 384, 128, 4).eval()
    ref.load_state_dict({k: t(v) for k, v in synth.refiner_state(seed).items()})
    return ref


def bn_features(m_bb, tok: np.ndarray, grid: int, train: bool = True) -> torch.Tensor:
    bb = make_backbone(m_bb, grid)
    bb.train(train)
    bb.dino.tokens = t(tok)
    with torch.no_grad():
        return bb(torch.zeros(tok.shape[0], 3, 16 * grid, 16 * grid))


# ----------------------------------------------------------------------------------------------- A2
def gen_bn(M):
    tok = synth.tokens(0, 28, batch=2)
    out = {}
    for tag, train, rows in [("train_b1_f0", True, slice(0, 1)), ("train_b1_f1", True, slice(1, 2)),
                             ("train_b2", True, slice(0, 2)), ("eval_b2", False, slice(0, 2))]:
        bb = make_backbone(M["bb"], 28)
        bb.train(train)
        bb.dino.tokens = t(tok[rows])
        with torch.no_grad():
            y = bb(torch.zeros(tok[rows].shape[0], 3, 448, 448)).numpy()
        out[tag + "_sub"] = y.reshape(y.shape[0], 784, 384)[:, ::13].copy()
        out[tag + "_sum64"] = y.astype(np.float64).sum(axis=(1, 2))
        out[tag + "_running_mean"] = bb.feature_norm.running_mean.numpy().copy()
        out[tag + "_running_var"] = bb.feature_norm.running_var.numpy().copy()
    save("bn_tokens", frame=0, grid=28, batch=2, sub_step=13, **out)


# ----------------------------------------------------------------------------------------- A3 / A4 / A5
def distinct(a):
    return np.unique(a).size == a.size


def run_select(sel, sal_map: np.ndarray, K, radius=2, pct=0.50):
    g = sal_map.shape[0]
    with torch.no_grad():
        kp, sc = sel.select_keypoints(t(sal_map).reshape(1, g, g, 1), num_keypoints=K, nms_radius=radius,
                                      min_score_percentile=pct)
    kp = kp[0].numpy()
    return kp, sc[0].numpy(), (kp[:, 1] * g + kp[:, 0]).astype(np.int32)


def gen_selector(M):
    sel = load_selector(M["sel"], 0, 256)
    out = {}
    feats = {}
    for grid, frame, K in [(28, 1, 500), (40, 2, 1024), (60, 3, 2048)]:
        f = bn_features(M["bb"], synth.tokens(frame, grid), grid)
        feats[grid] = f
        with torch.no_grad():
            sal = sel(f)[0, :, :, 0].numpy()
        assert distinct(sal), f"saliency G={grid} has exact ties; change the seed"
        kp, sc, idx = run_select(sel, sal, K)
        srt = np.sort(sal.ravel())[::-1]
        out[f"g{grid}_saliency"] = sal
        out[f"g{grid}_kp"] = kp
        out[f"g{grid}_scores"] = sc
        out[f"g{grid}_idx"] = idx
        out[f"g{grid}_K"] = K
        out[f"g{grid}_frame"] = frame
        out[f"g{grid}_min_gap"] = np.min(srt[:-1] - srt[1:])
        if grid == 28:
            with torch.no_grad():
                out["g28_nms"] = sel._apply_nms(t(sal)[None], 2)[0].numpy()
    # hidden=128 (the class default, keypoint_selector.py:25)
    sel128 = load_selector(M["sel"], 1, 128)
    with torch.no_grad():
        out["h128_saliency"] = sel128(feats[28])[0, :, :, 0].numpy()
    save("selector", **out)

    # branch coverage of select_keypoints on directly supplied saliency maps
    rng = np.random.Generator(np.random.PCG64(77))
    cases = {}

    def rand_map(g, lo, hi):
        while True:
            m = (rng.random((g, g)) * (hi - lo) + lo).astype(np.float32)
            if distinct(m):
                return m

    def add(tag, m, K, radius=2, pct=0.5):
        kp, sc, idx = run_select(sel, m, K, radius, pct)
        cases[tag + "_map"] = m
        cases[tag + "_K"] = K
        cases[tag + "_radius"] = radius
        cases[tag + "_pct"] = np.float64(pct)
        cases[tag + "_kp"] = kp
        cases[tag + "_scores"] = sc
        cases[tag + "_idx"] = idx

    add("A_k10", rand_map(28, 0.0, 1.0), 10)                       # |V| >= K: top-K of survivors
    add("A_k25_r1", rand_map(28, 0.0, 1.0), 25, radius=1)
    add("Bloop_r0", rand_map(28, 0.0, 1.0), 500, radius=0)         # no NMS: lower-percentile loop satisfied
    add("Bloop_r0_g40", rand_map(40, 0.2, 0.9), 1024, radius=0)
    m = rand_map(28, 0.0, 1.0)                                      # radius 1: loop satisfied at some percentile
    v = int((M["sel"].KeypointSelector._apply_nms(sel, t(m)[None], 1)[0].numpy() > max(np.quantile(m, 0.5), 0.1)).sum())
    add("Bloop_r1", m, v + 3, radius=1)
    add("Belse_r2", rand_map(28, 0.0, 1.0), 500)                    # pad with top raw saliency
    add("Belse_r3_g40", rand_map(40, 0.0, 1.0), 1024, radius=3)
    add("C_low", rand_map(28, 0.0, 0.0999), 100)                    # nothing above the 0.1 floor
    add("floor", rand_map(28, 0.0, 0.19), 300)                      # median < 0.1 -> floor 0.1 active
    add("pct70", rand_map(28, 0.0, 1.0), 200, pct=0.70)
    add("pct25_r0", rand_map(28, 0.0, 1.0), 700, radius=0, pct=0.25)
    add("full", rand_map(28, 0.0, 1.0), 784)                        # K == number of cells
    save("select_cases", tags=",".join(sorted({k.rsplit('_', 1)[0] for k in cases if k.endswith('_map')})), **cases)

    # quantile / lerp arithmetic on its own (torch.quantile, keypoint_selector.py:106,140)
    qs = []
    for n in (784, 1600, 3600, 100, 17):
        for _ in range(6):
            v = rng.random(n).astype(np.float32)
            for q in (0.5, 0.4, 0.3, 0.2, 0.1, 0.7, 0.25):
                qs.append((n, q, float(torch.quantile(t(v), q).item()), v))
    save("quantile", n=np.array([a[0] for a in qs]), q=np.array([a[1] for a in qs]),
         val=np.array([a[2] for a in qs], np.float32),
         data=np.concatenate([a[3] for a in qs]))
    return feats, out


# ----------------------------------------------------------------------------------------- A6 / A7 / A8
def gen_refine(M, feats, sel_out):
    ref = load_refiner(M["ref"], 0)
    bb = make_backbone(M["bb"], 28)
    out = {}
    kp = t(sel_out["g28_kp"])[None]
    with torch.no_grad():
        samp = bb.extract_at_keypoints(feats[28], kp)
        desc = ref(samp)
        pix = bb.patch_to_pixel(kp)
        back = bb.pixel_to_patch(pix)
    out["g28_sampled_sub"] = samp[0, ::10].numpy()
    out["g28_desc"] = desc[0].numpy()
    out["g28_pix"] = pix[0].numpy()
    out["g28_pix_back"] = back[0].numpy()
    # general (non-integer, border, out-of-range) coordinates through the same grid_sample call
    rng = np.random.Generator(np.random.PCG64(5))
    kq = (rng.random((64, 2)) * 29.0 - 1.0).astype(np.float32)
    kq[:8] = np.array([[0, 0], [27, 27], [27, 0], [0, 27], [26.5, 27], [27, 26.5], [-0.25, 3], [3, 27.75]], np.float32)
    with torch.no_grad():
        out["frac_kp"] = kq
        out["frac_sampled"] = bb.extract_at_keypoints(feats[28], t(kq)[None])[0].numpy()
        # refiner on its own, on rows that are not grid samples
        x = synth._normal(np.random.Generator(np.random.PCG64(6)), (96, 384), 2.0)
        out["mlp_in"] = x
        out["mlp_out"] = ref(t(x)[None])[0].numpy()
    for grid, K in [(40, 1024)]:
        kpg = t(sel_out[f"g{grid}_kp"])[None]
        bbg = make_backbone(M["bb"], grid)
        with torch.no_grad():
            d = ref(bbg.extract_at_keypoints(feats[grid], kpg))[0].numpy()
        out[f"g{grid}_desc_sub"] = d[::8]
    save("gather_refine", **out)
    return out["g28_desc"]


# ------------------------------------------------------------------------------------------- M1 .. M5
def pair(seed, n, m, dup, noise=0.25):
    d1 = synth.unit_descriptors(seed, n, 128, dup)
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


def gaps(d1, d2):
    s = (t(d1) @ t(d2).t()).numpy()

    def g(a):
        srt = np.sort(a, axis=1)
        d = srt[:, -1] - srt[:, -2]
        nz = d[d > 0]
        return float(nz.min()) if nz.size else 0.0
    return g(s), g(s.T)


def gen_match(M):
    out = {}
    mq = M["vms"].SequenceMatcher.match_with_quality
    specs = {"p500": (11, 500, 500, 30), "p500x480": (12, 500, 480, 20), "p1024": (13, 1024, 1024, 40),
             "p2048": (14, 2048, 2048, 64), "p33x70": (15, 33, 70, 0)}
    for tag, (seed, n, m, dup) in specs.items():
        d1, d2, s1, s2, i1, i2 = pair(seed, n, m, dup)
        out[f"{tag}_spec"] = np.array([seed, n, m, dup])
        out[f"{tag}_rowgap"], out[f"{tag}_colgap"] = gaps(d1, d2)
        runs = {
            "default": dict(),
            "cli": dict(saliency_weight=0.3, min_saliency=0.5, min_descriptor_sim=0.7, intensity1=i1,
                        intensity2=i2, min_intensity=0.15),
            "loose": dict(saliency_weight=0.45, min_saliency=0.0, min_descriptor_sim=-1.0),
            "tight": dict(min_saliency=0.75, min_descriptor_sim=0.9, intensity1=i1, intensity2=i2,
                          min_intensity=0.6),
            "none": dict(min_descriptor_sim=2.0),
        }
        for rtag, kw in runs.items():
            mt, q = mq(d1, d2, s1, s2, **kw)
            assert mt.dtype == np.int64 and q.dtype == np.float32
            out[f"{tag}_{rtag}_matches"] = mt
            out[f"{tag}_{rtag}_quality"] = q
        if n <= 1024:
            m2 = M["vm"].MatchVisualizer.find_matches(None, d1, d2, ratio_thresh=0.8)
            out[f"{tag}_m2_ij"] = np.array([(a, b) for a, b, _ in m2], np.int64).reshape(-1, 2)
            out[f"{tag}_m2_sim"] = np.array([c for _, _, c in m2], np.float32)
            if "tdq" in M:
                m4, dist = M["tdq"].DescriptorQualityTester.find_mutual_nearest_neighbors(None, d1, d2, 0.9)
                out[f"{tag}_m4_matches"] = m4.astype(np.int64)
                out[f"{tag}_m4_dist"] = dist.astype(np.float32)
            # M5 (test_tracking.py:159-161), restated: it is three numpy calls inside a longer method
            sim = d1 @ d2.T
            out[f"{tag}_m5_count"] = int((sim.max(axis=1) > 0.8).sum())
            out[f"{tag}_rowmax"] = sim.max(axis=1)
    # exact-threshold semantics: python-float thresholds against fp32 tensors (visualize_matches_sequence.py:166-175)
    d1, d2, s1, s2, i1, i2 = pair(16, 64, 64, 0, noise=0.0)
    s1[:] = 0.5
    s2[:] = 0.5
    mt, q = mq(d1, d2, s1, s2, min_saliency=0.5, min_descriptor_sim=0.7)
    out["edge_matches"], out["edge_quality"] = mt, q
    if "train" in M:
        fm = M["train"].SemanticSLAMTrainer._find_matches
        b1, b2 = [], []
        for seed, noise in [(21, 0.2), (22, 0.6), (23, 1.5)]:
            d1, d2, *_ = pair(seed, 200, 200, 10, noise)
            b1.append(d1)
            b2.append(d2)
        with torch.no_grad():
            out["m3_matches"] = fm(None, t(np.stack(b1)), t(np.stack(b2))).numpy()
        out["m3_seeds"] = np.array([21, 22, 23])
        out["m3_noise"] = np.array([0.2, 0.6, 1.5])
    save("matchers", **out)


# ---------------------------------------------------------------------------------------------- A0 / A9
MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def gen_preprocess(sel_out):
    from PIL import Image
    out = {}
    for tag, (h, w, size, frame) in {"vga448": (480, 640, 448, 0), "vga640": (480, 640, 640, 1),
                                     "big960": (960, 1280, 960, 2), "odd": (231, 517, 112, 3)}.items():
        img = synth.image(frame, h, w)
        pil = Image.fromarray(img, "RGB")
        rs = np.asarray(pil.resize((size, size), Image.BILINEAR))       # == torchvision Resize on a PIL image
        x = rs.astype(np.float32) / np.float32(255)                     # ToTensor
        x = (x - MEAN) / STD                                            # Normalize
        chw = np.ascontiguousarray(x.transpose(2, 0, 1))
        out[f"{tag}_spec"] = np.array([h, w, size, frame])
        out[f"{tag}_resized_rows"] = rs[::8].copy()
        out[f"{tag}_resized_sha"] = sha(rs)
        out[f"{tag}_chw_sum64"] = chw.astype(np.float64).sum(axis=(1, 2))
        out[f"{tag}_chw_rows"] = chw[:, ::16].copy()
        # A9 (visualize_matches_sequence.py:88-95): default-filter resize -> "L" -> /255
        gray = np.asarray(pil.resize((size, size)).convert("L"))
        out[f"{tag}_gray_rows"] = gray[::8].copy()
        out[f"{tag}_gray_sha"] = sha(gray)
        if tag == "vga448":
            kp = sel_out["g28_kp"] * 16 + 8
            g = gray.astype(np.float32) / 255.0
            xs = np.clip(kp[:, 0].round().astype(int), 0, g.shape[1] - 1)
            ys = np.clip(kp[:, 1].round().astype(int), 0, g.shape[0] - 1)
            out["vga448_intensity"] = g[ys, xs]
    save("preprocess", **out)


# ---------------------------------------------------------------------------------------------- end to end
def gen_e2e(M):
    from PIL import Image
    sel = load_selector(M["sel"], 0, 256)
    ref = load_refiner(M["ref"], 0)
    toks = synth.token_sequence(3, 28)
    imgs = synth.image_sequence(3)
    bb = make_backbone(M["bb"], 28)
    bb.train(True)                         # visualize_* never call backbone.eval()
    frames = []
    for i in range(3):
        bb.dino.tokens = t(toks[i:i + 1])
        with torch.no_grad():
            f = bb(torch.zeros(1, 3, 448, 448))
            sal = sel(f)
            kp, sc = sel.select_keypoints(sal, num_keypoints=500)
            desc = ref(bb.extract_at_keypoints(f, kp))
            pix = bb.patch_to_pixel(kp)[0].numpy()
        gray = np.asarray(Image.fromarray(imgs[i], "RGB").resize((448, 448)).convert("L"), dtype=np.float32) / 255.0
        xs = np.clip(pix[:, 0].round().astype(int), 0, 447)
        ys = np.clip(pix[:, 1].round().astype(int), 0, 447)
        frames.append(dict(kp=kp[0].numpy(), scores=sc[0].numpy(), desc=desc[0].numpy(), pix=pix,
                           intensity=gray[ys, xs], sal=sal[0, :, :, 0].numpy()))
        assert distinct(frames[-1]["sal"])
    out = {}
    for i, f in enumerate(frames):
        out[f"f{i}_kp"] = f["kp"]
        out[f"f{i}_scores"] = f["scores"]
        out[f"f{i}_intensity"] = f["intensity"]
        out[f"f{i}_desc_sub"] = f["desc"][::5]
        out[f"f{i}_desc_sha"] = sha(f["desc"])
    mq = M["vms"].SequenceMatcher.match_with_quality
    for a, b in [(0, 1), (1, 2), (0, 2)]:
        fa, fb = frames[a], frames[b]
        mt, q = mq(fa["desc"], fb["desc"], fa["scores"], fb["scores"], saliency_weight=0.3, min_saliency=0.5,
                   min_descriptor_sim=0.7, intensity1=fa["intensity"], intensity2=fb["intensity"],
                   min_intensity=0.15)
        out[f"pair{a}{b}_matches"] = mt
        out[f"pair{a}{b}_quality"] = q
        out[f"pair{a}{b}_rowgap"], out[f"pair{a}{b}_colgap"] = gaps(fa["desc"], fb["desc"])
    save("e2e", n_frames=3, **out)


def main():
    M = _import_reference()
    gen_bn(M)
    feats, sel_out = gen_selector(M)
    gen_refine(M, feats, sel_out)
    gen_match(M)
    gen_preprocess(sel_out)
    gen_e2e(M)


if __name__ == "__main__":
    main()

"""Pins the CPU oracle (oracle/sslam_oracle.c) against golden vectors produced by the reference's own modules
(tests/golden/make_golden.py).  CPU-only; runs in the `-m "not gpu"` suite.

Bars (BASELINE.json north_star): keypoint indices and match pairs identical; float values within 1e-4.
The observed float differences are ~1e-6 (fp32 summation-order noise), so the tests use tighter bounds.
"""
import hashlib
import os

import numpy as np
import pytest

import synth
from oracle import ora

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def as_str(a):
    return bytes(a).decode()


# ------------------------------------------------------------------------------------------------ A2
def test_bn_tokens_train_and_eval():
    g = gold("bn_tokens")
    tok = synth.tokens(0, 28, batch=2)
    y1, mean, var = ora.bn_tokens(tok, group=1, train=True)
    assert np.abs(y1[0, ::13] - g["train_b1_f0_sub"][0]).max() < 5e-6
    assert np.abs(y1[1, ::13] - g["train_b1_f1_sub"][0]).max() < 5e-6
    # per-channel sums of the whole output, in float64
    assert np.abs(y1[0].astype(np.float64).sum(axis=0) - g["train_b1_f0_sum64"][0]).max() < 1e-3
    y2, mean2, var2 = ora.bn_tokens(tok, group=2, train=True)
    assert np.abs(y2[:, ::13] - g["train_b2_sub"]).max() < 5e-6
    ye, _, _ = ora.bn_tokens(tok, group=2, train=False)
    assert np.array_equal(ye[:, ::13], g["eval_b2_sub"])          # x / sqrt(1 + 1e-5): bit-exact
    # running statistics after one train-mode call (momentum 0.1, unbiased variance)
    rm = 0.1 * mean[0]
    rv = 0.9 + 0.1 * var[0] * (784.0 / 783.0)
    assert np.abs(rm - g["train_b1_f0_running_mean"]).max() < 1e-6
    assert np.abs(rv - g["train_b1_f0_running_var"]).max() < 1e-5
    rv2 = 0.9 + 0.1 * var2[0] * (1568.0 / 1567.0)
    assert np.abs(rv2 - g["train_b2_running_var"]).max() < 1e-5


def test_bn_tokens_batch_of_four():
    """train.py:300-302 runs the backbone on B = 4: statistics over 4 x 784 tokens (bn_tokens_b4.npz, reference output)."""
    g = gold("bn_tokens_b4")
    tok = synth.tokens(int(g["frame"]), 28, batch=4)
    step = int(g["sub_step"])
    y, mean, var = ora.bn_tokens(tok, group=4, train=True)
    assert np.abs(y[:, ::step] - g["train_b4_sub"]).max() < 5e-6
    assert np.abs(y.astype(np.float64).sum(axis=1) - g["train_b4_sum64"]).max() < 1e-3
    assert np.abs(0.1 * mean[0] - g["train_b4_running_mean"]).max() < 1e-6
    assert np.abs(0.9 + 0.1 * var[0] * (3136.0 / 3135.0) - g["train_b4_running_var"]).max() < 1e-5
    ye, _, _ = ora.bn_tokens(tok, group=4, train=False)
    assert np.array_equal(ye[:, ::step], g["eval_b4_sub"])


# ------------------------------------------------------------------------------------------- A3 / A4 / A5
@pytest.fixture(scope="module")
def sel_feats():
    feats = {}
    for grid, frame in [(28, 1), (40, 2), (60, 3)]:
        feats[grid] = ora.bn_tokens(synth.tokens(frame, grid), group=1, train=True)[0].reshape(1, grid, grid, 384)
    return feats


@pytest.mark.parametrize("grid,K", [(28, 500), (40, 1024), (60, 2048)])
def test_selector_saliency_and_default_selection(sel_feats, grid, K):
    g = gold("selector")
    sal = ora.selector_saliency(sel_feats[grid], synth.selector_state(0))[0]
    assert np.abs(sal - g[f"g{grid}_saliency"]).max() < 5e-6
    # selection on the oracle's own saliency: indices identical to the reference's (the fixture is tie-free and
    # its smallest gap between consecutive sorted saliencies is recorded)
    assert g[f"g{grid}_min_gap"] > 0
    kp, sc, idx, st = ora.select_keypoints(sal, K)
    assert st[0] == 0
    gi = g[f"g{grid}_idx"]
    if g[f"g{grid}_min_gap"] > 5e-7:
        assert np.array_equal(idx[0], gi)
        assert np.array_equal(kp[0], g[f"g{grid}_kp"])
    else:
        # 3600 saliencies in [0.2, 0.92] cannot all be well separated: where the reference's own values are
        # within 1e-6 of each other (the fp32 summation-order noise of a 3456-term dot product; the
        # value bar is 1e-4), neighbours may swap in the ranking.  Same keypoint set, and every out-of-place
        # entry is such a near-tie.
        ref_sal = g[f"g{grid}_saliency"].ravel()
        bad = np.nonzero(idx[0] != gi)[0]
        assert sorted(idx[0]) == sorted(gi) and len(bad) <= 8
        assert np.abs(ref_sal[idx[0][bad]] - ref_sal[gi[bad]]).max() <= 1e-6
    assert np.abs(sc[0] - g[f"g{grid}_scores"]).max() < 5e-6
    # and on the reference's saliency bits: everything exact
    kp, sc, idx, st = ora.select_keypoints(g[f"g{grid}_saliency"], K)
    assert np.array_equal(idx[0], g[f"g{grid}_idx"]) and np.array_equal(sc[0], g[f"g{grid}_scores"])


def test_selector_hidden_128(sel_feats):
    g = gold("selector")
    sal = ora.selector_saliency(sel_feats[28], synth.selector_state(1, hidden=128))[0]
    assert np.abs(sal - g["h128_saliency"]).max() < 5e-6


def test_nms_map():
    g = gold("selector")
    assert np.array_equal(ora.nms(g["g28_saliency"], 2), g["g28_nms"])


def test_select_keypoints_branches():
    g = gold("select_cases")
    tags = as_str(g["tags"]).split(",")
    assert len(tags) >= 12
    for tag in tags:
        m = g[tag + "_map"]
        kp, sc, idx, st = ora.select_keypoints(m, int(g[tag + "_K"]), int(g[tag + "_radius"]), float(g[tag + "_pct"]))
        assert st[0] == 0, tag
        assert np.array_equal(idx[0], g[tag + "_idx"]), tag
        assert np.array_equal(kp[0], g[tag + "_kp"]), tag
        assert np.array_equal(sc[0], g[tag + "_scores"]), tag


def test_select_keypoints_k_too_large_flags_status():
    # the reference's torch.topk raises when K - |V| exceeds the number of cells
    g = gold("select_cases")
    _, _, _, st = ora.select_keypoints(g["Belse_r2_map"], 28 * 28 + 200)
    assert st[0] == 1


def test_quantile_bit_exact():
    g = gold("quantile")
    off = 0
    for n, q, val in zip(g["n"], g["q"], g["val"]):
        v = g["data"][off:off + n]
        off += n
        assert ora.quantile(v, float(q)) == val, (n, q)


# ------------------------------------------------------------------------------------------- A6 / A7 / A8
def test_gather_refine(sel_feats):
    g = gold("gather_refine")
    s = gold("selector")
    feat = sel_feats[28]
    kp = s["g28_kp"][None]
    samp = ora.gather(feat, kp)
    assert np.abs(samp[0, ::10] - g["g28_sampled_sub"]).max() < 2e-5
    desc = ora.refine(samp, synth.refiner_state(0))
    assert np.abs(desc[0] - g["g28_desc"]).max() < 5e-6
    assert np.abs(np.linalg.norm(desc[0].astype(np.float64), axis=1) - 1).max() < 1e-6
    # duplicated keypoints must give bit-identical descriptors
    idx = s["g28_idx"]
    first = {}
    dups = 0
    for r, c in enumerate(idx):
        if c in first:
            dups += 1
            assert np.array_equal(desc[0, r], desc[0, first[c]])
        else:
            first[c] = r
    assert dups > 10
    assert np.array_equal(ora.patch_to_pixel(kp[0]), g["g28_pix"])
    assert np.array_equal(ora.pixel_to_patch(g["g28_pix"]), g["g28_pix_back"])


def test_gather_fractional_and_out_of_range(sel_feats):
    g = gold("gather_refine")
    samp = ora.gather(sel_feats[28], g["frac_kp"][None])
    assert np.abs(samp[0] - g["frac_sampled"]).max() < 2e-5


def test_refiner_mlp_alone():
    g = gold("gather_refine")
    out = ora.refine(g["mlp_in"], synth.refiner_state(0))
    assert np.abs(out - g["mlp_out"]).max() < 5e-6


def test_gather_refine_g40(sel_feats):
    g = gold("gather_refine")
    s = gold("selector")
    desc = ora.refine(ora.gather(sel_feats[40], s["g40_kp"][None]), synth.refiner_state(0))
    assert np.abs(desc[0, ::8] - g["g40_desc_sub"]).max() < 5e-6


# ------------------------------------------------------------------------------------------------- M1
def _pair(seed, n, m, dup, noise=0.25):
    # must mirror tests/golden/make_golden.py:pair()
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


RUNS = {
    "default": lambda i1, i2: dict(),
    "cli": lambda i1, i2: dict(saliency_weight=0.3, min_saliency=0.5, min_descriptor_sim=0.7, intensity1=i1,
                               intensity2=i2, min_intensity=0.15),
    "loose": lambda i1, i2: dict(saliency_weight=0.45, min_saliency=0.0, min_descriptor_sim=-1.0),
    "tight": lambda i1, i2: dict(min_saliency=0.75, min_descriptor_sim=0.9, intensity1=i1, intensity2=i2,
                                 min_intensity=0.6),
    "none": lambda i1, i2: dict(min_descriptor_sim=2.0),
}


@pytest.mark.parametrize("tag", ["p500", "p500x480", "p1024", "p2048", "p33x70"])
def test_match_with_quality(tag):
    g = gold("matchers")
    seed, n, m, dup = (int(v) for v in g[f"{tag}_spec"])
    d1, d2, s1, s2, i1, i2 = _pair(seed, n, m, dup)
    assert g[f"{tag}_rowgap"] > 4e-6 and g[f"{tag}_colgap"] > 4e-6     # fixtures free of near-ties
    for rtag, kw in RUNS.items():
        mt, q = ora.match_with_quality(d1, d2, s1, s2, **kw(i1, i2))
        assert mt.dtype == np.int64 and q.dtype == np.float32
        assert np.array_equal(mt, g[f"{tag}_{rtag}_matches"]), (tag, rtag)
        if len(q):
            assert np.abs(q - g[f"{tag}_{rtag}_quality"]).max() < 1e-6, (tag, rtag)
        else:
            assert mt.shape == (0, 2) and q.shape == (0,)


def test_match_threshold_edge_semantics():
    # python-float thresholds are rounded to fp32 before the comparison: 0.5 >= 0.5 keeps every match
    g = gold("matchers")
    d1, d2, s1, s2, _, _ = _pair(16, 64, 64, 0, noise=0.0)
    s1[:] = 0.5
    s2[:] = 0.5
    mt, q = ora.match_with_quality(d1, d2, s1, s2, min_saliency=0.5, min_descriptor_sim=0.7)
    assert np.array_equal(mt, g["edge_matches"])
    assert np.abs(q - g["edge_quality"]).max() < 1e-6


def test_rowmax_and_tracking_count():
    g = gold("matchers")
    for tag in ["p500", "p1024"]:
        seed, n, m, dup = (int(v) for v in g[f"{tag}_spec"])
        d1, d2, *_ = _pair(seed, n, m, dup)
        _, s12, _, _ = ora.sim_argmax(d1, d2)
        assert np.abs(s12 - g[f"{tag}_rowmax"]).max() < 1e-6
        assert int((s12 > np.float32(0.8)).sum()) == int(g[f"{tag}_m5_count"])     # M5, test_tracking.py:159-161


# ---------------------------------------------------------------------------------------------- A0 / A9
@pytest.mark.parametrize("tag", ["vga448", "vga640", "big960", "odd"])
def test_pillow_resize_bit_exact(tag):
    g = gold("preprocess")
    h, w, size, frame = (int(v) for v in g[f"{tag}_spec"])
    img = synth.image(frame, h, w)
    rs, chw = ora.resize_rgb(img, size, bicubic=False)
    assert sha(rs) == as_str(g[f"{tag}_resized_sha"])
    assert np.array_equal(rs[::8], g[f"{tag}_resized_rows"])
    assert np.array_equal(chw[:, ::16], g[f"{tag}_chw_rows"])                # ToTensor + Normalize: bit-exact
    gray = ora.gray_resized(img, size)
    assert sha(gray) == as_str(g[f"{tag}_gray_sha"])


def test_intensity_lookup():
    g = gold("preprocess")
    s = gold("selector")
    img = synth.image(0, 480, 640)
    out = ora.intensity(img, 448, s["g28_kp"] * 16 + 8)
    assert np.array_equal(out, g["vga448_intensity"])


# ------------------------------------------------------------------------------------------- end to end
def test_end_to_end_three_frames():
    g = gold("e2e")
    toks = synth.token_sequence(3, 28)
    imgs = synth.image_sequence(3)
    feat, _, _ = ora.bn_tokens(toks, group=1, train=True)
    feat = feat.reshape(3, 28, 28, 384)
    sal = ora.selector_saliency(feat, synth.selector_state(0))
    kp, sc, idx, st = ora.select_keypoints(sal, 500)
    desc = ora.refine(ora.gather(feat, kp), synth.refiner_state(0))
    inten = [ora.intensity(imgs[i], 448, ora.patch_to_pixel(kp[i])) for i in range(3)]
    for i in range(3):
        assert np.array_equal(kp[i], g[f"f{i}_kp"])                           # keypoint indices: exact
        assert np.abs(sc[i] - g[f"f{i}_scores"]).max() < 5e-6
        assert np.array_equal(inten[i], g[f"f{i}_intensity"])
        assert np.abs(desc[i, ::5] - g[f"f{i}_desc_sub"]).max() < 1e-5
    for a, b in [(0, 1), (1, 2), (0, 2)]:
        assert g[f"pair{a}{b}_rowgap"] > 4e-6 and g[f"pair{a}{b}_colgap"] > 4e-6
        mt, q = ora.match_with_quality(desc[a], desc[b], sc[a], sc[b], 0.3, 0.5, 0.7, inten[a], inten[b], 0.15)
        assert np.array_equal(mt, g[f"pair{a}{b}_matches"])                   # match pairs: exact
        assert np.abs(q - g[f"pair{a}{b}_quality"]).max() < 1e-5
        assert len(mt) > 100


def _oracle_sequence(tag):
    import e2e_check
    g = e2e_check.gold(tag)
    grid, K, n = int(g["grid"]), int(g["K"]), int(g["n_frames"])
    toks = synth.token_sequence(n, grid)
    imgs = synth.image_sequence(n, int(g["height"]), int(g["width"]))
    feat = ora.bn_tokens(toks)[0].reshape(n, grid, grid, 384)
    sal = ora.selector_saliency(feat, synth.selector_state(0))
    kp, sc, idx, st = ora.select_keypoints(sal, K)
    assert not st.any()
    desc = ora.refine(ora.gather(feat, kp), synth.refiner_state(0))
    inten = np.stack([ora.intensity(imgs[i], 16 * grid, ora.patch_to_pixel(kp[i])) for i in range(n)])

    def match(a, b):
        return ora.match_with_quality(desc[a], desc[b], sc[a], sc[b], intensity1=inten[a], intensity2=inten[b],
                                      **e2e_check.CLI)
    return e2e_check.check_sequence(tag, idx, sc, desc, inten, match), sal


@pytest.mark.parametrize("tag", ["e2e_g40", "e2e_g60"])
def test_end_to_end_larger_grids(tag):
    """8 frames at G = 40 / K = 1024 (configs[2]) and 16 at G = 60 / K = 2048 (configs[4]) through the reference's whole chain
    (tests/golden/make_golden_e2e_grids.py) against the oracle; bars in tests/e2e_check.py.  On the reference's own saliency
    BITS the selection is index-exact wherever that map has no exactly equal values (torch.topk leaves those unordered)."""
    import e2e_check
    rep, sal = _oracle_sequence(tag)
    g = e2e_check.gold(tag)
    K = int(g["K"])
    for i, f in enumerate(rep["frames"]):
        assert np.abs(sal[i] - g[f"f{i}_sal"]).max() < 5e-6
        if f["ref_exact_ties"] == 0:
            _, _, idx, _ = ora.select_keypoints(g[f"f{i}_sal"], K)
            assert np.array_equal(idx[0], g[f"f{i}_idx"].astype(np.int32)), (tag, i)
    s = e2e_check.summarise(rep["frames"])
    if tag == "e2e_g40":
        assert s["swapped_positions_max"] <= 2 and s["index_exact_frames"] >= 7, s
        assert all(p["index_exact"] for p in rep["pairs"])
    assert all(p["cells_equal"] for p in rep["pairs"]) and len(rep["pairs"]) >= 8


def test_keypoint_order_rate_g60():
    """The measured size of the one caveat of the parity claim: over 32 + 16 frames at G = 60 / K = 2048, how often the
    oracle's keypoint ORDER differs from torch's (the set never does).  tools/order_swap_rate.py prints the table quoted in
    DESIGN.md; here the rate is bounded so that a regression of the canonical order shows."""
    import e2e_check

    def idx_of(i, seed):
        feat = ora.bn_tokens(synth.tokens(seed, 60))[0].reshape(1, 60, 60, 384)
        return ora.select_keypoints(ora.selector_saliency(feat, synth.selector_state(0)), 2048)[2][0]
    frames = e2e_check.check_order_set(idx_of)
    s = e2e_check.summarise(frames)
    assert s["frames"] == 32 and s["swapped_positions_mean"] < 8 and s["max_gap"] <= e2e_check.NEAR_TIE, s


def test_canonical_exp_accuracy():
    xs = np.linspace(-20, 20, 4001, dtype=np.float32)
    got = np.array([ora.expf(x) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 2.5e-7
    assert ora.sigmoid(0.0) == np.float32(0.5)
    assert ora.sigmoid(-200.0) >= 0 and ora.sigmoid(200.0) == np.float32(1.0)


# --------------------------------------------------------------------------------------------- M2 / M3 / M4
@pytest.mark.parametrize("tag", ["p500", "p500x480", "p1024", "p33x70"])
def test_ratio_test_matchers(tag):
    g = gold("matchers")
    seed, n, m, dup = (int(v) for v in g[f"{tag}_spec"])
    d1, d2, *_ = _pair(seed, n, m, dup)
    m2 = ora.find_matches_m2(d1, d2, 0.8)
    assert np.array_equal(np.array([(a, b) for a, b, _ in m2], np.int64).reshape(-1, 2), g[f"{tag}_m2_ij"])
    if m2:
        assert np.abs(np.array([c for *_, c in m2], np.float32) - g[f"{tag}_m2_sim"]).max() < 1e-6
    m4, dist = ora.find_mnn_m4(d1, d2, 0.9)
    assert np.array_equal(m4, g[f"{tag}_m4_matches"])
    if len(dist):
        assert np.abs(dist - g[f"{tag}_m4_dist"]).max() < 1e-6


def test_batched_mnn_padding_m3():
    # train.py:410-449: per-sample mutual-NN pairs, zero-padded to the longest
    g = gold("matchers")
    want = g["m3_matches"]
    for b, (seed, noise) in enumerate(zip(g["m3_seeds"], g["m3_noise"])):
        d1, d2, *_ = _pair(int(seed), 200, 200, 10, float(noise))
        nn12, _, nn21, _ = ora.sim_argmax(d1, d2)
        idx1 = np.nonzero(nn21[nn12] == np.arange(200))[0]
        got = np.stack([idx1, nn12[idx1]], axis=1)
        assert np.array_equal(got, want[b, :len(got)])
        assert not want[b, len(got):].any()


# ------------------------------------------------------------------------------ wide, randomised reference-pinned sets
def test_select_wide_vs_reference():
    """64 random (grid, K, radius, percentile, map kind) cases: the oracle's keypoint indices and scores equal what the
    reference's KeypointSelector.select_keypoints returned (tests/golden/make_golden_wide.py); where the reference
    raises (K beyond what the grid can supply) the oracle flags the frame."""
    g = gold("select_wide")
    raised = 0
    for s in range(int(g["count"])):
        m, K, radius, pct = synth.wide_map(s)
        kp, sc, idx, st = ora.select_keypoints(m, K, radius, pct)
        want = g[f"s{s}_idx"]
        if want.size == 0:
            assert st[0] == 1, s
            raised += 1
            continue
        assert st[0] == 0, s
        assert np.array_equal(idx[0], want.astype(np.int32)), (s, m.shape, K, radius, pct)
        assert np.array_equal(sc[0].view(np.uint32), g[f"s{s}_scores"].view(np.uint32)), s
    assert raised < int(g["count"]) // 2


def test_match_wide_vs_reference():
    """32 random descriptor pairs (ragged sizes, duplicated rows = exact ties, random thresholds, with and without the
    intensity test): match pairs equal the reference's SequenceMatcher.match_with_quality, quality within 1e-6."""
    g = gold("match_wide")
    total = 0
    for s in range(int(g["count"])):
        d1, d2, s1, s2, kw = synth.wide_pair(s)
        mt, q = ora.match_with_quality(d1, d2, s1, s2, **kw)
        want = g[f"p{s}_matches"].astype(np.int64).reshape(-1, 2)
        assert np.array_equal(mt, want), (s, d1.shape, d2.shape, kw.keys())
        assert np.abs(q - g[f"p{s}_quality"]).max(initial=0.0) < 1e-6, s
        total += len(mt)
    assert total > 500


def test_bf16_mode_checker_tracks_the_exact_oracle():
    """oracle/ora_bf16.py (the definition the bf16 throughput mode's kernels are checked against: the reference's algorithm
    with bf16-rounded GEMM operands) stays within the mode's drift bounds of the reference's own fp32 outputs (goldens made
    by running the reference's modules) - it is the same algorithm, only the operand rounding differs."""
    from oracle.ora_bf16 import bf16_round, refine_bf16_ref, saliency_bf16_ref
    x = np.array([1.0, 1.00390625, 1.01171875, -3.0e-39, 65504.0], np.float32)         # ties to even, denormal, large
    assert bf16_round(x).tolist() == [1.0, 1.0, 1.015625, bf16_round(np.float32(-3.0e-39)).item(), 65536.0]
    g = gold("selector")
    feat = ora.bn_tokens(synth.tokens(1, 28))[0].reshape(1, 28, 28, 384)
    sal = saliency_bf16_ref(feat, synth.selector_state(0))[0]
    assert np.abs(sal - g["g28_saliency"]).max() < 3e-2
    assert np.abs(sal - g["g28_saliency"]).mean() < 3e-3
    gd = gold("gather_refine")
    xk = ora.gather(feat, g["g28_kp"][None])
    desc = refine_bf16_ref(xk.reshape(-1, 384), synth.refiner_state(0))
    cos = (desc * gd["g28_desc"]).sum(-1)
    assert cos.min() > 0.999 and np.abs(np.linalg.norm(desc, axis=-1) - 1).max() < 1e-6

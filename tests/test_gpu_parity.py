"""GPU parity tests: every HIP entry point of libsslam_hip.so, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bar: BIT-EXACT - indices and floats alike (the kernels implement the oracle's canonical
fp32 evaluation order), plus the reference's golden vectors for indices / match pairs.

Run on the GPU box: python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

import synth
from oracle import ora

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def hip(T):
    from sslam_amd import lib
    lib.lib()          # raises if libsslam_hip.so is not built: no fallback
    return lib


def dev(T, a):
    return T.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bits(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not np.array_equal(bits(got), bits(want)):
        bad = np.nonzero(bits(got).ravel() != bits(want).ravel())[0]
        raise AssertionError(f"{what}: {bad.size}/{got.size} elements differ; first at {bad[0]}: "
                             f"{got.ravel()[bad[0]]!r} vs {want.ravel()[bad[0]]!r}; "
                             f"max abs diff {np.abs(got.astype(np.float64) - want.astype(np.float64)).max():.3e}")


# ---------------------------------------------------------------------------------------------- MFMA contract
def test_mfma_chain_is_fmaf_chain(T, hip):
    """The premise of the whole design: v_mfma_f32_32x32x2_f32 accumulates as one fmaf chain in increasing k."""
    d1 = synth.unit_descriptors(1, 200, 128) * np.float32(3.7)
    d2 = synth.unit_descriptors(2, 130, 128) * np.float32(0.9)
    nn12, s12, nn21, s21, _ = hip.sim_argmax(dev(T, d1), 0, 200, dev(T, d2), 0, 130, 1, want_s21=True)
    o12, os12, o21, os21 = ora.sim_argmax(d1, d2)
    assert_bits(s12.cpu().numpy()[0], os12, "row max")
    assert_bits(s21.cpu().numpy()[0], os21, "col max")
    assert np.array_equal(nn12.cpu().numpy()[0], o12) and np.array_equal(nn21.cpu().numpy()[0], o21)


# --------------------------------------------------------------------------------------------------------- A2
@pytest.mark.parametrize("sweeps", [False, True])
@pytest.mark.parametrize("grid,batch,group,train", [(28, 2, 1, True), (28, 2, 2, True), (28, 2, 1, False), (40, 1, 1, True), (40, 3, 1, True),
                                                    (60, 3, 3, True), (60, 2, 1, True), (50, 2, 1, True), (33, 2, 1, True), (27, 3, 1, True),
                                                    (7, 4, 1, True), (3, 2, 1, True)])
def test_bn_tokens(T, hip, grid, batch, group, train, sweeps, knob):
    """Per-frame statistics (group 1) run register-resident - one HBM read - at every grid up to 60 x 60 (three forms: 4 or 2
    channels per lane, the largest with part of the rows in LDS); `sweeps` forces the three-sweep kernel that groups > 1, eval
    mode and larger grids use.  All bit-identical to the oracle."""
    if sweeps:
        if not (train and group == 1):
            pytest.skip("already the three-sweep kernel")
        knob("SSLAM_BN_FORM", 1)
    tok = synth.tokens(10 + grid, grid, batch)
    rng = np.random.Generator(np.random.PCG64(grid))
    gamma = (1 + 0.1 * rng.standard_normal(384)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(384)).astype(np.float32)
    rmean = (0.2 * rng.standard_normal(384)).astype(np.float32)
    rvar = (1 + 0.3 * rng.random(384)).astype(np.float32)
    y, mean, var = hip.bn_tokens(dev(T, tok), 5, group, dev(T, gamma), dev(T, beta), dev(T, rmean), dev(T, rvar), train, 1e-5)
    oy, omean, ovar = ora.bn_tokens(tok, 5, group, gamma, beta, rmean, rvar, train, 1e-5)
    assert_bits(y.cpu().numpy(), oy, "bn output")
    if train:
        assert_bits(mean.cpu().numpy(), omean, "batch mean")
        assert_bits(var.cpu().numpy(), ovar, "batch var")


# --------------------------------------------------------------------------------------------------------- A3
def _packed_selector(T, hip, sd):
    w1p = dev(T, hip.pack_conv3x3(sd["conv.0.weight"]))
    return (w1p, dev(T, sd["conv.0.bias"]), dev(T, sd["conv.2.weight"].reshape(-1)), dev(T, sd["conv.2.bias"]),
            sd["conv.0.weight"].shape[0])


@pytest.mark.parametrize("form", ["latency2", "latency", "throughput", "throughput_tail", "throughput_stage"])
@pytest.mark.parametrize("grid,frames,hidden", [(28, 3, 256), (40, 1, 256), (60, 1, 256), (28, 2, 128), (5, 2, 256), (14, 5, 256),
                                                (40, 3, 256), (60, 2, 256), (44, 3, 256)])
def test_selector_saliency(T, hip, grid, frames, hidden, form, knob):
    """Every launch shape of the conv, all bit-identical: 32-cell tiles split over two workgroups on the halo image (few
    frames), 32-row workgroups of 8 waves, 128-row workgroups on the halo image (G = 28), the same with the last partial round cut into 32-cell tiles, and the
    stage-per-tap form.  At G = 40 / 44 / 60 the throughput forms run the PER-FRAME tiling of the halo form (a tile that crosses a
    frame boundary there needs more image rows than the LDS image has): several frames, frame ends inside 32-cell tiles (G = 60:
    16 of 32 cells; G = 44: 16), and with rounds of 4 big tiles a big tile cut into its four quarters (G = 44, 3 frames: 45 big
    tiles = 11 rounds + 1)."""
    knob("SSLAM_CONV_LATENCY_ROWS", "0" if form.startswith("throughput") else str(1 << 30))
    knob("SSLAM_CONV_LAT2_ROWS", str(1 << 30) if form == "latency2" else "0")
    if form == "throughput_stage":
        knob("SSLAM_CONV_NO_HALO", "1")
    if form == "throughput_tail":
        knob("SSLAM_CONV_TAIL", "4")          # rounds of 4 big tiles: the rest of the rows go to 32-cell tiles
    sd = synth.selector_state(0 if hidden == 256 else 1, hidden=hidden)
    feat = ora.bn_tokens(synth.tokens(20 + grid, grid, frames))[0].reshape(frames, grid, grid, 384)
    w1p, b1, w2, b2, hs = _packed_selector(T, hip, sd)
    sal = hip.selector_saliency(dev(T, feat), w1p, b1, w2, b2, hs)
    assert_bits(sal.cpu().numpy(), ora.selector_saliency(feat, sd), "saliency")


def test_selector_matches_reference_golden(T, hip):
    g = gold("selector")
    sd = synth.selector_state(0)
    feat = ora.bn_tokens(synth.tokens(1, 28))[0].reshape(1, 28, 28, 384)
    w1p, b1, w2, b2, hs = _packed_selector(T, hip, sd)
    sal = hip.selector_saliency(dev(T, feat), w1p, b1, w2, b2, hs).cpu().numpy()[0]
    assert np.abs(sal - g["g28_saliency"]).max() < 5e-6


# ---------------------------------------------------------------------------------------------------- A4 / A5
def _select(T, hip, sal, K, radius=2, pct=0.5):
    kp, sc, idx, px, st = hip.select_keypoints(dev(T, sal), K, radius, pct)
    return kp.cpu().numpy(), sc.cpu().numpy(), idx.cpu().numpy(), px.cpu().numpy(), st.cpu().numpy()


def test_select_keypoints_golden_branches(T, hip):
    g = gold("select_cases")
    tags = bytes(g["tags"]).decode().split(",")
    for tag in tags:
        m = g[tag + "_map"]
        kp, sc, idx, px, st = _select(T, hip, m[None], int(g[tag + "_K"]), int(g[tag + "_radius"]), float(g[tag + "_pct"]))
        assert st[0] == 0, tag
        assert np.array_equal(idx[0], g[tag + "_idx"]), tag          # keypoint indices == the reference's
        assert np.array_equal(kp[0], g[tag + "_kp"]), tag
        assert_bits(sc[0], g[tag + "_scores"], tag)
        assert_bits(px[0], g[tag + "_kp"] * 16 + 8, tag)


@pytest.mark.parametrize("grid,K,frames", [(28, 500, 7), (40, 1024, 3), (60, 2048, 2), (64, 4096, 1), (7, 20, 4)])
def test_select_keypoints_vs_oracle(T, hip, grid, K, frames):
    rng = np.random.Generator(np.random.PCG64(grid * 7 + K))
    sal = rng.random((frames, grid, grid)).astype(np.float32)
    sal[0].ravel()[::5] = sal[0].ravel()[3]      # exact ties: canonical order must hold (value desc, index asc)
    if frames > 1:
        sal[1] = np.float32(0.5)                 # a constant map: every cell a plateau maximum, nothing above median
    for radius, pct in [(2, 0.5), (0, 0.5), (1, 0.3), (3, 0.8)]:
        kp, sc, idx, px, st = _select(T, hip, sal, K, radius, pct)
        okp, osc, oidx, ost = ora.select_keypoints(sal, K, radius, pct)
        assert np.array_equal(st, ost)
        assert np.array_equal(idx, oidx), (radius, pct)
        assert_bits(kp, okp, "kp")
        assert_bits(sc, osc, "scores")


def test_select_keypoints_k_too_large_sets_status(T, hip):
    """K beyond what torch.topk can deliver (the reference raises, keypoint_selector.py:160/176): status = 1 and every slot
    is still written - the oracle's truncate / pad rule (:186-199, sslam_oracle.c select_one), bit for bit."""
    g = gold("select_cases")
    K = 28 * 28 + 200
    m = g["Belse_r2_map"][None]
    kp, sc, idx, px, st = _select(T, hip, m, K)
    okp, osc, oidx, ost = ora.select_keypoints(m, K, 2, 0.5)
    assert st[0] == 1 and ost[0] == 1
    assert np.array_equal(idx, oidx) and np.array_equal(kp, okp)
    assert_bits(sc, osc, "padded scores")
    assert_bits(px, okp * 16 + 8, "padded pixels")
    # branch C on a 14 x 14 grid with the default K = 500 (input_size 224): constant map, nothing above the threshold
    rng = np.random.Generator(np.random.PCG64(5))
    flat = np.stack([np.full((14, 14), 0.05, np.float32), rng.random((14, 14)).astype(np.float32) * np.float32(0.09)])
    kp, sc, idx, px, st = _select(T, hip, flat, 500)
    okp, osc, oidx, ost = ora.select_keypoints(flat, 500, 2, 0.5)
    assert np.array_equal(st, ost) and st.tolist() == [1, 1]
    assert np.array_equal(idx, oidx) and np.array_equal(kp, okp)
    assert_bits(sc, osc, "padded scores (branch C)")


# ---------------------------------------------------------------------------------------------------- A6 / A7
def _packed_refiner(T, hip, sd):
    return dev(T, hip.pack_refiner(ora.refiner_weight_list(sd, 2), 2))


def test_gather_and_refine_separately(T, hip):
    g = gold("gather_refine")
    sd = synth.refiner_state(0)
    feat = ora.bn_tokens(synth.tokens(1, 28))[0].reshape(1, 28, 28, 384)
    kq = g["frac_kp"][None]                                       # fractional, border and out-of-range coordinates
    samp = hip.gather(dev(T, feat), dev(T, kq)).cpu().numpy()
    assert_bits(samp, ora.gather(feat, kq), "gather")
    x = g["mlp_in"]
    out = hip.refine(dev(T, x), _packed_refiner(T, hip, sd), 2).cpu().numpy()
    assert_bits(out, ora.refine(x, sd), "refine")
    assert np.abs(out - g["mlp_out"]).max() < 5e-6                 # and the reference's own output


@pytest.mark.parametrize("grid,K,frames", [(28, 500, 3), (40, 1024, 1), (28, 37, 2)])
def test_gather_refine_fused(T, hip, grid, K, frames):
    sd = synth.refiner_state(0)
    feat = ora.bn_tokens(synth.tokens(30 + grid, grid, frames))[0].reshape(frames, grid, grid, 384)
    sal = ora.selector_saliency(feat, synth.selector_state(0))
    kp, _, idx, _ = ora.select_keypoints(sal, K)
    desc = hip.gather_refine(dev(T, feat), dev(T, kp), _packed_refiner(T, hip, sd), 2).cpu().numpy()
    assert_bits(desc, ora.refine(ora.gather(feat, kp), sd), "descriptors")


def test_descriptors_match_reference_golden(T, hip):
    g, s = gold("gather_refine"), gold("selector")
    feat = ora.bn_tokens(synth.tokens(1, 28))[0].reshape(1, 28, 28, 384)
    desc = hip.gather_refine(dev(T, feat), dev(T, s["g28_kp"][None]), _packed_refiner(T, hip, synth.refiner_state(0)), 2)
    assert np.abs(desc.cpu().numpy()[0] - g["g28_desc"]).max() < 5e-6


# ---------------------------------------------------------------------------------------------------- A0 / A9
@pytest.mark.parametrize("h,w,size", [(480, 640, 448), (480, 640, 640), (960, 1280, 960), (231, 517, 112)])
def test_preprocess_and_intensity(T, hip, h, w, size):
    imgs = np.stack([synth.image(40 + i, h, w) for i in range(2)])
    th = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(w, size, False))
    tv = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(h, size, False))
    out = hip.preprocess_u8(dev(T, imgs), size, th, tv).cpu().numpy()
    for i in range(2):
        _, chw = ora.resize_rgb(imgs[i], size)
        assert_bits(out[i], chw, "preprocess")
    # the same arithmetic written as the ViT's patch-embedding operand (bf16 rows of 768 per 16 x 16 patch, k = c*256 + ky*16 + kx)
    if size % 16 == 0:
        from oracle.ora_bf16 import bf16_round
        pt = hip.preprocess_u8_patches(dev(T, imgs), size, th, tv)
        g16 = size // 16
        want = out.reshape(2, 3, g16, 16, g16, 16).transpose(0, 2, 4, 1, 3, 5).reshape(2, g16 * g16, 768)
        if th[2] > 7:
            assert pt is None          # more than 7 horizontal taps: the tiled kernel does not cover the ratio (callers take the fp32 image)
        else:
            assert pt is not None and pt.shape == (2, g16 * g16, 768) and pt.dtype == T.bfloat16
            assert_bits(pt.float().cpu().numpy(), bf16_round(want), "patch rows")
    # a frame batch whose BASE is not dword-aligned (a view at an odd byte offset): the generic kernel serves it, same bits
    raw = T.empty(imgs.size + 3, dtype=T.uint8, device="cuda")
    for off in (1, 2):
        view = raw[off:off + imgs.size].view(imgs.shape)
        view.copy_(dev(T, imgs))
        assert view.data_ptr() % 4 == off
        assert_bits(hip.preprocess_u8(view, size, th, tv).cpu().numpy(), out, "preprocess, unaligned base")
    th = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(w, size, True))
    tv = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(h, size, True))
    rng = np.random.Generator(np.random.PCG64(size))
    kp = (rng.random((2, 300, 2)) * (size + 8) - 4).astype(np.float32)
    kp[:, :50] = np.floor(kp[:, :50]) + 0.5                       # half-way cases: round-half-even
    got = hip.keypoint_intensity(dev(T, imgs), size, th, tv, dev(T, kp)).cpu().numpy()
    for i in range(2):
        assert_bits(got[i], ora.intensity(imgs[i], size, kp[i]), "intensity")


def test_preprocess_matches_pillow_golden(T, hip):
    g = gold("preprocess")
    img = synth.image(0, 480, 640)
    th = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(640, 448, False))
    tv = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(480, 448, False))
    out = hip.preprocess_u8(dev(T, img[None]), 448, th, tv).cpu().numpy()[0]
    assert_bits(out[:, ::16], g["vga448_chw_rows"], "Pillow bilinear + ToTensor + Normalize")


# --------------------------------------------------------------------------------------------------------- M1
def _pair(seed, n, m, dup, noise=0.25):
    from test_oracle_golden import _pair as p
    return p(seed, n, m, dup, noise)


def _match(T, hip, d1, d2, s1, s2, sw=0.3, ms=0.2, md=0.7, i1=None, i2=None, mi=0.1):
    n, m = d1.shape[0], d2.shape[0]
    D1, D2 = dev(T, d1), dev(T, d2)
    nn12, s12, nn21, _, _ = hip.sim_argmax(D1, 0, n, D2, 0, m, 1)
    I1 = None if i1 is None else dev(T, i1)
    I2 = None if i2 is None else dev(T, i2)
    mt, q, cnt = hip.match_finalize(nn12, s12, nn21, n, m, 1, dev(T, s1), 0, dev(T, s2), 0, I1, I2, 1.0 - sw, sw, ms, md, mi)
    c = int(cnt.cpu()[0])
    return mt.cpu().numpy()[0, :c], q.cpu().numpy()[0, :c]


@pytest.mark.parametrize("tag", ["p500", "p500x480", "p1024", "p2048", "p33x70"])
def test_match_with_quality(T, hip, tag):
    from test_oracle_golden import RUNS
    g = gold("matchers")
    seed, n, m, dup = (int(v) for v in g[f"{tag}_spec"])
    d1, d2, s1, s2, i1, i2 = _pair(seed, n, m, dup)
    for rtag, kwf in RUNS.items():
        kw = kwf(i1, i2)
        mt, q = _match(T, hip, d1, d2, s1, s2, kw.get("saliency_weight", 0.3), kw.get("min_saliency", 0.2),
                       kw.get("min_descriptor_sim", 0.7), kw.get("intensity1"), kw.get("intensity2"),
                       kw.get("min_intensity", 0.1))
        omt, oq = ora.match_with_quality(d1, d2, s1, s2, **kw)
        assert np.array_equal(mt, omt) and mt.dtype == np.int64, (tag, rtag)
        assert_bits(q, oq, (tag, rtag))
        assert np.array_equal(mt, g[f"{tag}_{rtag}_matches"]), (tag, rtag)      # == the reference's match pairs


def test_batched_pairs_and_strides(T, hip):
    """N frames -> N-1 consecutive pairs in one launch, descriptors addressed by stride (the pipeline's layout)."""
    N, K = 6, 500
    desc = np.stack([synth.unit_descriptors(60 + i, K, 128, dup=25) for i in range(N)])
    sc = np.stack([np.random.Generator(np.random.PCG64(i)).random(K).astype(np.float32) for i in range(N)])
    D, S = dev(T, desc), dev(T, sc)
    nn12, s12, nn21, _, _ = hip.sim_argmax(D, K * 128, K, D[1:], K * 128, K, N - 1)
    mt, q, cnt = hip.match_finalize(nn12, s12, nn21, K, K, N - 1, S, K, S[1:], K, None, None, 0.7, 0.3, 0.0, -1.0, 0.0)
    for p in range(N - 1):
        omt, oq = ora.match_with_quality(desc[p], desc[p + 1], sc[p], sc[p + 1], 0.3, 0.0, -1.0)
        c = int(cnt.cpu()[p])
        assert np.array_equal(mt.cpu().numpy()[p, :c], omt)
        assert_bits(q.cpu().numpy()[p, :c], oq, f"pair {p}")


@pytest.mark.parametrize("variant", ["1", "2"])
@pytest.mark.parametrize("n,m,dup", [(500, 500, 40), (33, 70, 5), (700, 129, 60), (128, 1, 0), (1, 300, 0), (1024, 1000, 100)])
def test_sim_argmax_both_forms(T, hip, variant, n, m, dup, knob):
    """SSLAM_M1_VARIANT=1: S per direction; =2: S once + 64-bit key reduction for the column direction (the form batched
    calls use).  Both must give the oracle's first-maximum indices (duplicated descriptors = exact ties) and values."""
    knob("SSLAM_M1_VARIANT", variant)
    d1, d2, *_ = _pair(900 + n + m, n, m, dup)
    pairs = 3
    D1 = dev(T, np.stack([d1, d1[::-1], d1]))
    D2 = dev(T, np.stack([d2, d2, d2[::-1]]))
    nn12, s12, nn21, s21, sec = hip.sim_argmax(D1, n * 128, n, D2, m * 128, m, pairs, want_s21=True, want_second=True)
    for p, (a, b) in enumerate([(d1, d2), (d1[::-1], d2), (d1, d2[::-1])]):
        o12, os12, o21, os21 = ora.sim_argmax(np.ascontiguousarray(a), np.ascontiguousarray(b))
        assert np.array_equal(nn12.cpu().numpy()[p], o12) and np.array_equal(nn21.cpu().numpy()[p], o21), (variant, p)
        assert_bits(s12.cpu().numpy()[p], os12, "s12")
        assert_bits(s21.cpu().numpy()[p], os21, "s21")
        S = ora.sim_matrix(np.ascontiguousarray(a), np.ascontiguousarray(b))
        S[np.arange(n), o12] = -np.inf
        assert_bits(sec.cpu().numpy()[p], S.max(1), "second12")


def test_wide_reference_goldens_on_gpu(T, hip):
    """The wide randomised fixtures (reference outputs, tests/golden/make_golden_wide.py) straight against the HIP kernels:
    64 selection cases and 32 matcher cases - indices / match pairs equal the reference's."""
    g = gold("select_wide")
    for s in range(int(g["count"])):
        m, K, radius, pct = synth.wide_map(s)
        kp, sc, idx, px, st = _select(T, hip, m[None], K, radius, pct)
        want = g[f"s{s}_idx"]
        if want.size == 0:
            assert st[0] == 1, s                       # the reference raises here
            continue
        assert st[0] == 0 and np.array_equal(idx[0], want.astype(np.int32)), (s, m.shape, K, radius, pct)
        assert_bits(sc[0], g[f"s{s}_scores"], f"scores {s}")
    g = gold("match_wide")
    for s in range(int(g["count"])):
        d1, d2, s1, s2, kw = synth.wide_pair(s)
        mt, q = _match(T, hip, d1, d2, s1, s2, kw["saliency_weight"], kw["min_saliency"], kw["min_descriptor_sim"],
                       kw.get("intensity1"), kw.get("intensity2"), kw["min_intensity"])
        assert np.array_equal(mt, g[f"p{s}_matches"].astype(np.int64).reshape(-1, 2)), s
        assert np.abs(q - g[f"p{s}_quality"]).max(initial=0.0) < 1e-6, s


# ------------------------------------------------------------------------------------------------ end to end
def test_end_to_end_golden(T, hip):
    """tokens + images -> keypoints, descriptors, intensity, matches: indices / pairs equal the reference's."""
    g = gold("e2e")
    toks, imgs = synth.token_sequence(3, 28), synth.image_sequence(3)
    ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
    ones, zeros = dev(T, np.ones(384, np.float32)), dev(T, np.zeros(384, np.float32))
    feat, _, _ = hip.bn_tokens(dev(T, toks), 5, 1, ones, zeros, zeros, ones, True, 1e-5)
    feat = feat.view(3, 28, 28, 384)
    w1p, b1, w2, b2, hs = _packed_selector(T, hip, ssd)
    sal = hip.selector_saliency(feat, w1p, b1, w2, b2, hs)
    kp, sc, idx, px, st = hip.select_keypoints(sal, 500)
    desc = hip.gather_refine(feat, kp, _packed_refiner(T, hip, rsd), 2)
    th = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(640, 448, True))
    tv = tuple(dev(T, a) if isinstance(a, np.ndarray) else a for a in hip.resample_table(480, 448, True))
    inten = hip.keypoint_intensity(dev(T, imgs), 448, th, tv, px)
    for i in range(3):
        assert np.array_equal(kp.cpu().numpy()[i], g[f"f{i}_kp"])
        assert np.array_equal(inten.cpu().numpy()[i], g[f"f{i}_intensity"])
        assert np.abs(desc.cpu().numpy()[i, ::5] - g[f"f{i}_desc_sub"]).max() < 1e-5
    nn12, s12, nn21, _, _ = hip.sim_argmax(desc, 500 * 128, 500, desc[1:], 500 * 128, 500, 2)
    mt, q, cnt = hip.match_finalize(nn12, s12, nn21, 500, 500, 2, sc, 500, sc[1:], 500, inten, inten[1:], 0.7, 0.3, 0.5, 0.7, 0.15)
    for p, (a, b) in enumerate([(0, 1), (1, 2)]):
        c = int(cnt.cpu()[p])
        assert np.array_equal(mt.cpu().numpy()[p, :c], g[f"pair{a}{b}_matches"])
        assert np.abs(q.cpu().numpy()[p, :c] - g[f"pair{a}{b}_quality"]).max() < 1e-5
    assert hip.launch_count() > 0


@pytest.mark.parametrize("tag", ["e2e_g40", "e2e_g60"])
def test_end_to_end_golden_larger_grids(T, hip, tag):
    """BASELINE configs[2] / configs[4] shapes against the reference's own chain (8 frames at G = 40 / K = 1024, 16 at G = 60 /
    K = 2048; tests/golden/make_golden_e2e_grids.py): keypoint sets, order up to near-tie swaps, scores / descriptors /
    intensities by cell, match pairs as indices where both frames are index-exact and as (cell, cell) pairs always - the bars
    of tests/e2e_check.py, the same the CPU oracle is held to - and the HIP path equal to the oracle bit for bit."""
    import e2e_check
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    g = e2e_check.gold(tag)
    grid, K, n = int(g["grid"]), int(g["K"]), int(g["n_frames"])
    toks = synth.token_sequence(n, grid)
    imgs = synth.image_sequence(n, int(g["height"]), int(g["width"]))
    pipe = SequencePipeline(ExtractorConfig(input_size=16 * grid, num_keypoints=K), synth.selector_state(0), synth.refiner_state(0))
    ex = pipe.extract(dev(T, toks), dev(T, imgs))
    idx, sc = ex["idx"].cpu().numpy(), ex["scores"].cpu().numpy()
    desc, inten = ex["descriptors"].cpu().numpy(), ex["intensity"].cpu().numpy()
    assert not ex["status"].cpu().numpy().any()

    def match(a, b):
        sel = T.tensor([a, b], device="cuda")
        m = pipe.match(ex["descriptors"][sel], ex["scores"][sel], ex["intensity"][sel], spacing=1)
        c = int(m["match_count"].cpu()[0])
        return m["matches"].cpu().numpy()[0, :c], m["quality"].cpu().numpy()[0, :c]
    rep = e2e_check.check_sequence(tag, idx, sc, desc, inten, match)
    assert len(rep["pairs"]) >= 8
    # and the same frames through the oracle: bit-identical (so the order report of the CPU suite is the GPU's too)
    feat = ora.bn_tokens(toks)[0].reshape(n, grid, grid, 384)
    okp, osc, oidx, _ = ora.select_keypoints(ora.selector_saliency(feat, synth.selector_state(0)), K)
    assert np.array_equal(idx, oidx)
    assert_bits(sc, osc, "scores")
    assert_bits(desc, ora.refine(ora.gather(feat, okp), synth.refiner_state(0)), "descriptors")
    # all spacing-1 pairs in ONE batched launch equal the per-pair calls above
    m = pipe.match(ex["descriptors"], ex["scores"], ex["intensity"], spacing=1)
    for p in range(n - 1):
        c = int(m["match_count"].cpu()[p])
        want = g[f"pair_{p}_{p + 1}_matches"].astype(np.int64)
        got_c = e2e_check.cell_pairs(m["matches"].cpu().numpy()[p, :c], idx[p], idx[p + 1])
        want_c = e2e_check.cell_pairs(want, g[f"f{p}_idx"], g[f"f{p + 1}_idx"])
        assert np.array_equal(got_c[np.lexsort(got_c.T[::-1])], want_c[np.lexsort(want_c.T[::-1])]), (tag, p)


def test_bn_tokens_batch_of_four_golden(T, hip):
    """B = 4 as train.py:300-302 calls the backbone: one set of statistics over 4 x 784 tokens (reference output)."""
    g = gold("bn_tokens_b4")
    tok = synth.tokens(int(g["frame"]), 28, batch=4)
    ones, zeros = dev(T, np.ones(384, np.float32)), dev(T, np.zeros(384, np.float32))
    for train, key in [(True, "train_b4_sub"), (False, "eval_b4_sub")]:
        y, _, _ = hip.bn_tokens(dev(T, tok), 5, 4, ones, zeros, zeros, ones, train, 1e-5)
        assert_bits(y.cpu().numpy(), ora.bn_tokens(tok, group=4, train=train)[0], "bn group of 4")
        assert np.abs(y.cpu().numpy()[:, ::int(g["sub_step"])] - g[key]).max() < 5e-6


# ------------------------------------------------------------------------- the matcher module (M1..M5 signatures)
def test_matching_module_m1_to_m5(T, hip):
    import matching
    g = gold("matchers")
    for tag in ["p500", "p500x480", "p1024", "p33x70"]:
        seed, n, m, dup = (int(v) for v in g[f"{tag}_spec"])
        d1, d2, s1, s2, i1, i2 = _pair(seed, n, m, dup)
        mt, q = matching.match_with_quality(d1, d2, s1, s2, saliency_weight=0.3, min_saliency=0.5, min_descriptor_sim=0.7,
                                            intensity1=i1, intensity2=i2, min_intensity=0.15)              # M1
        assert mt.dtype == np.int64 and q.dtype == np.float32
        assert np.array_equal(mt, g[f"{tag}_cli_matches"]) and np.abs(q - g[f"{tag}_cli_quality"]).max() < 1e-6
        mt, q = matching.match_with_quality(d1, d2, s1, s2, min_descriptor_sim=2.0)
        assert mt.shape == (0, 2) and mt.dtype == np.int64 and q.shape == (0,) and q.dtype == np.float32
        m2 = matching.find_matches(d1, d2, ratio_thresh=0.8)                                              # M2
        om2 = ora.find_matches_m2(d1, d2, 0.8)
        assert [(a, b) for a, b, _ in m2] == [(a, b) for a, b, _ in om2]
        assert_bits(np.array([c for *_, c in m2], np.float32), np.array([c for *_, c in om2], np.float32), "m2 sims")
        assert np.array_equal(np.array([(a, b) for a, b, _ in m2], np.int64).reshape(-1, 2), g[f"{tag}_m2_ij"])
        m4, dist = matching.find_mutual_nearest_neighbors(d1, d2, 0.9)                                      # M4
        om4, odist = ora.find_mnn_m4(d1, d2, 0.9)
        assert np.array_equal(m4, om4) and np.array_equal(m4, g[f"{tag}_m4_matches"])
        assert_bits(dist, odist, "m4 distances")
        assert matching.count_tracked(d1, d2, 0.8) == int(g[f"{tag}_m5_count"])                             # M5
    b1, b2 = [], []
    for seed, noise in zip(g["m3_seeds"], g["m3_noise"]):
        d1, d2, *_ = _pair(int(seed), 200, 200, 10, float(noise))
        b1.append(d1)
        b2.append(d2)
    out = matching.find_matches_batched(T.from_numpy(np.stack(b1)).cuda(), T.from_numpy(np.stack(b2)).cuda())  # M3
    assert out.dtype == T.int64 and np.array_equal(out.cpu().numpy(), g["m3_matches"])


def test_dropin_modules_on_gpu(T, hip):
    """The reference's call sequence (visualize_matches_sequence.py:69-85) through the drop-in nn.Modules on cuda."""
    from models.descriptor_refiner import DescriptorRefiner
    from models.dino_backbone import DinoBackbone
    from models.keypoint_selector import KeypointSelector
    from test_models_api import TokenDino
    g = gold("e2e")
    toks = synth.token_sequence(3, 28)
    bb = DinoBackbone(input_size=448, freeze=True, dino=TokenDino(), vit_precision="eager").cuda()    # a token stand-in: nothing to convert
    sel = KeypointSelector(384, 256).cuda()
    ref = DescriptorRefiner(384, 384, 128).cuda()
    sel.load_state_dict({k: T.from_numpy(v) for k, v in synth.selector_state(0).items()})
    ref.load_state_dict({k: T.from_numpy(v) for k, v in synth.refiner_state(0).items()})
    sel.eval()
    ref.eval()
    before = hip.launch_count()
    with T.no_grad():
        for i in range(3):
            bb.dino.tokens = T.from_numpy(toks[i:i + 1]).cuda()
            f = bb(T.zeros(1, 3, 448, 448, device="cuda"))
            sal = sel(f)
            kp, sc = sel.select_keypoints(sal, num_keypoints=500)
            desc = ref(bb.extract_at_keypoints(f, kp))
            pix = bb.patch_to_pixel(kp)
            assert f.shape == (1, 28, 28, 384) and sal.shape == (1, 28, 28, 1) and desc.shape == (1, 500, 128)
            assert np.array_equal(kp[0].cpu().numpy(), g[f"f{i}_kp"])
            assert np.abs(sc[0].cpu().numpy() - g[f"f{i}_scores"]).max() < 5e-6
            assert np.abs(desc[0, ::5].cpu().numpy() - g[f"f{i}_desc_sub"]).max() < 1e-5
            assert np.array_equal(pix[0].cpu().numpy(), g[f"f{i}_kp"] * 16 + 8)
    assert hip.launch_count() >= before + 3 * 5, "the HIP kernels must have served these calls"
    assert int(bb.feature_norm.num_batches_tracked) == 3
    bb.eval()
    with T.no_grad():
        bb.dino.tokens = T.from_numpy(toks[:2]).cuda()
        f = bb(T.zeros(2, 3, 448, 448, device="cuda"))
    rm, rv = bb.feature_norm.running_mean.cpu().numpy(), bb.feature_norm.running_var.cpu().numpy()
    want = ora.bn_tokens(toks[:2], 5, 2, run_mean=rm, run_var=rv, train=False)[0]
    assert_bits(f.cpu().numpy().reshape(2, 784, 384), want, "eval-mode BN with the tracked running stats")


def test_pipeline_with_hip_vit_end_to_end(T, hip):
    """images -> A0 -> HIP ViT (A1) -> A2..A9 -> M1 in one call; and the drop-in DinoBackbone uses the HIP ViT too."""
    from models.dino_backbone import DinoBackbone
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    from sslam_amd.vit import DinoV3ViT
    T.manual_seed(3)
    vit = DinoV3ViT().cuda().eval()
    imgs = T.from_numpy(synth.image_sequence(5)).cuda()
    pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cuda", vit=vit)
    out = pipe.run(imgs)
    assert out["descriptors"].shape == (5, 500, 128) and out["match_count"].shape == (4,)
    assert int(out["status"].sum()) == 0
    # the stages after the ViT are bit-exact given its tokens
    tok = pipe.tokens_from_images(imgs).cpu().numpy()
    feat = ora.bn_tokens(tok)[0].reshape(5, 28, 28, 384)
    _, _, oidx, _ = ora.select_keypoints(ora.selector_saliency(feat, synth.selector_state(0)), 500)
    assert np.array_equal(out["idx"].cpu().numpy(), oidx)
    # tokens agree with the fp32 torch evaluation of the same weights at bf16-operand tolerance
    with T.no_grad():
        want = vit.forward_features(pipe.preprocess(imgs))
    rel = float((T.from_numpy(tok).cuda() - want).norm() / want.norm())
    assert rel < 2.5e-2, rel
    bb = DinoBackbone(input_size=448, dino=vit).cuda()
    before = hip.launch_count()
    with T.no_grad():
        f = bb(pipe.preprocess(imgs[:2]))
    assert f.shape == (2, 28, 28, 384) and hip.launch_count() >= before + 60      # 12 layers x (5 or 6) launches

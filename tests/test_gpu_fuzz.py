"""Seeded differential fuzzing of the order-sensitive HIP kernels against the CPU oracle: many random shapes, plateau /
tie patterns and thresholds per test (bit-exact bar, as in test_gpu_parity.py).  The fixed cases there follow the
reference's own fixtures; these look for what nobody thought of - ragged sizes, quantised saliency maps with massive
ties, tiny grids, K near the number of cells, descriptor sets with repeated rows in both frames.
"""
import numpy as np
import pytest

import synth
from oracle import ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def hip(T):
    from sslam_amd import lib
    lib.lib()
    return lib


def dev(T, a):
    return T.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_select_keypoints(T, hip, seed):
    rng = np.random.Generator(np.random.PCG64(4242 + seed))
    for case in range(8):
        g = int(rng.integers(3, 65))                # the kernel holds one frame in LDS: G <= 64
        frames = int(rng.integers(1, 5))
        cells = g * g
        K = int(rng.integers(1, min(cells, 4096) + 1))
        radius = int(rng.integers(0, 5))
        pct = float(rng.choice([0.0, 0.1, 0.5, 0.73, 0.9, 1.0]))
        kind = case % 4
        sal = rng.random((frames, g, g)).astype(np.float32)
        if kind == 1:                                   # heavily quantised: plateaus and exact ties everywhere
            sal = (np.floor(sal * rng.integers(2, 9)) / 8).astype(np.float32)
        elif kind == 2:                                 # smooth bumps: few, wide maxima
            yy, xx = np.mgrid[0:g, 0:g].astype(np.float32)
            sal = np.stack([np.sin(xx * rng.random() + f) * np.cos(yy * rng.random() - f) * 0.5 + 0.5 for f in range(frames)]).astype(np.float32)
        elif kind == 3:                                 # sigmoid-like band hugging 0.5 (what an untrained selector emits)
            sal = (0.5 + 0.01 * (sal - 0.5)).astype(np.float32)
        kp, sc, idx, px, st = hip.select_keypoints(dev(T, sal), K, radius, pct)
        okp, osc, oidx, ost = ora.select_keypoints(sal, K, radius, pct)
        tag = (seed, case, g, frames, K, radius, pct, kind)
        assert np.array_equal(st.cpu().numpy(), ost), tag
        ok = ost == 0                                   # frames whose request the grid cannot supply only carry the flag
        assert np.array_equal(idx.cpu().numpy()[ok], oidx[ok]), tag
        assert np.array_equal(bits(kp.cpu().numpy()[ok]), bits(okp[ok])), tag
        assert np.array_equal(bits(sc.cpu().numpy()[ok]), bits(osc[ok])), tag
        assert np.array_equal(bits(px.cpu().numpy()[ok]), bits(ora.patch_to_pixel(okp[ok]))), tag


@pytest.mark.parametrize("variant", ["1", "2"])
@pytest.mark.parametrize("seed", range(4))
def test_fuzz_sim_argmax_and_match(T, hip, seed, variant, knob):
    knob("SSLAM_M1_VARIANT", variant)
    rng = np.random.Generator(np.random.PCG64(777 + seed))
    for case in range(6):
        n, m = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        d1 = synth.unit_descriptors(1000 * seed + case, n, 128, dup=int(rng.integers(0, max(1, n // 3))) if n > 3 else 0)
        perm = rng.integers(0, n, size=m)
        d2 = d1[perm].copy()                            # exact copies -> similarities of exactly 1 and many exact ties
        noisy = rng.random(m) < 0.6
        d2[noisy] += (0.3 * rng.standard_normal((int(noisy.sum()), 128)) / np.sqrt(128)).astype(np.float32)
        d2 = (d2 / np.linalg.norm(d2.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
        s1, s2 = rng.random(n).astype(np.float32), rng.random(m).astype(np.float32)
        i1, i2 = rng.random(n).astype(np.float32), rng.random(m).astype(np.float32)
        nn12, s12, nn21, s21, sec = hip.sim_argmax(dev(T, d1), 0, n, dev(T, d2), 0, m, 1, want_s21=True, want_second=True)
        o12, os12, o21, os21 = ora.sim_argmax(d1, d2)
        tag = (seed, case, n, m, variant)
        assert np.array_equal(nn12.cpu().numpy()[0], o12) and np.array_equal(nn21.cpu().numpy()[0], o21), tag
        assert np.array_equal(bits(s12.cpu().numpy()[0]), bits(os12)) and np.array_equal(bits(s21.cpu().numpy()[0]), bits(os21)), tag
        kw = dict(saliency_weight=float(rng.random()), min_saliency=float(rng.random() * 0.6),
                  min_descriptor_sim=float(rng.random()), intensity1=i1, intensity2=i2, min_intensity=float(rng.random() * 0.5))
        mt, q, cnt = hip.match_finalize(nn12, s12, nn21, n, m, 1, dev(T, s1), 0, dev(T, s2), 0, dev(T, i1), dev(T, i2),
                                        1.0 - kw["saliency_weight"], kw["saliency_weight"], kw["min_saliency"],
                                        kw["min_descriptor_sim"], kw["min_intensity"])
        omt, oq = ora.match_with_quality(d1, d2, s1, s2, **kw)
        c = int(cnt.cpu()[0])
        assert c == len(omt) and np.array_equal(mt.cpu().numpy()[0, :c], omt), tag
        assert np.array_equal(bits(q.cpu().numpy()[0, :c]), bits(oq)), tag


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_gather_and_refine(T, hip, seed):
    """Fractional and out-of-range keypoint coordinates (grid_sample zero padding), ragged row counts (tile tails)."""
    rng = np.random.Generator(np.random.PCG64(31 + seed))
    sd = synth.refiner_state(seed)
    packed = dev(T, hip.pack_refiner(ora.refiner_weight_list(sd, 2), 2))
    for case in range(4):
        g = int(rng.integers(2, 41))
        frames = int(rng.integers(1, 4))
        K = int(rng.integers(1, 300))
        feat = rng.standard_normal((frames, g, g, 384)).astype(np.float32)
        kp = (rng.random((frames, K, 2)) * (g + 3) - 2).astype(np.float32)        # [-2, g+1): some taps fall outside
        kp[:, ::7] = np.floor(kp[:, ::7])                                           # integer coordinates
        got_x = hip.gather(dev(T, feat), dev(T, kp)).cpu().numpy()
        want_x = ora.gather(feat, kp)
        assert np.array_equal(bits(got_x), bits(want_x)), (seed, case, g, frames, K)
        got = hip.gather_refine(dev(T, feat), dev(T, kp), packed, 2).cpu().numpy()
        assert np.array_equal(bits(got), bits(ora.refine(want_x, sd))), (seed, case, g, frames, K)


@pytest.mark.parametrize("seed", range(2))
def test_fuzz_preprocess_sizes(T, hip, seed):
    """Random image / target sizes through both the fast and the generic resampling kernels, against the oracle
    (Pillow's fixed-point arithmetic, itself pinned against Pillow by the golden fixtures)."""
    rng = np.random.Generator(np.random.PCG64(99 + seed))
    for case in range(5):
        h, w = int(rng.integers(40, 700)), int(rng.integers(40, 900))
        size = int(rng.choice([64, 112, 224, 320, 448]))
        n = int(rng.integers(1, 3))
        img = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
        tabs = []
        ok = True
        for n_in in (w, h):
            try:
                b, c, k = hip.resample_table(n_in, size, False)
            except Exception:
                ok = False
                break
            tabs.append((dev(T, b), dev(T, c), k))
        if not ok:
            continue
        try:
            out = hip.preprocess_u8(dev(T, img), size, tabs[0], tabs[1]).cpu().numpy()
        except hip.SslamHipError:
            continue                                   # shapes the kernel declines (too many taps / rows per tile) are loud, not wrong
        for i in range(n):
            _, chw = ora.resize_rgb(img[i], size)
            assert np.array_equal(bits(out[i]), bits(chw)), (seed, case, h, w, size)

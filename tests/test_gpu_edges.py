"""Degenerate inputs on the HIP path, against the oracle: a one-frame sequence (no pairs), constant tokens (zero BatchNorm
variance, a saliency plateau over the whole grid: every NMS tie survives, every descriptor is the same and every similarity an
exact tie), black / white frames (intensity thresholds), and a sequence whose frames are all identical (every keypoint matches
itself).  The reference has no tests for these; its behaviour on them is what the oracle restates (first maximum wins,
`nms == pooled` keeps plateaus: keypoint_selector.py:209-226, visualize_matches_sequence.py:147-152)."""
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
def pipe(T):
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    return SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cuda")


def _oracle(pipe, toks, imgs):
    cfg = pipe.cfg
    n, g, K = toks.shape[0], cfg.grid, cfg.num_keypoints
    ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
    feat = ora.bn_tokens(toks)[0].reshape(n, g, g, 384)
    kp, sc, idx, _ = ora.select_keypoints(ora.selector_saliency(feat, ssd), K)
    desc = ora.refine(ora.gather(feat, kp), rsd)
    inten = np.stack([ora.intensity(imgs[i], cfg.input_size, ora.patch_to_pixel(kp[i])) for i in range(n)])
    pairs = [ora.match_with_quality(desc[i], desc[i + 1], sc[i], sc[i + 1], cfg.saliency_weight, cfg.min_saliency,
                                    cfg.min_descriptor_sim, inten[i], inten[i + 1], cfg.min_intensity) for i in range(n - 1)]
    return idx, desc, inten, pairs


def _check(T, pipe, toks, imgs, tag):
    out = pipe.run(T.from_numpy(imgs).cuda(), T.from_numpy(toks).cuda())
    idx, desc, inten, pairs = _oracle(pipe, toks, imgs)
    assert np.array_equal(out["idx"].cpu().numpy(), idx), tag
    assert np.array_equal(out["descriptors"].cpu().numpy().view(np.uint32), desc.view(np.uint32)), tag
    assert np.array_equal(out["intensity"].cpu().numpy().view(np.uint32), inten.view(np.uint32)), tag
    assert out["matches"].shape[0] == len(pairs) == toks.shape[0] - 1
    for p, (omt, oq) in enumerate(pairs):
        c = int(out["match_count"][p])
        assert c == len(omt) and np.array_equal(out["matches"][p, :c].cpu().numpy(), omt), (tag, p)
        assert np.array_equal(out["quality"][p, :c].cpu().numpy().view(np.uint32), oq.view(np.uint32)), (tag, p)
    return out, pairs


def test_one_frame_sequence_has_no_pairs(T, pipe):
    out, pairs = _check(T, pipe, synth.token_sequence(1, 28), synth.image_sequence(1), "one frame")
    assert out["matches"].shape == (0, 500, 2) and out["match_count"].shape == (0,) and pairs == []


def test_constant_tokens_plateau_and_exact_ties(T, pipe):
    toks = np.full((2, 789, 384), 0.75, np.float32)          # zero variance per channel: BatchNorm output 0 everywhere
    toks[1] = -3.0
    out, pairs = _check(T, pipe, toks, synth.image_sequence(2), "constant tokens")
    d = out["descriptors"].cpu().numpy()
    assert np.array_equal(d[0], np.broadcast_to(d[0, :1], d[0].shape))          # one descriptor, 500 times
    # every similarity is the same number: the first maximum wins in both directions, so (0, 0) is the only mutual pair
    if len(pairs[0][0]):
        assert pairs[0][0].tolist() == [[0, 0]]


@pytest.mark.parametrize("level", [0, 255])
def test_black_and_white_frames(T, pipe, level):
    imgs = np.full((3, 480, 640, 3), level, np.uint8)
    out, _ = _check(T, pipe, synth.token_sequence(3, 28), imgs, f"level {level}")
    want = np.float32(level / 255.0)
    assert np.all(out["intensity"].cpu().numpy() == want)
    if level == 0:
        assert int(out["match_count"].sum()) == 0              # below min_intensity: every candidate is dropped


def test_identical_frames_match_themselves(T, pipe):
    toks = np.repeat(synth.token_sequence(1, 28), 4, axis=0)
    imgs = np.repeat(synth.image_sequence(1), 4, axis=0)
    out, pairs = _check(T, pipe, toks, imgs, "identical frames")
    for p in range(3):
        c = int(out["match_count"][p])
        m = out["matches"][p, :c].cpu().numpy()
        # a keypoint's best partner is the FIRST keypoint with its descriptor: itself unless an earlier duplicate exists
        assert np.all(m[:, 1] <= m[:, 0])
        assert T.equal(out["matches"][p], out["matches"][0]) and T.equal(out["quality"][p], out["quality"][0])


@pytest.mark.parametrize("variant", ["1", "2"])
def test_non_finite_descriptors_keep_indices_in_range(T, variant, knob):
    """NaN / Inf descriptors are outside the matcher's contract (include/sslam_hip.h: values and the choice among candidates
    unspecified) but must never produce an index outside the arrays: the finalise kernel and the sibling matchers index with
    nn12 / nn21.  Both launch forms, ragged sizes (lanes beyond the last query contribute the key of -inf), whole rows NaN,
    a non-finite ROW 0 (the row those lanes re-read), +-Inf entries; the finite pairs of the same launch stay bit-exact."""
    from sslam_amd import lib
    knob("SSLAM_M1_VARIANT", variant)
    n, m, pairs = 77, 131, 17
    d1 = np.stack([synth.unit_descriptors(800 + p, n, 128) for p in range(pairs)])
    d2 = np.stack([synth.unit_descriptors(900 + p, m, 128) for p in range(pairs)])
    d1[0, 5] = np.nan                      # one NaN row
    d2[1, 0] = np.nan                      # row 0 of the candidate side
    d1[2, 0] = np.inf                      # row 0 of the query side: what lanes beyond nq re-read
    d1[3] = np.nan                         # everything NaN
    d2[4, 7, 3] = -np.inf
    d1[5, :, 0] = np.where(np.arange(n) % 2, np.nan, d1[5, :, 0])
    d2[5, :, 1] = -np.nan
    D1, D2 = T.from_numpy(d1).cuda(), T.from_numpy(d2).cuda()
    ws = T.empty(max(8, lib.workspace_bytes(1, 28, max(n, m), pairs)), dtype=T.uint8, device="cuda")
    nn12, s12, nn21, s21, _ = lib.sim_argmax(D1, n * 128, n, D2, m * 128, m, pairs, want_s21=True, workspace=ws)
    a, b = nn12.cpu().numpy(), nn21.cpu().numpy()
    assert a.min() >= 0 and a.max() < m and b.min() >= 0 and b.max() < n
    sc1, sc2 = T.rand((pairs, n), device="cuda"), T.rand((pairs, m), device="cuda")
    mt, q, cnt = lib.match_finalize(nn12, s12, nn21, n, m, pairs, sc1, n, sc2, m, None, None, 0.7, 0.3, 0.0, -1.0, 0.0)
    T.cuda.synchronize()
    mtc = mt.cpu().numpy()
    assert mtc[..., 0].max() < n and mtc[..., 1].max() < m and mtc.min() >= 0
    for p in range(6, pairs):              # untouched pairs: the oracle's bits
        o12, os12, o21, os21 = ora.sim_argmax(d1[p], d2[p])
        assert np.array_equal(a[p], o12) and np.array_equal(b[p], o21)
        assert np.array_equal(s12.cpu().numpy()[p].view(np.uint32), os12.view(np.uint32))
        assert np.array_equal(s21.cpu().numpy()[p].view(np.uint32), os21.view(np.uint32))
    # arrays that did not come from sslam_sim_argmax: an out-of-range nn12 entry is skipped, not dereferenced
    bad = nn12.clone()
    bad[6, :4] = T.tensor([m, 1 << 30, -1, -(1 << 31)], dtype=bad.dtype, device="cuda")
    mt2, _, cnt2 = lib.match_finalize(bad, s12, nn21, n, m, pairs, sc1, n, sc2, m, None, None, 0.7, 0.3, 0.0, -1.0, 0.0)
    T.cuda.synchronize()
    assert int(cnt2[6]) <= int(cnt[6]) and T.equal(cnt2[7:], cnt[7:])

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

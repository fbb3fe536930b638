"""BASELINE.json's configurations at their FULL per-GPU sizes on the MI355X (the sizes the oracle cannot cover frame by frame in
a unit test): size-independent properties of the whole result, plus the oracle over whole sequences.

  * idempotence: the same pass twice gives the same bits everywhere (no uninitialised slot, no race, no atomics-order effect);
  * independence of the launch structure: the sequence pushed through the streaming scheduler in odd-sized chunks equals the
    one-shot pass (keypoints, descriptors, intensities, matches, quality, counts);
  * structure: every keypoint index inside the grid, patch coordinates consistent with the index,
    unit descriptors, match slots: idx1 strictly ascending, idx2 distinct (a mutual match is injective), slots past the count zero;
  * the CPU oracle, bit for bit (tests/oracle_check.py): EVERY frame and EVERY pair of all four configurations (613 / 2 965 / 647 /
    512 frames: 3 / 28 / 3 / 10 s of oracle time on the box's 16 host threads).
Synthetic sequences are bench.synth_sequence (device-side) - no TUM data exists on the box."""
import pytest

import synth

pytestmark = pytest.mark.gpu

# name -> (frames on one GPU, H, W, input_size, K): bench.WORKLOADS, the multi-GPU configs at their per-GPU share
CONFIGS = {
    "configs[1] fr1/desk": (613, 480, 640, 448, 500),
    "configs[2] fr2/desk 1024 kp": (2965, 480, 640, 640, 1024),
    "configs[3] fr3/long_office, share of one of 4 GPUs": (647, 480, 640, 448, 500),
    "configs[4] 1280x960 2048 kp, share of one of 8 GPUs": (512, 960, 1280, 960, 2048),
}


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_properties_and_spot_checks(T, name):
    import bench
    from sslam_amd.harness import StreamingSequence
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    n, h, w, size, K = CONFIGS[name]
    grid = size // 16
    dev = T.device("cuda")
    cfg = ExtractorConfig(input_size=size, num_keypoints=K)
    ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
    pipe = SequencePipeline(cfg, ssd, rsd, device=dev)
    imgs, toks = bench.synth_sequence(n, 0, n, h, w, grid, dev, seed=1234)
    one = pipe.run(imgs, toks)
    keys = ("idx", "keypoints_patch", "scores", "descriptors", "intensity", "matches", "quality", "match_count", "status")

    # ---- idempotence ------------------------------------------------------------------------------------------------
    two = pipe.run(imgs, toks)
    for k in keys:
        assert T.equal(one[k], two[k]), (name, k)
    del two

    # ---- independent of the launch structure --------------------------------------------------------------------------
    chunk = n // 3 + 7
    st = StreamingSequence(pipe, (cfg.spacing,)).run(toks, imgs, chunk=chunk)
    for k in ("idx", "descriptors", "intensity"):
        assert T.equal(st["frames"][k], one[k]), (name, "chunked", k)
    for k in ("matches", "quality", "match_count"):
        assert T.equal(st[cfg.spacing][k], one[k]), (name, "chunked", k)
    del st

    # ---- structure ----------------------------------------------------------------------------------------------------
    assert int(one["status"].abs().sum()) == 0
    idx = one["idx"].long()
    assert int(idx.min()) >= 0 and int(idx.max()) < grid * grid
    # (indices may repeat inside a frame: branches B / C of select_keypoints pad with the top raw-saliency cells, which can already
    # be among the NMS survivors - keypoint_selector.py:157-199 - so distinctness is NOT a property of the reference)
    kp = one["keypoints_patch"]
    assert T.equal(kp[..., 0], (idx % grid).float()) and T.equal(kp[..., 1], (idx // grid).float())      # (x, y) = (idx % W, idx // W)
    norm = one["descriptors"].double().pow(2).sum(-1).sqrt()
    assert float((norm - 1).abs().max()) < 1e-6
    cnt = one["match_count"].long()
    assert int(cnt.min()) >= 0 and int(cnt.max()) <= K and int(cnt.sum()) > 0
    slot = T.arange(K, device=dev)[None, :]
    valid = slot < cnt[:, None]
    m1, m2 = one["matches"][..., 0], one["matches"][..., 1]
    assert int(m1[~valid].abs().sum()) == 0 and int(m2[~valid].abs().sum()) == 0 and float(one["quality"][~valid].abs().sum()) == 0.0
    asc = (m1[:, 1:] > m1[:, :-1]) | ~valid[:, 1:]
    assert bool(asc.all()), "idx1 not strictly ascending inside a pair"
    # idx2 distinct among a pair's valid slots: push the invalid slots to distinct values beyond K, sort, compare neighbours
    m2v = T.where(valid, m2, K + slot.expand_as(m2))
    s2 = m2v.sort(dim=1).values
    assert bool((s2[:, 1:] != s2[:, :-1]).all()), "two matches of a pair share idx2"

    # ---- the oracle: every frame and every pair of every configuration (2 965 frames at G = 40: 28 s of oracle time) --------
    from oracle_check import blocks_for, check_pass
    res = check_pass(one, imgs, toks, ssd, rsd, size, K, cfg, blocks_for(n, n, cfg.spacing))
    assert res["bit_exact"], (name, res["first_mismatch"])
    assert res["frames_checked_vs_oracle"] == n and res["pairs_checked"] == n - 1, res
    assert res["matches_checked"] > 0
    print(f"\n{name}: {res['frames_checked_vs_oracle']} frames, {res['pairs_checked']} pairs, {res['matches_checked']} matches bit-exact vs the oracle")

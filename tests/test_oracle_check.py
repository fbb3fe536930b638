"""The checker behind the parity gates (tests/oracle_check.py) tested on the CPU: a pass assembled from the oracle's own outputs
passes; one flipped descriptor bit, one wrong match index, one non-zero slot past a pair's count, or one wrong count fails and
names the place; the block plans cover every pair exactly once."""
import numpy as np
import pytest
import torch

import synth
from oracle_check import blocks_for, check_pass, oracle_block


class _Cfg:
    spacing = 1
    bn_train_mode, bn_eps = True, 1e-5
    nms_radius, min_score_percentile = 2, 0.5
    saliency_weight, min_saliency, min_descriptor_sim, min_intensity, use_intensity = 0.3, 0.5, 0.7, 0.15, True


def _padded(o, K, sp):
    n = o["idx"].shape[0]
    p = n - sp
    mt, q, c = np.zeros((p, K, 2), np.int64), np.zeros((p, K), np.float32), np.zeros((p,), np.int32)
    for i, (m, qq) in enumerate(zip(o["matches"], o["quality"])):
        mt[i, :len(m)], q[i, :len(m)], c[i] = m, qq, len(m)
    out = {k: torch.from_numpy(o[k].copy()) for k in ("idx", "scores", "descriptors", "intensity")}
    out.update(matches=torch.from_numpy(mt), quality=torch.from_numpy(q), match_count=torch.from_numpy(c))
    return out


@pytest.mark.parametrize("sp", [1, 2])
def test_check_pass_accepts_the_oracle_and_names_a_corruption(sp):
    cfg = _Cfg()
    cfg.spacing = sp
    n, K = 7, 500
    ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
    toks, imgs = torch.from_numpy(synth.token_sequence(n, 28)), torch.from_numpy(synth.image_sequence(n))
    o = oracle_block(imgs.numpy(), toks.numpy(), ssd, rsd, 448, K, cfg)
    assert sum(len(m) for m in o["matches"]) > 0
    out = _padded(o, K, sp)
    for blocks in (blocks_for(n, n, sp), blocks_for(n, n, sp, block=3), [(0, n)]):
        res = check_pass(out, imgs, toks, ssd, rsd, 448, K, cfg, blocks)
        assert res["bit_exact"] and res["frames_checked_vs_oracle"] == n and res["pairs_checked"] == n - sp, (blocks, res)
        assert res["matches_checked"] == sum(len(m) for m in o["matches"])
    # one bit of one descriptor
    bad = {k: v.clone() for k, v in out.items()}
    bad["descriptors"].view(torch.int32)[4, 17, 3] ^= 1
    res = check_pass(bad, imgs, toks, ssd, rsd, 448, K, cfg, blocks_for(n, n, sp, block=3))
    assert not res["bit_exact"] and "descriptors" in res["first_mismatch"]
    # one match index of the last pair
    bad = {k: v.clone() for k, v in out.items()}
    p = max(i for i in range(n - sp) if len(o["matches"][i]))
    bad["matches"][p, 0, 1] += 1
    res = check_pass(bad, imgs, toks, ssd, rsd, 448, K, cfg, blocks_for(n, n, sp))
    assert not res["bit_exact"] and "matches differ" in res["first_mismatch"]
    # a count that is one short, and a non-zero slot past the count
    bad = {k: v.clone() for k, v in out.items()}
    bad["match_count"][p] -= 1
    assert not check_pass(bad, imgs, toks, ssd, rsd, 448, K, cfg, blocks_for(n, n, sp))["bit_exact"]
    bad = {k: v.clone() for k, v in out.items()}
    bad["quality"][p, K - 1] = 0.5
    res = check_pass(bad, imgs, toks, ssd, rsd, 448, K, cfg, blocks_for(n, n, sp))
    assert not res["bit_exact"] and "past the count" in res["first_mismatch"]


@pytest.mark.parametrize("n,want,sp", [(613, 613, 1), (647, 647, 1), (2965, 258, 1), (512, 258, 1), (50, 50, 5), (21, 21, 1), (2, 2, 1),
                                       (300, 258, 1), (97, 97, 1), (98, 98, 1)])
def test_block_plans(n, want, sp):
    blocks = blocks_for(n, want, sp)
    pairs = [p for a, b in blocks for p in range(a, b - sp)]
    assert len(pairs) == len(set(pairs)), "a pair is covered twice"
    assert all(0 <= a < b <= n for a, b in blocks)
    if want >= n or 3 * (want // 3) >= n:
        assert sorted(pairs) == list(range(n - sp)), "a whole-sequence plan covers every pair"
    else:
        frames = sum(b - a for a, b in blocks)
        assert frames >= want - 2 and len(pairs) >= frames - 3 * sp
        assert blocks[0][0] == 0 and blocks[-1][1] == n                  # both ends

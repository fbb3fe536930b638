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


def test_n_rank_gate_pieces_on_a_two_shard_split():
    """The rank-0 side of bench.py's N > 1 gate, on arrays assembled from the oracle (no GPU, no process group): a 7-frame
    sequence cut 4 | 3.  The boundary pair (frames 3 | 4) is checked through the 'gathered' arrays against the oracle on
    regenerated frames; the gathered rows of each rank against the digest of what that rank 'computed'; per-rank reports are
    merged.  One flipped index in the boundary row, one gathered row that differs from its sender's - each is refused and named."""
    from oracle_check import check_boundaries, check_gathered_rows, digest_matches, merge_rank_reports
    cfg = _Cfg()
    cfg.spacing = 1
    n, K, cut = 7, 500, 4
    ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
    toks, imgs = synth.token_sequence(n, 28), synth.image_sequence(n)
    whole = _padded(oracle_block(imgs, toks, ssd, rsd, 448, K, cfg), K, 1)            # 6 pairs: rows 0..5
    gathered = {k: whole[k] for k in ("matches", "quality", "match_count")}
    regen = lambda a, b: (imgs[a:b], toks[a:b])                                        # noqa: E731
    ok, pairs, nm, why = check_boundaries(gathered, [cut], regen, ssd, rsd, 448, K, cfg)
    assert ok and pairs == 1 and nm == int(whole["match_count"][cut - 1]) and why is None
    # rank 0 computed pairs 0..3 (its 4 frames + the halo frame), rank 1 pairs 4..5
    ppr = [cut, n - 1 - cut]
    local = [{k: gathered[k][:cut] for k in gathered}, {k: gathered[k][cut:] for k in gathered}]
    digs = [digest_matches(m["matches"], m["quality"], m["match_count"]) for m in local]
    assert check_gathered_rows(gathered, ppr, digs)
    rep = merge_rank_reports([dict(frames_checked_vs_oracle=4, pairs_checked=3, matches_checked=10, bit_exact=True, first_mismatch=None, rank=0, digest=digs[0]),
                              dict(frames_checked_vs_oracle=3, pairs_checked=2, matches_checked=7, bit_exact=True, first_mismatch=None, rank=1, digest=digs[1])])
    assert rep["bit_exact"] and rep["frames_checked_vs_oracle"] == 7 and rep["pairs_checked"] == 5 and "digest" not in rep
    rep = merge_rank_reports([dict(frames_checked_vs_oracle=4, pairs_checked=3, matches_checked=10, bit_exact=True, first_mismatch=None, rank=0),
                              dict(frames_checked_vs_oracle=1, pairs_checked=0, matches_checked=0, bit_exact=False, first_mismatch="frames [4, 7): idx differs at block frame 0", rank=1)])
    assert not rep["bit_exact"] and rep["first_mismatch"].startswith("rank 1: ")
    # a flipped index in the boundary row
    bad = {k: v.clone() for k, v in gathered.items()}
    bad["matches"][cut - 1, 0, 1] = (bad["matches"][cut - 1, 0, 1] + 1) % K
    ok, _, _, why = check_boundaries(bad, [cut], regen, ssd, rsd, 448, K, cfg)
    assert not ok and "boundary of rank 1 (frame 4)" in why
    assert not check_gathered_rows(bad, ppr, digs)                       # ... which is also not what rank 0 sent
    # a row of rank 1's block changed in transit: the boundary check cannot see it, the digest does
    bad = {k: v.clone() for k, v in gathered.items()}
    bad["quality"][n - 2, 0] += 1.0
    assert check_boundaries(bad, [cut], regen, ssd, rsd, 448, K, cfg)[0] and not check_gathered_rows(bad, ppr, digs)

"""The CPU oracle run over whole frame blocks, and the comparison of a GPU pass with it - the checker behind bench.py's parity
gate and tests/test_gpu_fullsize.py (TEST INFRASTRUCTURE: the product never imports this).

What is compared, bit for bit (visualize_matches_sequence.py:69-104 extract, :106-197 match_with_quality, :297-320 the pair loop):
keypoint indices, scores, descriptors, intensities of every frame of a block, and for every pair (i, i + spacing) inside the block
the match count, the match pairs and the match quality."""
from __future__ import annotations

import numpy as np

from oracle import ora

FRAME_KEYS = ("idx", "scores", "descriptors", "intensity")


def oracle_block(imgs: np.ndarray, toks: np.ndarray, ssd: dict, rsd: dict, size: int, K: int, cfg, with_a0: bool = False) -> dict:
    """Frames of one block through the oracle: A2 .. A9 once per frame, M1 for every pair (i, i + cfg.spacing) inside the block
    (with_a0: also the Pillow-exact resize of every frame, as the timed CPU baseline does)."""
    n, grid, sp = toks.shape[0], size // 16, cfg.spacing
    if with_a0:
        for i in range(n):
            ora.resize_rgb(imgs[i], size)                                                  # A0
    feat = ora.bn_tokens(toks, train=cfg.bn_train_mode, eps=cfg.bn_eps)[0].reshape(n, grid, grid, 384)   # A2
    sal = ora.selector_saliency(feat, ssd)                                                 # A3
    kp, sc, idx, _ = ora.select_keypoints(sal, K, cfg.nms_radius, cfg.min_score_percentile)   # A4 / A5
    desc = ora.refine(ora.gather(feat, kp), rsd)                                           # A6 / A7
    del feat
    inten = np.stack([ora.intensity(imgs[i], size, ora.patch_to_pixel(kp[i])) for i in range(n)])   # A8 / A9
    matches, quality = [], []
    for i in range(n - sp):                                                                # M1
        m, q = ora.match_with_quality(desc[i], desc[i + sp], sc[i], sc[i + sp], cfg.saliency_weight, cfg.min_saliency,
                                      cfg.min_descriptor_sim, inten[i] if cfg.use_intensity else None,
                                      inten[i + sp] if cfg.use_intensity else None, cfg.min_intensity)
        matches.append(m)
        quality.append(q)
    return dict(idx=idx, scores=sc, descriptors=desc, intensity=inten, matches=matches, quality=quality)


def _bits(a: np.ndarray) -> np.ndarray:
    return a.view(np.uint32) if a.dtype == np.float32 else a


def compare_block(o: dict, frames: dict, match: dict | None, pair_row0: int, K: int) -> tuple:
    """o: oracle_block's result for frames [a, b); frames: name -> (b - a, ...) numpy arrays of the GPU pass for the same
    frames; match: {'matches', 'quality', 'match_count'} numpy arrays whose row pair_row0 + p is the block's pair p (None: frames
    only).  Returns (ok, frames, pairs, matches, first mismatch or None)."""
    n = o["idx"].shape[0]
    for k in FRAME_KEYS:
        if k in frames and not np.array_equal(_bits(frames[k]), _bits(o[k])):
            bad = int(np.nonzero((_bits(frames[k]) != _bits(o[k])).reshape(n, -1).any(axis=1))[0][0])
            return False, n, 0, 0, f"{k} differs at block frame {bad}"
    pairs = nm = 0
    if match is not None:
        for p, (m, q) in enumerate(zip(o["matches"], o["quality"])):
            row = pair_row0 + p
            c = int(match["match_count"][row])
            if c != len(m) or not np.array_equal(match["matches"][row, :c], m) or \
                    not np.array_equal(_bits(match["quality"][row, :c]), _bits(q)):
                return False, n, pairs, nm, f"matches differ at block pair {p} (row {row}): {c} vs {len(m)} from the oracle"
            if match["matches"][row, c:].any() or match["quality"][row, c:].any():
                return False, n, pairs, nm, f"slots past the count are not zero at row {row}"
            pairs += 1
            nm += len(m)
    return True, n, pairs, nm, None


def blocks_for(n: int, want: int, spacing: int = 1, block: int = 96) -> list:
    """Frame ranges to check: the whole sequence when want >= n (blocks of `block` pairs; consecutive blocks overlap by
    `spacing` frames so that every pair lies inside exactly one block), else ~want frames in three blocks taken from both ends
    and the middle (a block of L frames yields L - spacing pairs)."""
    if want >= n:
        return [(a, min(a + block + spacing, n)) for a in range(0, max(1, n - spacing), block)]
    per = max(spacing + 1, want // 3)
    if 3 * per >= n:
        return blocks_for(n, n, spacing, block)
    return [(0, per), (n // 2 - per // 2, n // 2 - per // 2 + per), (n - per, n)]


def check_pass(out: dict, imgs, toks, ssd: dict, rsd: dict, size: int, K: int, cfg, blocks: list, frame0: int = 0,
               match_keys=("matches", "quality", "match_count"), pair_row0: int = 0) -> dict:
    """A GPU pass (`out`: device tensors as SequencePipeline.run / ShardedSequenceRunner.run return them, rows = frames
    frame0 .. of `imgs` / `toks`) against the oracle on the frame ranges `blocks` (local to imgs / toks).  Consecutive blocks
    of a whole-sequence check overlap by `spacing` frames so that every pair is covered exactly once; the overlap's frames are
    counted once.  Row of pair p (local first frame) in the match arrays: pair_row0 + p."""
    sp = cfg.spacing
    tot = dict(frames_checked_vs_oracle=0, pairs_checked=0, matches_checked=0, bit_exact=True, first_mismatch=None)
    seen_to = -1
    for a, b in blocks:
        im = imgs[a:b].cpu().numpy()
        tk = toks[a:b].cpu().numpy()
        o = oracle_block(im, tk, ssd, rsd, size, K, cfg)
        fr = {k: out[k][a:b].cpu().numpy() for k in FRAME_KEYS if k in out}
        npairs = max(0, b - a - sp)
        mt = None
        if npairs and match_keys[0] in out:
            r0 = pair_row0 + a
            mt = {k2: out[k1][r0:r0 + npairs].cpu().numpy() for k1, k2 in zip(match_keys, ("matches", "quality", "match_count"))}
        ok, nf, npz, nm, why = compare_block(o, fr, mt, 0, K)
        tot["frames_checked_vs_oracle"] += b - max(a, min(seen_to, b))          # an overlap's frames are counted once
        seen_to = max(seen_to, b)
        tot["pairs_checked"] += npz
        tot["matches_checked"] += nm
        if not ok:
            tot["bit_exact"] = False
            tot["first_mismatch"] = f"frames [{frame0 + a}, {frame0 + b}): {why}"
            break
    return tot


# ----------------------------------------------------------------------------------------- the gate of an N-rank run (bench.py)
def digest_matches(matches, quality, match_count) -> str:
    """SHA-256 of a rank's match arrays (torch tensors, any device): what a rank computed, to be compared with its rows of the
    gathered arrays on rank 0."""
    import hashlib
    h = hashlib.sha256()
    for t in (matches, quality, match_count):
        h.update(t.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def merge_rank_reports(per_rank: list) -> dict:
    """check_pass results of every rank (each with a 'rank' key) -> one report: counts added, the first mismatch named by rank."""
    merged = dict(per_rank[0], frames_checked_vs_oracle=0, pairs_checked=0, matches_checked=0, bit_exact=True, first_mismatch=None)
    for pr in per_rank:
        for k in ("frames_checked_vs_oracle", "pairs_checked", "matches_checked"):
            merged[k] += pr[k]
        if not pr["bit_exact"]:
            merged["bit_exact"] = False
            merged["first_mismatch"] = merged["first_mismatch"] or f"rank {pr['rank']}: {pr['first_mismatch']}"
    merged.pop("rank", None)
    merged.pop("digest", None)
    return merged


def check_boundaries(gathered: dict, boundaries: list, regen, ssd: dict, rsd: dict, size: int, K: int, cfg) -> tuple:
    """The pairs that cross a shard boundary - the only ones that exercise the halo exchange and the in-place gather - through
    the GATHERED arrays: for every boundary frame b (the first frame of a rank > 0), regen(b - spacing, b + spacing) returns the
    (images, tokens) numpy arrays of those frames, the oracle runs on them, and rows b - spacing .. b - 1 of gathered
    ['matches' | 'quality' | 'match_count'] (row = the sequence's pair number = its first frame) must equal the oracle's pairs.
    Returns (ok, pairs, matches, first mismatch or None)."""
    sp = cfg.spacing
    pairs = nm = 0
    for r, b in enumerate(boundaries, start=1):
        im, tk = regen(b - sp, b + sp)
        o = oracle_block(im, tk, ssd, rsd, size, K, cfg)
        rows = {k: gathered[k][b - sp:b].cpu().numpy() for k in ("matches", "quality", "match_count")}
        ok, _, np_, nm_, why = compare_block(o, {}, rows, 0, K)
        pairs += np_
        nm += nm_
        if not ok:
            return False, pairs, nm, f"boundary of rank {r} (frame {b}): {why}"
    return True, pairs, nm, None


def check_gathered_rows(gathered: dict, pairs_per_rank: list, digests: list) -> bool:
    """What arrived on rank 0 is what every rank computed: the digest of rank r's rows of the gathered arrays == the digest rank
    r took of its local match arrays."""
    off = 0
    for npairs, d in zip(pairs_per_rank, digests):
        sl = slice(off, off + npairs)
        if digest_matches(gathered["matches"][sl], gathered["quality"][sl], gathered["match_count"][sl]) != d:
            return False
        off += npairs
    return True

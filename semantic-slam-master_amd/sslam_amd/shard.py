"""Frame-sharded execution of the hot path over the GPUs of one node - new in this build; the reference
is single-process.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  A sequence of
N frames is cut into contiguous blocks, one per rank.  Extraction is independent per frame (BatchNorm statistics
are per frame), so the only data that crosses GPUs is what the matcher needs at a block boundary:

  1. halo - each rank sends the descriptors / scores / intensities of its FIRST `spacing` frames to rank-1
     (point-to-point, <= 260 KB per frame at K = 500: latency-bound on one xGMI link, never bandwidth-bound);
  2. results - per-rank pair counts are all-gathered, then the padded match records are all-gathered and kept by
     rank 0 (all_gather rather than gather: fixed-size, the most widely exercised RCCL collective).

No all-reduce, no weight traffic after the optional initial broadcast of the packed weights.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def shard_bounds(n_frames: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: the first (n % world) ranks get one extra frame."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_weights(tensors: list, src: int = 0):
    """Rank `src` owns the checkpoint; everyone else receives the packed device buffers (6.7 MB fp32, once)."""
    for t in tensors:
        dist.broadcast(t, src=src)


class ShardedSequenceRunner:
    """extract_fn(tokens, images) -> dict with 'descriptors' (n, K, D), 'scores' (n, K), optional 'intensity' (n, K)
    match_fn(desc, scores, intensity, spacing) -> dict with 'matches' (p, K, 2) int64, 'quality' (p, K), 'match_count' (p,)
    Both run on this rank's device; in production they are SequencePipeline.extract / .match."""

    def __init__(self, extract_fn: Callable, match_fn: Callable, spacing: int = 1, group=None):
        self.extract_fn, self.match_fn, self.spacing, self.group = extract_fn, match_fn, spacing, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    # ------------------------------------------------------------------------------------------------ halo
    def _exchange_halo(self, fields: dict) -> dict:
        """Send my first `spacing` frames to rank-1, receive rank+1's.  Returns the received tensors ({} on the last rank)."""
        sp, r, w = self.spacing, self.rank, self.world
        if w == 1:
            return {}
        ops, recv = [], {}
        for name, t in fields.items():
            if t is None:
                continue
            if r > 0:
                ops.append(dist.P2POp(dist.isend, t[:sp].contiguous(), r - 1, self.group))
            if r < w - 1:
                recv[name] = torch.empty((sp,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                ops.append(dist.P2POp(dist.irecv, recv[name], r + 1, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return recv

    # ------------------------------------------------------------------------------------------------- run
    def run(self, tokens_local: torch.Tensor, images_local=None, gather_results: bool = True) -> dict:
        """Processes this rank's block.  Every rank must hold at least `spacing` frames."""
        sp = self.spacing
        assert tokens_local.shape[0] >= sp, "each shard needs at least `spacing` frames"
        ex = self.extract_fn(tokens_local, images_local)
        fields = dict(descriptors=ex["descriptors"], scores=ex["scores"], intensity=ex.get("intensity"))
        halo = self._exchange_halo(fields)
        if halo:
            desc = torch.cat([fields["descriptors"], halo["descriptors"]])
            sc = torch.cat([fields["scores"], halo["scores"]])
            inten = None if fields["intensity"] is None else torch.cat([fields["intensity"], halo["intensity"]])
        else:
            desc, sc, inten = fields["descriptors"], fields["scores"], fields["intensity"]
        m = self.match_fn(desc, sc, inten, sp)          # pairs (i, i+sp) for every local i that has a partner
        out = dict(ex)
        out.update(m)
        out["n_local_pairs"] = int(m["match_count"].shape[0])
        if gather_results and self.world > 1:
            out.update(self._gather(m))
        return out

    # ---------------------------------------------------------------------------------------------- gather
    def _gather(self, m: dict) -> dict:
        """All ranks learn every rank's pair count, then the padded match records are all-gathered (one fixed-size
        collective per array; <= 10 KB per pair) and rank 0 keeps them, concatenated in frame order."""
        w, r = self.world, self.rank
        dev = m["match_count"].device
        npairs = torch.tensor([m["match_count"].shape[0]], dtype=torch.int64, device=dev)
        all_np = [torch.zeros_like(npairs) for _ in range(w)]
        dist.all_gather(all_np, npairs, group=self.group)
        counts = [int(x.item()) for x in all_np]
        pmax = max(counts)
        K = m["matches"].shape[1]

        def pad(t, shape, dtype):
            buf = torch.zeros((pmax,) + shape, dtype=dtype, device=dev)
            buf[: t.shape[0]] = t
            return buf

        send = [pad(m["match_count"], (), torch.int32), pad(m["quality"], (K,), torch.float32),
                pad(m["matches"], (K, 2), torch.int64)]
        res = {}
        for name, t in zip(("all_match_count", "all_quality", "all_matches"), send):
            bufs = [torch.empty_like(t) for _ in range(w)]
            dist.all_gather(bufs, t, group=self.group)
            if r == 0:
                res[name] = torch.cat([b[:c] for b, c in zip(bufs, counts)])
        res["pairs_per_rank"] = counts
        return res

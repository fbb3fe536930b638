"""Frame-sharded execution of the hot path over the GPUs of one node - new in this build; the reference
is single-process.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  A sequence of
N frames is cut into contiguous blocks, one per rank (`shard_bounds`).  Extraction is independent per frame (BatchNorm
statistics are per frame), so the only data that crosses GPUs is what the matcher needs at a block boundary and the
results:

  1. halo - each rank sends the descriptors / scores / intensities of its FIRST `spacing` frames to rank-1 (point to
     point, <= 260 KB per frame at K = 500: latency-bound on one xGMI link).  Default ("late"): the block is extracted as one
     launch group into buffers with `spacing` spare rows, then the exchange is posted and the matcher waits for it - one small
     point-to-point transfer exposed per step.  "early": those frames are extracted FIRST, as their own small launch group,
     and the isend / irecv are posted before the rest of the block is extracted - the transfer hides under the extraction, but
     the one-frame extraction costs more than it hides (ShardedSequenceRunner.__init__);
  2. results - rank 0 alone receives the matches, in one of two forms (`ShardedSequenceRunner.run(gather=...)`):
     "padded" (default): every rank r > 0 sends its fixed-capacity match arrays as they are - (pairs, K, 2) int64, (pairs, K)
     fp32, (pairs,) int32 counts; 20 bytes per SLOT, 6.1 MB per 612 pairs at K = 500 - and rank 0 receives them IN PLACE into
     its rows of the sequence-sized result (slots past a pair's count are zero: the arrays are a pure function of the inputs).
     No device-side work, and when the caller passes the ranks' frame counts (`frames_per_rank`, known from `shard_bounds`)
     no size exchange and no host synchronisation: the step is enqueue-only on every rank.
     "records": the matches are COMPACTED on the sending GPU - one int32 record (pair index - local on the wire, turned into
     the sequence's pair number on rank 0 -, idx1, idx2, quality bits) per match, 16 bytes, about half the bytes of the padded
     form at the bench's match rate - sizes are exchanged in one 16-byte all-gather (a host synchronisation), one send / recv
     per rank, and rank 0 expands the records into the padded arrays.  For links where bytes, not latency, are the cost:
     `tools/gather_cost.py` puts compaction + expansion at 0.1 + 0.3 ms per step beside a 10.8 ms step, which is why it is
     not the default on xGMI.

No all-reduce, no all-gather of payload, no weight traffic after the initial broadcast of the packed weights from rank 0
(`pipeline_from_rank0`: one rank reads and packs the checkpoint, the others receive 6.7 MB once).
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


def _host_staged(group=None) -> bool:
    """True on a backend without device-memory transport (gloo: the CPU tests, and the shared-GPU rehearsal of bench.py):
    CUDA tensors then travel through host memory.  On nccl (= RCCL over xGMI) device buffers are handed over as they are."""
    return dist.get_backend(group) == "gloo"


def _broadcast(t: torch.Tensor, src: int, group=None):
    if t.is_cuda and _host_staged(group):
        c = t.cpu()
        dist.broadcast(c, src=src, group=group)
        t.copy_(c)
    else:
        dist.broadcast(t, src=src, group=group)


class _P2P:
    """One batch of isend / irecv (batch_isend_irecv); CUDA tensors are staged through host memory where _host_staged()."""

    def __init__(self, group=None):
        self.group, self.ops, self.after, self.reqs = group, [], [], []
        self.staged = _host_staged(group)

    def send(self, t: torch.Tensor, dst: int):
        self.ops.append(dist.P2POp(dist.isend, t.cpu() if (t.is_cuda and self.staged) else t, dst, self.group))

    def recv(self, t: torch.Tensor, src: int):
        if t.is_cuda and self.staged:
            buf = torch.empty(t.shape, dtype=t.dtype)
            self.after.append((t, buf))
            t = buf
        self.ops.append(dist.P2POp(dist.irecv, t, src, self.group))

    def post(self):
        self.reqs = dist.batch_isend_irecv(self.ops) if self.ops else []
        return self

    def wait(self):
        for req in self.reqs:
            req.wait()
        for t, buf in self.after:
            t.copy_(buf)


def broadcast_weights(tensors: list, src: int = 0, group=None):
    """Rank `src` owns the checkpoint; everyone else receives the packed device buffers (6.7 MB fp32, once)."""
    for t in tensors:
        _broadcast(t, src, group)


def pipeline_from_rank0(cfg, selector_state: dict | None, refiner_state: dict | None, device, bn_state: dict | None = None,
                        src: int = 0, group=None):
    """rank `src` alone reads the checkpoint and packs the weights into kernel order; the other ranks
    (selector_state = refiner_state = None) allocate the packed buffers uninitialised and receive them by broadcast
    (RCCL over xGMI on the GPUs; gloo in the CPU tests).  A three-number header (selector hidden width, residual blocks,
    BatchNorm-state flag) travels first so that the receivers can size their buffers.  Returns this rank's SequencePipeline."""
    from .pipeline import SequencePipeline, refiner_weight_list
    rank = dist.get_rank(group)
    dev = torch.device(device)
    head = torch.zeros(3, dtype=torch.int64, device=dev)
    if rank == src:
        if selector_state is None or refiner_state is None:
            raise ValueError(f"rank {src} must hold the state dicts")
        head[0] = int(tuple(selector_state["conv.0.weight"].shape)[0])
        head[1] = refiner_weight_list(refiner_state)[1]
        head[2] = 1
    _broadcast(head, src, group)
    hidden, n_blocks, ok = (int(v) for v in head.tolist())
    if not ok:
        raise RuntimeError("weight broadcast header missing")
    if rank == src:
        pipe = SequencePipeline(cfg, selector_state, refiner_state, bn_state, device=dev)
    else:
        pipe = SequencePipeline(cfg, None, None, None, device=dev, empty_shapes=(hidden, n_blocks))
    broadcast_weights(pipe.weight_tensors(), src=src, group=group)
    return pipe


def compact_records(matches: torch.Tensor, quality: torch.Tensor, match_count: torch.Tensor, first_pair: int = 0):
    """(p, K, 2) int64 / (p, K) fp32 / (p,) counts -> ((p*K, 4) int32 buffer, device scalar M): rows 0..M-1 are (pair index
    + first_pair, idx1, idx2, quality bits), pairs ascending and idx1 ascending inside a pair - exactly the valid slots, in
    order.  Prefix-sum + scatter on the device: no host synchronisation (boolean-mask indexing would need one)."""
    p, K = quality.shape
    dev = quality.device
    buf = torch.zeros((p * K + 1, 4), dtype=torch.int32, device=dev)          # last row: dump slot for the padding
    if p == 0:
        return buf[:0], torch.zeros((), dtype=torch.int64, device=dev)
    cnt = match_count.to(torch.int64)
    start = torch.cumsum(cnt, 0) - cnt
    slot = torch.arange(K, device=dev)
    keep = slot[None, :] < cnt[:, None]
    dst = torch.where(keep, start[:, None] + slot[None, :], torch.full((), p * K, dtype=torch.int64, device=dev))
    pair = (torch.arange(p, device=dev, dtype=torch.int32) + first_pair)[:, None].expand(p, K)
    rec = torch.stack([pair, matches[..., 0].to(torch.int32), matches[..., 1].to(torch.int32), quality.view(torch.int32)], dim=-1)
    buf[dst.reshape(-1)] = rec.reshape(-1, 4)
    return buf[: p * K], cnt.sum()


def expand_records(rec: torch.Tensor, n_pairs: int, K: int) -> dict:
    """Inverse of compact_records on the receiving side: the padded arrays of a single-process run.
    The records must be what compact_records emits - pair indices in [0, n_pairs), non-decreasing, at most K per pair;
    anything else raises ValueError here instead of an out-of-bounds indexed write on the device."""
    dev = rec.device
    matches = torch.zeros((n_pairs, K, 2), dtype=torch.int64, device=dev)
    quality = torch.zeros((n_pairs, K), dtype=torch.float32, device=dev)
    pair = rec[:, 0].long()
    if rec.shape[0]:
        bad = (pair < 0).any() | (pair >= n_pairs).any() | (pair[1:] < pair[:-1]).any()
        if bool(bad):
            raise ValueError("match records out of order or out of range (pair index must be non-decreasing in [0, n_pairs))")
    count = torch.bincount(pair, minlength=n_pairs).to(torch.int32)
    if rec.shape[0] and int(count.max()) > K:
        raise ValueError(f"more than K = {K} match records for one pair")
    start = torch.cumsum(count, 0) - count
    slot = torch.arange(rec.shape[0], device=dev) - start[pair]
    matches[pair, slot, 0] = rec[:, 1].long()
    matches[pair, slot, 1] = rec[:, 2].long()
    quality[pair, slot] = rec[:, 3].contiguous().view(torch.float32)
    return dict(all_matches=matches, all_quality=quality, all_match_count=count)


def _takes_out(fn) -> bool:
    import inspect
    try:
        return "out" in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


class ShardedSequenceRunner:
    """extract_fn(tokens, images[, out]) -> dict with 'descriptors' (n, K, D), 'scores' (n, K), optional 'intensity' (n, K);
    if it takes `out` (a dict of row slices of preallocated buffers, as SequencePipeline.extract does) the block's outputs
    are written straight into block-sized buffers - no concatenation of the boundary group and the rest.
    match_fn(desc, scores, intensity, spacing) -> dict with 'matches' (p, K, 2) int64, 'quality' (p, K), 'match_count' (p,)
    Both run on this rank's device; in production they are SequencePipeline.extract / .match.
    Pair numbers in the gathered records are the SEQUENCE's pair numbers: every rank sends local pair indices and rank 0
    adds the exclusive prefix sum of the ranks' pair counts (from the size exchange) - callers pass no offset."""

    def __init__(self, extract_fn: Callable, match_fn: Callable, spacing: int = 1, group=None, alloc_fn: Callable | None = None,
                 halo: str = "late"):
        """alloc_fn(rows) -> dict of output buffers of extract_fn for `rows` frames (SequencePipeline.alloc_extract).
        halo: "late" (needs alloc_fn and an extract_fn that takes `out`) - the block is extracted as ONE launch group into
        buffers with `spacing` spare rows, then the first frames' fields go to rank - 1 and the neighbour's arrive in the
        spare rows: what stays exposed is one small point-to-point exchange in front of the matcher;
        "early" - the boundary frames are extracted first, as their own group, and travel while the rest of the block is
        extracted: nothing exposed, but a one-frame extraction is a latency-bound pass of its own (tools/halo_cost.py:
        +0.26 ms per 613-frame block, 2.4 % of the step - more than the exchange it hides).  Without alloc_fn: "early"."""
        if halo not in ("late", "early"):
            raise ValueError(f"halo must be 'late' or 'early', got {halo!r}")
        self.extract_fn, self.match_fn, self.spacing, self.group = extract_fn, match_fn, spacing, group
        self.alloc_fn, self.halo = alloc_fn, halo
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    # ------------------------------------------------------------------------------------------------ halo
    def _post_halo(self, send: dict, recv: dict):
        """Post the sends of my first `spacing` frames (`send`: name -> (sp, ...) tensor) to rank-1 and the receives from
        rank+1 into `recv` (name -> (sp, ...) view of the block buffer's halo rows); returns the requests."""
        r, w = self.rank, self.world
        p2p = _P2P(self.group)
        for name, t in send.items():
            if t is None:
                continue
            if r > 0:
                p2p.send(t, r - 1)
            if r < w - 1:
                p2p.recv(recv[name], r + 1)
        return p2p.post()

    # ------------------------------------------------------------------------------------------------- run
    def run(self, tokens_local: torch.Tensor, images_local=None, gather_results: bool = True, gather: str = "padded",
            frames_per_rank: list | None = None) -> dict:
        """Processes this rank's block.  Every rank must hold at least `spacing` frames.
        gather: "padded" | "records" (module docstring, item 2).  frames_per_rank: the frame count of every rank's block
        (shard_bounds) - with it the padded gather needs no size exchange, hence no host synchronisation in the step."""
        if gather not in ("padded", "records"):
            raise ValueError(f"gather must be 'padded' or 'records', got {gather!r}")
        if frames_per_rank is not None and (len(frames_per_rank) != self.world or int(frames_per_rank[self.rank]) != tokens_local.shape[0]):
            raise ValueError("frames_per_rank must list every rank's frame count (this rank's entry = tokens_local.shape[0])")
        sp, n = self.spacing, tokens_local.shape[0]
        assert n >= sp, "each shard needs at least `spacing` frames"
        names = ("descriptors", "scores", "intensity")
        img = (lambda a, b: None if images_local is None else images_local[a:b])
        if self.world == 1:
            ex = self.extract_fn(tokens_local, images_local)
            fields = {k: ex.get(k) for k in names}
        elif self.halo == "late" and self.alloc_fn is not None and _takes_out(self.extract_fn):
            # the whole block as ONE launch group, written into buffers with `sp` spare rows; then the exchange, then the matcher
            n_halo = sp if self.rank < self.world - 1 else 0
            full = self.alloc_fn(n + n_halo)
            if images_local is None:
                # alloc_fn does not know whether pixels were passed: without them nothing writes 'intensity' - it must not be
                # exchanged or handed to the matcher (whose intensity threshold would then read uninitialised memory)
                full = {k: v for k, v in full.items() if k != "intensity"}
            self.extract_fn(tokens_local, images_local, out={k: v[:n] for k, v in full.items()})
            self._post_halo({k: (full[k][:sp] if k in full else None) for k in names},
                            {k: full[k][n:n + sp] for k in names if k in full}).wait()
            ex = {k: v[:n] for k, v in full.items()}
            fields = {k: (full[k][:n + n_halo] if k in full else None) for k in names}
        else:
            # boundary frames first, as their own small launch group: their descriptors travel while the rest of the block
            # is being extracted.  Block buffers hold n rows + `sp` halo rows for the fields the matcher reads, so the
            # boundary group, the rest of the block and the neighbour's halo all land in place.
            n_halo = sp if self.rank < self.world - 1 else 0
            ex_head = self.extract_fn(tokens_local[:sp], img(0, sp))
            full = {k: torch.empty((n + (n_halo if k in names else 0),) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                    for k, v in ex_head.items() if isinstance(v, torch.Tensor)}
            for k, v in full.items():
                v[:sp] = ex_head[k]
            reqs = self._post_halo({k: (full[k][:sp] if k in full else None) for k in names},
                                   {k: full[k][n:n + sp] for k in names if k in full})
            if n > sp:
                if _takes_out(self.extract_fn):
                    self.extract_fn(tokens_local[sp:], img(sp, n), out={k: v[sp:n] for k, v in full.items()})
                else:
                    ex_tail = self.extract_fn(tokens_local[sp:], img(sp, n))
                    for k, v in full.items():
                        v[sp:n] = ex_tail[k]
            reqs.wait()
            ex = {k: v[:n] for k, v in full.items()}
            fields = {k: full.get(k) for k in names}      # n + halo rows
        m = self.match_fn(fields["descriptors"], fields["scores"], fields["intensity"], sp)   # every local i that has a partner
        out = dict(ex)
        out.update(m)
        out["n_local_pairs"] = int(m["match_count"].shape[0])
        if gather_results and self.world > 1:
            out.update(self._gather(m) if gather == "records" else self._gather_padded(m, n, frames_per_rank))
        return out

    # ------------------------------------------------------------------------------------------- gather (padded)
    def _gather_padded(self, m: dict, n_local: int, frames_per_rank: list | None) -> dict:
        """The fixed-capacity match arrays of every rank -> rank 0, received in place into the sequence-sized result
        ('all_matches', 'all_quality', 'all_match_count'); other ranks return only the pair counts."""
        w, r, sp = self.world, self.rank, self.spacing
        dev = m["matches"].device
        if frames_per_rank is None:
            # the blocks' sizes are not known here: one 8-byte all-gather (a host synchronisation; callers that know them pass them)
            mine = torch.tensor([n_local], dtype=torch.int64, device="cpu" if _host_staged(self.group) else dev)
            got = [torch.zeros_like(mine) for _ in range(w)]
            dist.all_gather(got, mine, group=self.group)
            frames_per_rank = [int(t.item()) for t in got]
        # every local frame has a partner except the last `spacing` frames of the LAST rank (the halo covers the others)
        pairs = [int(f) - (sp if q == w - 1 else 0) for q, f in enumerate(frames_per_rank)]
        assert pairs[r] == m["match_count"].shape[0], (pairs, r, m["match_count"].shape)
        res = {"pairs_per_rank": pairs}
        names = ("matches", "quality", "match_count")
        p2p = _P2P(self.group)
        if r == 0:
            total = sum(pairs)
            full = {k: torch.empty((total,) + tuple(m[k].shape[1:]), dtype=m[k].dtype, device=dev) for k in names}
            off = 0
            for src in range(w):
                for k in names:
                    if src == 0:
                        full[k][: pairs[0]] = m[k]
                    elif pairs[src]:
                        p2p.recv(full[k][off:off + pairs[src]], src)
                off += pairs[src]
            p2p.post().wait()
            res.update(all_matches=full["matches"], all_quality=full["quality"], all_match_count=full["match_count"])
        elif pairs[r]:
            for k in names:
                p2p.send(m[k].contiguous(), 0)
            p2p.post().wait()
        return res

    # ------------------------------------------------------------------------------------------ gather (records)
    def _gather(self, m: dict) -> dict:
        """Compacted match records -> rank 0 (module docstring, item 2, "records").  Rank 0 returns the padded arrays of the whole
        sequence ('all_matches', 'all_quality', 'all_match_count') and the raw records; other ranks only the sizes."""
        w, r = self.world, self.rank
        rec, n_valid = compact_records(m["matches"], m["quality"], m["match_count"])      # LOCAL pair indices
        dev = rec.device
        sizes = torch.stack([torch.tensor(m["match_count"].shape[0], dtype=torch.int64, device=dev), n_valid])
        if sizes.is_cuda and _host_staged(self.group):
            sizes = sizes.cpu()
        all_sizes = [torch.zeros_like(sizes) for _ in range(w)]
        dist.all_gather(all_sizes, sizes, group=self.group)
        table = torch.stack(all_sizes).tolist()       # the step's one host synchronisation: [[pairs, records], ...] per rank
        pairs = [int(t[0]) for t in table]
        nrec = [int(t[1]) for t in table]
        res = {"pairs_per_rank": pairs, "records_per_rank": nrec}
        K = m["matches"].shape[1]
        if r == 0:
            buf = torch.empty((sum(nrec), 4), dtype=torch.int32, device=dev)
            buf[: nrec[0]] = rec[: nrec[0]]
            p2p, off = _P2P(self.group), nrec[0]
            for src in range(1, w):
                if nrec[src]:
                    p2p.recv(buf[off:off + nrec[src]], src)
                off += nrec[src]
            p2p.post().wait()
            # local -> sequence pair numbers: rank src's pairs start at the sum of the pair counts of the ranks before it
            off, first = nrec[0], pairs[0]
            for src in range(1, w):
                if nrec[src]:
                    buf[off:off + nrec[src], 0] += first
                off, first = off + nrec[src], first + pairs[src]
            res["records"] = buf
            res.update(expand_records(buf, sum(pairs), K))
        elif nrec[r]:
            p2p = _P2P(self.group)
            p2p.send(rec[: nrec[r]].contiguous(), 0)
            p2p.post().wait()
        return res

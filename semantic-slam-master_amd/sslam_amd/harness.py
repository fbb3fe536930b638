"""Counterpart of the reference's per-pair harness (SequenceMatcher + process_spacing,
semantic-slam/visualize_matches_sequence.py:28-104, 272-357) on the batched HIP pipeline, without the plotting.

* `SequenceMatcher.extract(image_path)` / `.match_with_quality(...)`: same call shapes and return dict keys as the
  reference, for scripts that go pair by pair.
* `StreamingSequence`: the throughput form - every frame is extracted ONCE, its descriptors stay on the
  device, and all requested spacings (default 1, 5, 10, 15, 20 as at visualize_matches_sequence.py:369) are matched
  from that one set of descriptors, one batched launch pair per spacing.
"""
from __future__ import annotations

import numpy as np
import torch

import matching
from . import lib
from .pipeline import N_PREFIX, ExtractorConfig, SequencePipeline


class SequenceMatcher:
    def __init__(self, backbone, selector_state: dict, refiner_state: dict, cfg: ExtractorConfig | None = None,
                 device: str = "cuda"):
        """backbone: a models.dino_backbone.DinoBackbone (its `dino` ViT produces the tokens; third-party weights)."""
        self.cfg = cfg or ExtractorConfig()
        self.device = torch.device(device)
        self.backbone = backbone.to(self.device)
        bn = self.backbone.feature_norm
        self.pipe = SequencePipeline(self.cfg, selector_state, refiner_state,
                                     dict(weight=bn.weight, bias=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var),
                                     device=self.device)

    @torch.no_grad()
    def extract_batch(self, images_u8: np.ndarray) -> dict:
        """(n, H, W, 3) uint8 frames -> the reference's per-frame dict, batched, still on the device."""
        img = torch.from_numpy(np.ascontiguousarray(images_u8)).to(self.device)
        tokens = self.backbone.forward_tokens(self.pipe.preprocess(img)).float().contiguous()   # same ViT path as backbone.forward()
        out = self.pipe.extract(tokens, img)
        return {"saliency": out["saliency"], "keypoints_pixel": out["keypoints_pixel"], "scores": out["scores"],
                "intensity": out["intensity"], "descriptors": out["descriptors"]}

    def extract(self, image_path: str) -> dict:
        """One image -> numpy dict with the reference's keys (visualize_matches_sequence.py:97-104)."""
        from PIL import Image
        image = Image.open(image_path).convert("RGB")
        out = self.extract_batch(np.array(image)[None])
        res = {k: v[0].cpu().numpy() for k, v in out.items()}
        res["image"] = image
        return res

    match_with_quality = staticmethod(matching.match_with_quality)


class StreamingSequence:
    """Streaming pair scheduler.  Frames arrive in chunks (`push`); every frame is extracted once; each push
    matches, for every spacing s, exactly the pairs (j - s, j) whose SECOND frame j arrived in that push - so over a
    whole sequence every pair (i, i + s) is matched once, whatever the chunking.

    Two storage modes:
      * `reset(capacity=n)` (a sequence of known length; `run`, `run_frames`, `run_directory`): keypoints / descriptors /
        scores / intensities of ALL n frames and the match arrays of every spacing live in sequence-sized device buffers
        allocated once (613 frames x 264 KB = 162 MB - nothing beside 288 GB); every push writes its rows in place and the
        matcher reads pair (i, i + s) by pointer offset: no ring, no concatenation, no copy;
      * `reset()` (unbounded stream): a ring of the last max(spacings) frames' descriptors / scores / intensities.

    The reference's process_spacing (visualize_matches_sequence.py:298-300) visits only i = 0, s, 2s, ... and stops
    after `max_pairs` pairs: `reference_pairs()` selects those rows from the result."""

    def __init__(self, pipe: SequencePipeline, spacings=(1, 5, 10, 15, 20)):
        self.pipe, self.spacings = pipe, tuple(int(s) for s in spacings)
        if not self.spacings or min(self.spacings) < 1:
            raise ValueError("spacings must be positive")
        self.reset()

    def reset(self, capacity: int | None = None):
        self.n_seen = 0
        self.capacity = capacity
        self._ring = None        # ring mode: dict of (r, ...) tensors, the last r <= max(spacings) frames
        self._store = None       # capacity mode: extract buffers of `capacity` frames
        self._pairs = {}         # capacity mode: spacing -> match buffers of (capacity - s) pairs

    # ------------------------------------------------------------------------------------ capacity mode
    def _push_in_place(self, tokens, images_u8, images_ready=None):
        pipe, m, n0 = self.pipe, tokens.shape[0], self.n_seen
        if n0 + m > self.capacity:
            raise ValueError(f"push past the declared capacity ({n0} + {m} > {self.capacity})")
        if self._store is None:
            self._store = pipe.alloc_extract(self.capacity, images_u8 is not None)
            for s in self.spacings:
                if self.capacity > s:
                    self._pairs[s] = pipe.alloc_match(self.capacity - s)
        st = self._store
        pipe.extract(tokens, images_u8, out={k: v[n0:n0 + m] for k, v in st.items()}, images_ready=images_ready)
        res = {"frames": {k: v[n0:n0 + m] for k, v in st.items()}}
        for s in self.spacings:
            lo = max(n0, s) - s                                   # first pair whose second frame is new
            cnt = n0 + m - s - lo
            if cnt <= 0:
                continue
            mm = pipe.match(st["descriptors"][lo:lo + cnt + s], st["scores"][lo:lo + cnt + s],
                            st["intensity"][lo:lo + cnt + s] if "intensity" in st else None, spacing=s,
                            out={k: v[lo:lo + cnt] for k, v in self._pairs[s].items()})
            mm["first"] = torch.arange(lo, lo + cnt, dtype=torch.int64, device=tokens.device)
            res[s] = mm
        self.n_seen += m
        return res

    def result(self) -> dict:
        """Capacity mode: {'frames': ..., s: match dict with one row per pair (i, i + s) seen so far} - views, no copies."""
        n = self.n_seen
        res = {"frames": {k: v[:n] for k, v in (self._store or {}).items()}}
        for s, bufs in self._pairs.items():
            if n > s:
                res[s] = {k: v[:n - s] for k, v in bufs.items()}
                res[s]["first"] = torch.arange(0, n - s, dtype=torch.int64, device=bufs["match_count"].device)
        return res

    # ---------------------------------------------------------------------------------------------- push
    def push(self, tokens: torch.Tensor, images_u8: torch.Tensor | None = None, images_ready=None) -> dict:
        """Extract `tokens.shape[0]` new frames and match them against the earlier ones.  Returns
        {'frames': extract dict of the new frames, s: {'first': global index of each pair's first frame (device int64),
        'matches', 'quality', 'match_count', ...}} for every spacing that has a pair ending in this chunk.
        images_ready: event of an upload of images_u8 still in flight (SequencePipeline.extract waits for it only in front
        of the first kernel that reads pixels)."""
        if self.capacity is not None:
            return self._push_in_place(tokens, images_u8, images_ready)
        ex = self.pipe.extract(tokens, images_u8, images_ready=images_ready)
        m = tokens.shape[0]
        keys = ["descriptors", "scores"] + (["intensity"] if "intensity" in ex else [])
        cat = {k: (ex[k] if self._ring is None else torch.cat([self._ring[k], ex[k]])) for k in keys}
        r = 0 if self._ring is None else self._ring["descriptors"].shape[0]
        base = self.n_seen - r                                   # global index of cat[...][0]
        res = {"frames": ex}
        for s in self.spacings:
            lo = max(self.n_seen, s) - s                         # global index of the first pair's first frame
            cnt = self.n_seen + m - s - lo                       # pairs (i, i + s) with n_seen <= i + s < n_seen + m
            if cnt <= 0:
                continue
            a = lo - base
            sub = {k: v[a:a + cnt + s] for k, v in cat.items()}
            mm = self.pipe.match(sub["descriptors"], sub["scores"], sub.get("intensity"), spacing=s)
            mm["first"] = torch.arange(lo, lo + cnt, dtype=torch.int64, device=tokens.device)
            res[s] = mm
        keep = max(self.spacings)
        self._ring = {k: v[-keep:] for k, v in cat.items()}
        self.n_seen += m
        return res

    def run(self, tokens: torch.Tensor, images_u8: torch.Tensor | None = None, chunk: int | None = None) -> dict:
        """Whole sequence: {'frames': ..., s: match dict with one row per pair (i, i + s), i = 0 .. n - s - 1}.
        chunk=None pushes everything at once; otherwise frames are pushed `chunk` at a time (same result)."""
        n = tokens.shape[0]
        self.reset(capacity=n)
        step = n if chunk is None else int(chunk)
        for a in range(0, n, step):
            self.push(tokens[a:a + step], None if images_u8 is None else images_u8[a:a + step])
        return self.result()

    @staticmethod
    def reference_pairs(n_frames: int, spacing: int, max_pairs: int | None = None) -> list:
        """First-frame indices process_spacing visits: range(0, n - spacing, spacing), at most max_pairs of them."""
        idx = list(range(0, n_frames - spacing, spacing))
        return idx if max_pairs is None else idx[:max_pairs]


# ------------------------------------------------------------------------------------------------------------------
# Host -> device feed.  The reference loads and uploads one frame at a time, synchronously
# (visualize_matches_sequence.py:71-72).  Here chunk i + 1 is produced on the host (PNG decode on a thread pool, or a slice
# of a host array) and uploaded on a side stream while chunk i is being extracted and matched.
class FrameFeeder:
    """Chunked upload of (n, H, W, 3) uint8 frames on a side stream, running AHEAD of the compute.

    fill(host_rows: np.ndarray (m, H, W, 3) uint8 view of a PINNED staging buffer, a, b) writes frames [a, b) into
    host_rows (called on the feeder thread).  If `pinned_source` (a pinned (n, H, W, 3) uint8 torch tensor) is given
    instead, chunks upload straight from it - no staging copy.

    Device side: HBM is sized for whole sequences (613 frames = 565 MB, 2 585 = 2.4 GB of 288 GB), so by default every
    chunk uploads into ITS rows of one sequence-sized device buffer: no slot is ever reused, the uploads never wait for
    the compute and stream back to back at the PCIe rate while the compute follows one chunk behind.  Above `max_bytes`
    (long or unbounded sequences) the device buffer is a ring of `ring` chunk slots, each reused only after the compute
    that read it has passed (release events).  Iterating yields (a, b, device_images, ready): `ready` is the event of
    the chunk's upload - the CONSUMER makes its stream wait for it, in front of the first kernel that reads pixels."""

    _cache: dict = {}      # (device, shape) -> device buffer kept between sequences (a 565 MB hipMalloc costs milliseconds)

    def __init__(self, n: int, h: int, w: int, device, bounds: list, fill=None, pinned_source: torch.Tensor | None = None,
                 max_bytes: int = 16 << 30, ring: int = 4):
        if (fill is None) == (pinned_source is None):
            raise ValueError("exactly one of fill / pinned_source")
        self.n, self.h, self.w, self.device, self.bounds = n, h, w, torch.device(device), list(bounds)
        self.fill, self.src = fill, pinned_source
        cmax = max(b - a for a, b in self.bounds)
        nb = len(self.bounds)
        self.whole = n * h * w * 3 <= max_bytes
        self.n_slots = nb if self.whole else min(ring, nb)
        shape = (n, h, w, 3) if self.whole else (self.n_slots, cmax, h, w, 3)
        key = (str(self.device), "dev")
        buf = FrameFeeder._cache.get(key)
        numel = int(np.prod(shape))
        if buf is None or buf.numel() < numel:
            buf = FrameFeeder._cache[key] = torch.empty(numel, dtype=torch.uint8, device=self.device)
        self.dev = buf[:numel].view(shape)
        self.pin = None
        if fill is not None:
            pkey, pn = (str(self.device), "pin"), 2 * cmax * h * w * 3
            pbuf = FrameFeeder._cache.get(pkey)
            if pbuf is None or pbuf.numel() < pn:
                pbuf = FrameFeeder._cache[pkey] = torch.empty(pn, dtype=torch.uint8).pin_memory()
                FrameFeeder._cache[(str(self.device), "staged")] = {}
            # the two halves of THIS feeder may overlap either half of an earlier feeder's split of the same pinned buffer
            # (other chunk size): the events of the last uploads FROM the buffer live beside it, keyed by byte range, and a
            # half is written only after every recorded upload from an overlapping range has completed
            self.pin_range = [(i * (pn // 2), (i + 1) * (pn // 2)) for i in range(2)]
            self.pin = [pbuf[a:b].view(cmax, h, w, 3) for a, b in self.pin_range]
            self._staged_events = FrameFeeder._cache.setdefault((str(self.device), "staged"), {})
        skey = (str(self.device), "stream")
        if skey not in FrameFeeder._cache:
            FrameFeeder._cache[skey] = torch.cuda.Stream(self.device)
        self.copy_stream = FrameFeeder._cache[skey]
        self.uploaded = [None] * nb                  # event per chunk: its H2D is complete (the compute waits on it)
        self.released = [None] * self.n_slots        # ring mode: the compute has finished with the slot (the copy waits on it)

    def _view(self, i: int):
        a, b = self.bounds[i]
        return self.dev[a:b] if self.whole else self.dev[i % self.n_slots][: b - a]

    def _enqueue_upload(self, i: int, src: torch.Tensor):
        """H2D of chunk i on the copy stream (ring mode: after the compute that last read the device slot) + its event."""
        with torch.cuda.device(self.device), torch.cuda.stream(self.copy_stream):
            if not self.whole:
                rel = self.released[i % self.n_slots]
                if rel is not None:
                    self.copy_stream.wait_event(rel)
            self._view(i).copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.uploaded[i] = ev
        return ev

    def _produce(self, i: int):
        """fill mode, chunk i on the feeder thread: fill the pinned staging buffer (after the previous upload FROM it has
        completed - PIL decode and numpy copies release the GIL), then enqueue its upload."""
        a, b = self.bounds[i]
        ps = i & 1
        lo, hi = self.pin_range[ps]
        for (ea, eb), ev in list(self._staged_events.items()):
            if ea < hi and lo < eb:                  # an upload FROM overlapping pinned bytes (this feeder's or an earlier one's)
                ev.synchronize()
                if lo <= ea and eb <= hi:
                    self._staged_events.pop((ea, eb), None)
        self.fill(self.pin[ps][: b - a].numpy(), a, b)
        self._staged_events[(lo, hi)] = self._enqueue_upload(i, self.pin[ps][: b - a])
        return i

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        nb = len(self.bounds)
        cur = torch.cuda.current_stream(self.device)
        self.copy_stream.wait_stream(cur)              # the device buffer may still be read by earlier work of the caller
        # how far the uploads may run ahead of the consumer: everything (whole-sequence buffer) or the free ring slots
        ahead = nb if self.whole else self.n_slots - 1
        # a pinned source needs no host work: the CALLING thread enqueues the copies (a helper thread would wait for the GIL
        # behind the launch loop - measured: uploads enqueued milliseconds late); a `fill` source decodes on a feeder thread
        feeder = ThreadPoolExecutor(max_workers=1) if self.fill is not None else None
        try:
            futs, nxt = {}, 0
            for i in range(nb):
                while nxt < nb and nxt <= i + ahead:
                    # ring mode: the slot of chunk nxt was last read by chunk nxt - n_slots <= i - 1, released below
                    if feeder is not None:
                        futs[nxt] = feeder.submit(self._produce, nxt)
                    else:
                        a, b = self.bounds[nxt]
                        self._enqueue_upload(nxt, self.src[a:b])
                    nxt += 1
                if feeder is not None:
                    futs.pop(i).result()                  # chunk i is filled and its upload enqueued
                a, b = self.bounds[i]
                yield a, b, self._view(i), self.uploaded[i]
                if not self.whole:
                    rel = torch.cuda.Event()
                    rel.record(cur)                       # the compute of chunk i is enqueued
                    self.released[i % self.n_slots] = rel
        finally:
            if feeder is not None:
                feeder.shutdown(wait=True)


def chunk_bounds(n: int, chunk: int, first: int | None = None) -> list:
    """[a, b) chunk boundaries: an optional smaller first chunk (its upload is the only one that nothing overlaps)."""
    out, a = [], 0
    if first and first < chunk and n > first:
        out.append((0, first))
        a = first
    while a < n:
        out.append((a, min(a + chunk, n)))
        a += chunk
    return out


@torch.no_grad()
def spacing_summary(result: dict, spacing: int, max_pairs: int | None = None) -> dict:
    """The statistics process_spacing prints for one spacing (visualize_matches_sequence.py:345-356), from a
    StreamingSequence / run_directory / run_frames result: the reference visits the pairs (i, i + spacing) for
    i = 0, spacing, 2 spacing, ... and stops after `max_pairs` of them (:297-299; CLI default 1, :371) - the result holds
    EVERY pair (i, i + spacing), so the reference's subset is a strided slice of it.  One device-side reduction and one
    host read-back of six numbers.  Returns {'pairs', 'matches', 'mean_quality', 'min_quality', 'max_quality',
    'high_quality'} ('high_quality': matches with quality > 0.8; the three quality figures are None without matches, where the
    reference prints no summary)."""
    if spacing not in result:
        return dict(pairs=0, matches=0, mean_quality=None, min_quality=None, max_quality=None, high_quality=0)
    m = result[spacing]
    n_pairs_all = m["match_count"].shape[0]
    first = StreamingSequence.reference_pairs(n_pairs_all + spacing, spacing, max_pairs)
    if not first:
        return dict(pairs=0, matches=0, mean_quality=None, min_quality=None, max_quality=None, high_quality=0)
    step = spacing
    q = m["quality"][first[0]:first[-1] + 1:step]
    cnt = m["match_count"][first[0]:first[-1] + 1:step].long()
    valid = torch.arange(q.shape[1], device=q.device)[None, :] < cnt[:, None]
    total = cnt.sum()
    qs = torch.where(valid, q, torch.zeros((), dtype=q.dtype, device=q.device))
    stats = torch.stack([total.double(), qs.double().sum(),
                         torch.where(valid, q, torch.full((), float("inf"), dtype=q.dtype, device=q.device)).min().double(),
                         torch.where(valid, q, torch.full((), float("-inf"), dtype=q.dtype, device=q.device)).max().double(),
                         (valid & (q > 0.8)).sum().double()]).tolist()
    tot = int(stats[0])
    return dict(pairs=len(first), matches=tot, mean_quality=(stats[1] / tot if tot else None),
                min_quality=(stats[2] if tot else None), max_quality=(stats[3] if tot else None), high_quality=int(stats[4]))


def _push_vit_groups(pipe: SequencePipeline, seq: "StreamingSequence", feeder: "FrameFeeder", n: int, chunk: int,
                     group: int | None = None):
    """ViT-inside feed loop: the ViT runs chunk by chunk as the uploads arrive (its launch group, 82 frames at 448 x 448), but
    the authored stages run over GROUPS of chunks - an 82-frame extraction is 1.67 workgroup rounds of the descriptor MLP and
    one partial round of the conv; half a sequence at a time they run at their large-launch rates.  Tokens of a group collect in
    one buffer; its pixels are one contiguous view of the feeder's sequence-sized device buffer (ring mode, i.e. sequences
    beyond FrameFeeder.max_bytes: chunk by chunk as before)."""
    cur = torch.cuda.current_stream(pipe.device)
    if group is None:
        group = min(pipe.launch_group(), max(2 * chunk, -(-((n + 1) // 2) // chunk) * chunk))
    grouped = feeder.whole and group > chunk
    t_tok = N_PREFIX + pipe.cfg.grid ** 2
    tok, off, g0 = None, 0, 0
    for a, b, img, ready in feeder:
        cur.wait_event(ready)
        if not grouped:
            seq.push(pipe.tokens_from_images(img, batch_frames=n), img)
            continue
        if tok is None:
            tok = torch.empty((group, t_tok, lib.C_FEAT), dtype=torch.float32, device=pipe.device)
        if off == 0:
            g0 = a
        pipe.tokens_from_images(img, out=tok[off:off + b - a], batch_frames=n)
        off += b - a
        if b == n or off + chunk > group:
            seq.push(tok[:off], feeder.dev[g0:b])
            off = 0


def run_frames(pipe: SequencePipeline, n: int, h: int, w: int, spacings=(1,), tokens: torch.Tensor | None = None, fill=None,
               pinned_source: torch.Tensor | None = None, chunk: int | None = None, first_chunk: int | None = None,
               preprocess_too: bool = False, feeder_kw: dict | None = None) -> dict:
    """Host-resident frames -> matches, with the upload of chunk i + 1 overlapping the compute of chunk i.
    tokens: device-resident ViT tokens of the n frames (tokens-in mode); None: the pipeline's HIP ViT computes them
    (pipe built with vit=).  preprocess_too: in tokens-in mode also run A0 on every chunk (the ViT input a real backbone would
    consume; bench.py counts it in `value`).  Returns StreamingSequence.result() (sequence-sized buffers, written in place)."""
    if tokens is None and pipe.vit_hip is None:
        raise ValueError("tokens, or a pipeline built with vit=, required")
    if chunk is None:
        # ViT inside: whole launch groups of the ViT (82 frames at 448 x 448) - every stage needs the pixels, but the ViT
        # dwarfs the upload.  Tokens in: A2..A7 of a chunk read only tokens and run while ITS pixels are still uploading, so
        # large chunks win (fewer partial rounds of tiles, tools/upload_sweep.py: 613 frames in chunks of 83 / 167 / 307 /
        # 613 -> 47.8 / 51.6 / 54.5 / 55.8 k frames/s): half the sequence, within one launch group
        if tokens is None:
            chunk = pipe.vit_hip.chunk_frames(pipe.cfg.input_size)
        else:
            chunk = max(1, min(pipe.launch_group(), max((n + 1) // 2, (1024 * 128) // pipe.cfg.grid ** 2)))
    seq = StreamingSequence(pipe, spacings)
    seq.reset(capacity=n)
    feeder = FrameFeeder(n, h, w, pipe.device, chunk_bounds(n, chunk, first_chunk), fill=fill, pinned_source=pinned_source,
                         **(feeder_kw or {}))
    if tokens is None:
        _push_vit_groups(pipe, seq, feeder, n, chunk)
        return seq.result()
    for a, b, img, ready in feeder:
        # the pixels are first read by A9, the last stage of the extraction: A2..A7 run while the upload is in flight
        seq.push(tokens[a:b], img, images_ready=ready)
        if preprocess_too:
            pipe.preprocess(img, reuse=True)
    return seq.result()


def run_directory(root: str, sequence: str = "", spacings=(1, 5, 10, 15, 20), pipe: SequencePipeline | None = None,
                  selector_state: dict | None = None, refiner_state: dict | None = None, cfg: ExtractorConfig | None = None,
                  vit=None, tokens_fn=None, max_frames: int | None = None, chunk: int | None = None,
                  decode_workers: int | None = None, device="cuda") -> dict:
    """A TUM RGB-D sequence directory -> matches for every requested spacing: the batched counterpart of the reference's
    main() -> process_spacing() -> extract(path) -> match loop (visualize_matches_sequence.py:272-357, 360-448) over the
    directory layout of data/tum_dataset.py:210-224 (rgb/*.png sorted by name).

    TUMSequence lists the frames; a thread pool decodes the PNGs of chunk i + 1 (PIL, 'RGB' as at :71) into a pinned
    staging buffer while chunk i is uploaded on a side stream and extracted + matched on the compute stream.
    Tokens come from the HIP ViT (`vit`: an sslam_amd.vit.DinoV3ViT holding the weights) or from `tokens_fn(a, b) ->
    (b - a, 5 + G^2, 384) device tensor` (any other backbone).  Returns StreamingSequence.result() plus 'files'."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    from PIL import Image

    from .tum import TUMSequence
    tum = TUMSequence(root, sequence, max_frames=max_frames)
    n = len(tum)
    if n == 0:
        raise ValueError(f"no rgb frames under {tum.rgb_dir}")
    if pipe is None:
        pipe = SequencePipeline(cfg or ExtractorConfig(), selector_state, refiner_state, device=device, vit=vit)
    if tokens_fn is None and pipe.vit_hip is None:
        raise ValueError("vit= (HIP ViT) or tokens_fn required")
    with Image.open(tum.rgb_path(0)) as im0:
        w, h = im0.size
    workers = decode_workers or max(1, min(16, (os.cpu_count() or 2) - 1))
    pool = ThreadPoolExecutor(max_workers=workers)

    def decode(i, dst):
        with Image.open(tum.rgb_path(i)) as im:
            arr = np.asarray(im.convert("RGB"))
        if arr.shape != dst.shape:
            raise ValueError(f"{tum.rgb_files[i]}: {arr.shape[:2]} differs from the first frame's {(h, w)}")
        dst[...] = arr

    def fill(host_rows, a, b):
        list(pool.map(lambda j: decode(a + j, host_rows[j]), range(b - a)))

    if chunk is None:
        chunk = pipe.vit_hip.chunk_frames(pipe.cfg.input_size) if tokens_fn is None else 64
    seq = StreamingSequence(pipe, spacings)
    seq.reset(capacity=n)
    try:
        with torch.no_grad():
            feeder = FrameFeeder(n, h, w, pipe.device, chunk_bounds(n, chunk, min(chunk, 16)), fill=fill)
            if tokens_fn is None:
                _push_vit_groups(pipe, seq, feeder, n, chunk)
            else:
                for a, b, img, ready in feeder:
                    seq.push(tokens_fn(a, b), img, images_ready=ready)
    finally:
        pool.shutdown(wait=True)
    res = seq.result()
    res["files"] = list(tum.rgb_files)
    res["timestamps"] = list(tum.timestamps)
    return res

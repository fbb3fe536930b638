"""The ONLINE caller's step - one frame in, its features and the matches against the previous frame out - on static buffers,
optionally replayed from a HIP graph.

Every caller the reference has works frame by frame (B = 1): `visualize_matches_sequence.py:306-357` extracts and matches pair
by pair, `test/test_tracking.py:146-178` keeps the previous frame's descriptors and matches each new frame against them,
`test/test_performance.py:89-131` times one image at a time (the "143 FPS" of its notes is that protocol).  `FrameStepper` is
that loop on the HIP path: the frame is copied into a static buffer, A0 -> A1 -> A2 .. A9 run on it, M1 matches it against the
previous frame's descriptors (kept on the device), and the frame's descriptors become the previous ones - no concatenation, no
allocation per frame, no host synchronisation (the caller reads what it wants from the returned views).

`use_graph=True` captures that step ONCE with `torch.cuda.graph` (hipGraph underneath) and replays it: one graph launch instead
of 8 (tokens in) / 72 (bf16 ViT inside) / 96 (fp32 ViT inside) library calls.  The C ABI was shaped for this (no allocation, no
synchronisation, no host read-back inside the library; caller-owned workspace), and the replay is bit-identical - but MEASURED it
buys nothing on this chip: 0.297 against 0.289 ms per frame with tokens in, 1.215 against 1.209 ms with the bf16 ViT inside,
2.12 against 2.11 ms with the fp32 ViT (tools/online_probe.py).  At B = 1 the step is bound by the DURATION of its ~70 dependent
kernels (a frame is 7 row tiles: most launches occupy a few dozen of the 256 CUs for 5-20 us each), not by the host's launch rate -
the host is already ahead of the device.  It is kept as an option because it takes the host out of the loop (0 library calls per
frame), which matters to a caller that has other work for its CPU thread.

Results are the kernels' results: bit-identical to the same frames going through `SequencePipeline.run` as one batch
(tests/test_gpu_harness.py::test_online_stepper_*).
"""
from __future__ import annotations

import torch

from . import lib
from .pipeline import N_PREFIX, SequencePipeline


class FrameStepper:
    def __init__(self, pipe: SequencePipeline, height: int, width: int, use_graph: bool = True, tokens_in: bool = False):
        """pipe: a SequencePipeline (with vit= unless tokens_in).  height / width: the frames' size (uint8 RGB).
        tokens_in: the caller brings the ViT's tokens with every frame (the third-party ViT stays outside).
        use_graph=False: the same step as ordinary launches (the A/B for the graph, and the fallback while debugging)."""
        cfg = pipe.cfg
        if cfg.num_keypoints > cfg.grid ** 2:
            raise ValueError("num_keypoints > grid cells: that case reads a status word back on the host (SURVEY H6) and cannot be captured")
        if not tokens_in and pipe.vit_hip is None:
            raise lib.SslamHipError("this pipeline was built without a ViT: pass tokens_in=True, or construct it with vit=")
        self.pipe, self.cfg, self.device = pipe, cfg, pipe.device
        self.tokens_in, self.use_graph = tokens_in, use_graph
        dev, K = self.device, cfg.num_keypoints
        self.image = torch.zeros((1, height, width, 3), dtype=torch.uint8, device=dev)
        self.tokens = torch.zeros((1, N_PREFIX + cfg.grid ** 2, lib.C_FEAT), dtype=torch.float32, device=dev)
        # slot 0: the previous frame, slot 1: this frame - the matcher's pair (0, 1) without any concatenation
        self.pair = pipe.alloc_extract(2, True)
        for v in self.pair.values():
            v.zero_()
        self.cur = {k: v[1:2] for k, v in self.pair.items()}
        self.m = pipe.alloc_match(1, K)
        self.n_frames = 0
        self._graph = None
        self._aux = None
        self._held = None

    # the captured region: only launches on the current stream, static buffers on both sides
    def _body(self) -> None:
        p = self.pipe
        if not self.tokens_in:
            p.tokens_from_images(self.image, out=self.tokens)
        p.extract(self.tokens, self.image, out=self.cur)
        self._aux = p.match(self.pair["descriptors"], self.pair["scores"], self.pair["intensity"], spacing=1, out=self.m)
        for k in ("descriptors", "scores", "intensity", "keypoints_pixel"):      # this frame becomes the previous one
            self.pair[k][0].copy_(self.pair[k][1])

    def _capture(self) -> None:
        # warm-up outside the capture: resampling tables, RoPE tables, workspaces and the ViT's buffers are created on first use
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        # The graph bakes in the ADDRESSES of buffers the stepper does not own: the pipeline's scratch, the ViT's workspaces, the
        # resampling and RoPE tables.  Their owners grow them by replacement (a later, larger pipe.run / tokens_from_images drops
        # the old tensor), which would leave the graph reading and writing freed memory.  The stepper therefore keeps its own
        # reference to every one of them from the warm-up on: a replaced buffer stays alive - and exclusively the graph's - for
        # as long as the graph does.
        self._held = self._external_buffers()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self._graph = g

    def _external_buffers(self) -> list:
        p, held = self.pipe, []
        held.append(p._ws)
        held.extend(p._stage.values())
        held.extend(p.tables._cache.values())
        vh = p.vit_hip
        if vh is not None:
            held.extend(getattr(vh, "_side_ws", []) or [])
            held.extend(getattr(vh, "_ws", []) or [])
            held.extend(vh._rope.values())
            held.append(vh._keep)
        return held

    @torch.no_grad()
    def step(self, image_u8: torch.Tensor, tokens: torch.Tensor | None = None) -> dict:
        """image_u8: (H, W, 3) or (1, H, W, 3) uint8, on the device or in host memory (pinned: the copy is asynchronous).
        tokens: (T, 384) / (1, T, 384) fp32 when the stepper was built with tokens_in.  Returns this frame's saliency /
        keypoints_pixel / scores / idx / descriptors / intensity (views of static buffers, (K, ...) without the batch axis) and,
        from the second frame on, matches (K, 2) int64 / quality (K,) / match_count against the previous frame (None before)."""
        if self.use_graph and self._graph is None:
            self._capture()                      # runs the body on whatever the buffers hold; the first real frame has no previous one
        self.image.copy_(image_u8.reshape(self.image.shape), non_blocking=True)
        if self.tokens_in:
            if tokens is None:
                raise ValueError("this stepper takes the frame's tokens (tokens_in=True)")
            self.tokens.copy_(tokens.reshape(self.tokens.shape), non_blocking=True)
        if self.use_graph:
            self._graph.replay()
        else:
            self._body()
        first = self.n_frames == 0
        self.n_frames += 1
        out = {k: v[0] for k, v in self.cur.items()}
        out["matches"] = None if first else self.m["matches"][0]
        out["quality"] = None if first else self.m["quality"][0]
        out["match_count"] = None if first else self.m["match_count"][0]
        return out

    def reset(self) -> None:
        """Forget the previous frame (the next step returns no matches)."""
        self.n_frames = 0

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
from .pipeline import ExtractorConfig, SequencePipeline


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
    """Streaming pair scheduler.  Frames arrive in chunks (`push`); every frame is extracted once; a
    ring of the last max(spacings) frames' descriptors / scores / intensities stays on the device, and each push
    matches, for every spacing s, exactly the pairs (j - s, j) whose SECOND frame j arrived in that push - so over a
    whole sequence every pair (i, i + s) is matched once, whatever the chunking.

    The reference's process_spacing (visualize_matches_sequence.py:298-300) visits only i = 0, s, 2s, ... and stops
    after `max_pairs` pairs: `reference_pairs()` selects those rows from the result."""

    def __init__(self, pipe: SequencePipeline, spacings=(1, 5, 10, 15, 20)):
        self.pipe, self.spacings = pipe, tuple(int(s) for s in spacings)
        if not self.spacings or min(self.spacings) < 1:
            raise ValueError("spacings must be positive")
        self.reset()

    def reset(self):
        self.n_seen = 0
        self._ring = None        # dict of (r, ...) tensors: the last r <= max(spacings) frames

    def push(self, tokens: torch.Tensor, images_u8: torch.Tensor | None = None) -> dict:
        """Extract `tokens.shape[0]` new frames and match them against the ring.  Returns
        {'frames': extract dict of the new frames, s: {'first': global index of each pair's first frame (device int64),
        'matches', 'quality', 'match_count', ...}} for every spacing that has a pair ending in this chunk."""
        ex = self.pipe.extract(tokens, images_u8)
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
        self.reset()
        n = tokens.shape[0]
        step = n if chunk is None else int(chunk)
        outs = [self.push(tokens[a:a + step], None if images_u8 is None else images_u8[a:a + step])
                for a in range(0, n, step)]
        res = {"frames": {k: torch.cat([o["frames"][k] for o in outs]) for k in outs[0]["frames"]}}
        for s in self.spacings:
            rows = [o[s] for o in outs if s in o]
            if rows:
                res[s] = {k: torch.cat([r_[k] for r_ in rows]) for k in rows[0]}
        return res

    @staticmethod
    def reference_pairs(n_frames: int, spacing: int, max_pairs: int | None = None) -> list:
        """First-frame indices process_spacing visits: range(0, n - spacing, spacing), at most max_pairs of them."""
        idx = list(range(0, n_frames - spacing, spacing))
        return idx if max_pairs is None else idx[:max_pairs]

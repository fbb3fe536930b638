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
        tokens = self.backbone.dino.forward_features(self.pipe.preprocess(img)).float().contiguous()
        out = self.pipe.extract(tokens, img)
        return {"saliency": out["saliency"], "keypoints_pixel": out["keypoints_pixel"], "scores": out["scores"],
                "intensity": out["intensity"], "descriptors": out["descriptors"]}

    def extract(self, image_path: str) -> dict:
        """One image -> numpy dict with the reference's keys (visualize_matches_sequence.py:97-104)."""
        from PIL import Image
        image = Image.open(image_path).convert("RGB")
        out = self.extract_batch(np.asarray(image)[None])
        res = {k: v[0].cpu().numpy() for k, v in out.items()}
        res["image"] = image
        return res

    match_with_quality = staticmethod(matching.match_with_quality)


class StreamingSequence:
    def __init__(self, pipe: SequencePipeline, spacings=(1, 5, 10, 15, 20)):
        self.pipe, self.spacings = pipe, tuple(spacings)

    def run(self, tokens: torch.Tensor, images_u8: torch.Tensor | None) -> dict:
        """Extract once, then M1 for every pair (i, i + s), s in spacings.  Returns {'frames': ..., s: match dict}."""
        ex = self.pipe.extract(tokens, images_u8)
        res = {"frames": ex}
        for s in self.spacings:
            if tokens.shape[0] > s:
                res[s] = self.pipe.match(ex["descriptors"], ex["scores"], ex.get("intensity"), spacing=s)
        return res

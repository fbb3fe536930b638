"""Drop-in for the reference's `models.keypoint_selector` (semantic-slam/models/keypoint_selector.py).

Same class name, constructor, sub-module layout (`conv.0`, `conv.2` -> identical state_dict keys), methods and
return conventions.  Execution:

* CUDA tensor and no autograd graph needed (`torch.no_grad()` / eval scripts)  ->  hand-written HIP kernels
  (libsslam_hip.so: sslam_selector_saliency, sslam_select_keypoints).  If the library is missing this RAISES.
* autograd needed (train.py back-propagates through the saliency CNN) or a CPU tensor  ->  ordinary
  torch ops on the same Parameters, so AdamW / clip_grad_norm_ / .to() / state_dict() behave as in the reference.
"""
from __future__ import annotations

import warnings
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from sslam_amd import lib
from sslam_amd.pipeline import PackedSelector


def _needs_graph(module: nn.Module, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    return any(t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


class KeypointSelector(nn.Module):
    def __init__(self, input_dim: int = 384, hidden_dim: int = 128):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(input_dim, hidden_dim, kernel_size=3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(hidden_dim, 1, kernel_size=1),
        )
        self._packed = None
        self._packed_key = None
        self._warned = False
        self._init_weights()

    def _init_weights(self):
        # keypoint_selector.py:38-43
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=0.5)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0.0)

    # ------------------------------------------------------------------------------------------ HIP plumbing
    def _packed_weights(self) -> PackedSelector:
        ps = list(self.conv.parameters())
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        if self._packed is None or key != self._packed_key:
            self._packed = PackedSelector({k: v for k, v in self.conv.state_dict(prefix="conv.").items()}, ps[0].device)
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------------------------------------ API
    def _hip_ok(self, x: torch.Tensor) -> bool:
        """HIP path only for the shapes the kernels are built for and when weights and input share a GPU; any other
        `hidden_dim` / `input_dim` the reference accepts (keypoint_selector.py:22-36) runs as eager torch ops."""
        w = self.conv[0].weight
        if not PackedSelector.supported(w.shape) or x.shape[-1] != lib.C_FEAT:
            if not self._warned:
                warnings.warn(f"KeypointSelector: Conv2d weight {tuple(w.shape)} is outside the HIP kernels' shapes "
                              f"(384 -> 128 | 256); using the eager torch path")
                self._warned = True
            return False
        return w.device == x.device      # mismatch: let the torch ops below raise their usual device error

    def forward(self, dino_features: torch.Tensor) -> torch.Tensor:
        """(B, H, W, C) patch features -> (B, H, W, 1) saliency in [0, 1]  (keypoint_selector.py:45-67)."""
        if dino_features.is_cuda and not _needs_graph(self, dino_features) and self._hip_ok(dino_features):
            pk = self._packed_weights()
            x = dino_features.detach().contiguous().float()
            sal = lib.selector_saliency(x, pk.w1p, pk.b1, pk.w2, pk.b2, pk.hidden)
            return sal.unsqueeze(-1)
        x = dino_features.permute(0, 3, 1, 2)
        return torch.sigmoid(self.conv(x)).permute(0, 2, 3, 1)

    def select_keypoints(self, saliency_map: torch.Tensor, num_keypoints: int = 500, nms_radius: int = 2,
                         min_score_percentile: float = 0.50) -> Tuple[torch.Tensor, torch.Tensor]:
        """(B, H, W, 1) -> keypoints (B, N, 2) fp32 (x, y) in patch units, scores (B, N)  (keypoint_selector.py:69-207).

        Order among equal values is value-descending / flat-index-ascending (torch.topk leaves it unspecified)."""
        B, H, W, _ = saliency_map.shape
        if H != W:
            raise lib.SslamHipError("square patch grids only (DinoBackbone always produces grid_h == grid_w)")
        sal = saliency_map.squeeze(-1)
        if sal.is_cuda:
            kp, sc, idx, _, st = lib.select_keypoints(sal.detach().contiguous().float(), num_keypoints, nms_radius,
                                                      min_score_percentile, want_pixel=False)
            if num_keypoints > H * W and bool(st.any()):     # only then can torch.topk have raised
                raise RuntimeError("selected index k out of range")
            if torch.is_grad_enabled() and saliency_map.requires_grad:
                sc = sal.reshape(B, -1).gather(1, idx.long())  # same values, attached to the autograd graph
            return kp, sc
        return _select_keypoints_eager(sal, num_keypoints, nms_radius, min_score_percentile)

    def _apply_nms(self, saliency: torch.Tensor, radius: int) -> torch.Tensor:
        """(B, H, W) -> (B, H, W): keep exact local maxima of the (2r+1)^2 window (keypoint_selector.py:209-226)."""
        if radius == 0:
            return saliency
        pooled = F.max_pool2d(saliency.unsqueeze(1), kernel_size=2 * radius + 1, stride=1, padding=radius).squeeze(1)
        return saliency * (saliency == pooled).float()


def _topk_canonical(values: torch.Tensor, flat_index: torch.Tensor, k: int):
    """top-k by (value desc, flat index asc)."""
    if k > values.numel():
        raise RuntimeError("selected index k out of range")
    order = torch.argsort(flat_index, stable=True)
    order = order[torch.argsort(values[order], descending=True, stable=True)]
    return order[:k]


def _select_keypoints_eager(sal: torch.Tensor, K: int, radius: int, pct: float):
    """Autograd-/CPU-side restatement of the selection rule with plain torch ops (one frame at a time: the
    control flow is data dependent)."""
    B, H, W = sal.shape
    nms_all = sal if radius == 0 else sal * (sal == F.max_pool2d(sal.unsqueeze(1), 2 * radius + 1, 1, radius).squeeze(1)).float()
    kps, scs = [], []
    cells = torch.arange(H * W, device=sal.device)
    for b in range(B):
        raw, nms = sal[b].reshape(-1), nms_all[b].reshape(-1)
        thr = max(torch.quantile(raw, pct).item(), 0.1)
        valid = nms > thr
        nv = int(valid.sum())
        if nv >= K:
            sel = cells[valid][_topk_canonical(nms[valid], cells[valid], K)]
            sc = nms[sel]
        elif nv > 0:
            sel, sc = cells[valid], nms[valid]
            remaining = K - nv
            for p in (0.40, 0.30, 0.20, 0.10):
                lower = max(torch.quantile(raw, p).item(), 0.05)
                extra = (nms > lower) & ~valid
                if int(extra.sum()) >= remaining:
                    e = cells[extra][_topk_canonical(nms[extra], cells[extra], remaining)]
                    sel, sc = torch.cat([sel, e]), torch.cat([sc, nms[e]])
                    break
            else:
                e = _topk_canonical(raw, cells, remaining)
                sel, sc = torch.cat([sel, e]), torch.cat([sc, raw[e]])
        else:
            sel = _topk_canonical(raw, cells, K)
            sc = raw[sel]
        kps.append(torch.stack([sel % W, sel // W], dim=1).float())
        scs.append(sc)
    return torch.stack(kps), torch.stack(scs)

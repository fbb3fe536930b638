"""Drop-in for the reference's `models.descriptor_refiner` (semantic-slam/models/descriptor_refiner.py).

Same classes (`DescriptorRefiner`, `ResidualBlock`), constructor and state_dict keys
(`input_proj`, `residual_blocks.{i}.{norm1,fc1,norm2,fc2}`, `output_proj`).  Under `torch.no_grad()` on a CUDA tensor
the whole MLP runs as ONE fused HIP kernel (sslam_refine); with autograd enabled (train.py) or on CPU
tensors it runs as ordinary torch ops on the same Parameters.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from sslam_amd import lib
from sslam_amd.pipeline import PackedRefiner


class ResidualBlock(nn.Module):
    """x -> ReLU(fc2(LN(ReLU(fc1(LN(x))))) + x)   (descriptor_refiner.py:94-126)."""

    def __init__(self, dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.fc2 = nn.Linear(dim, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = F.relu(self.fc1(self.norm1(x)))
        h = self.fc2(self.norm2(h))
        return F.relu(h + x)


class DescriptorRefiner(nn.Module):
    def __init__(self, input_dim: int = 384, hidden_dim: int = 384, output_dim: int = 128, num_layers: int = 4):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.input_proj = nn.Linear(input_dim, hidden_dim)
        self.residual_blocks = nn.ModuleList([ResidualBlock(hidden_dim) for _ in range(num_layers - 2)])
        self.output_proj = nn.Linear(hidden_dim, output_dim)
        self._packed = None
        self._packed_key = None
        self._warned = False
        self._init_weights()

    def _init_weights(self):
        # descriptor_refiner.py:47-56
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight, gain=1.0)
                if m.bias is not None:
                    nn.init.uniform_(m.bias, -0.1, 0.1)

    def _packed_weights(self) -> PackedRefiner:
        ps = list(self.parameters())
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        if self._packed is None or key != self._packed_key:
            self._packed = PackedRefiner(self.state_dict(), ps[0].device)
            self._packed_key = key
        return self._packed

    def _hip_ok(self, x: torch.Tensor) -> bool:
        """HIP path only for 384 -> 384 -> 128 (any other dims the reference accepts, descriptor_refiner.py:22-45, run
        as eager torch ops) and when weights and input share a GPU."""
        w_in, w_out = self.input_proj.weight, self.output_proj.weight
        if not PackedRefiner.supported(w_in.shape[1], w_in.shape[0], w_out.shape[0], len(self.residual_blocks)):
            if not self._warned:
                warnings.warn(f"DescriptorRefiner: {w_in.shape[1]} -> {w_in.shape[0]} -> {w_out.shape[0]} is outside the HIP "
                              f"kernel's shape (384 -> 384 -> 128); using the eager torch path")
                self._warned = True
            return False
        return w_in.device == x.device

    def forward(self, dino_features: torch.Tensor) -> torch.Tensor:
        """(B, N, C) features at keypoints -> (B, N, output_dim) L2-normalised descriptors (descriptor_refiner.py:58-91)."""
        B, N, C = dino_features.shape
        needs_graph = torch.is_grad_enabled() and (dino_features.requires_grad or any(p.requires_grad for p in self.parameters()))
        if dino_features.is_cuda and not needs_graph and self._hip_ok(dino_features):
            pk = self._packed_weights()
            x = dino_features.detach().contiguous().float().reshape(B * N, C)
            return lib.refine(x, pk.packed, pk.n_blocks).reshape(B, N, self.output_dim)
        x = F.relu(self.input_proj(dino_features.reshape(B * N, C)))
        for block in self.residual_blocks:
            x = block(x)
        return F.normalize(self.output_proj(x), p=2, dim=-1).reshape(B, N, self.output_dim)

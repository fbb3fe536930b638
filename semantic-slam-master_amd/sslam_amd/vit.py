"""DINOv3 ViT-S/16 forward: the backbone the reference obtains from the third-party `timm`
package (semantic-slam/models/dino_backbone.py:44-48, 85), restated from the public architecture so that
`DinoBackbone` can be constructed without `timm` and without network access.

Architecture (DINOv3, ViT-S/16): 16x16 patch embedding; [CLS] + 4 register tokens; 12 pre-LN blocks, each
{LayerNorm -> MHA(6 heads x 64, q/v/proj bias, no k bias, axial RoPE on the patch tokens, theta = 100) -> LayerScale
-> +residual -> LayerNorm -> MLP(384 -> 1536 GELU -> 384) -> LayerScale -> +residual}; final LayerNorm.
RoPE: patch-centre coordinates in [-1, 1] (y, x), angles = 2*pi*coord*theta^(-4i/64), i = 0..15, laid out
[y-freqs, x-freqs] and tiled twice; q' = q*cos + rotate_half(q)*sin.

This module is the eager (torch-op) definition: it carries the Parameters, loads weights from a local file and is
the fp32 reference that the HIP ViT (later round) is checked against.  Parity of this restatement is pinned in
tests/test_vit.py against `transformers.DINOv3ViTModel` built from config with random weights - the same
architecture family, no weights fetched.  It says nothing about the reference's *pretrained* weights, which are a
remote fetch and unavailable offline ("parity unpinned" for A1 with real weights).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_dim: int, eps: float):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.q_proj = nn.Linear(dim, dim, bias=True)
        self.k_proj = nn.Linear(dim, dim, bias=False)
        self.v_proj = nn.Linear(dim, dim, bias=True)
        self.o_proj = nn.Linear(dim, dim, bias=True)
        self.ls1 = nn.Parameter(torch.ones(dim))
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.up_proj = nn.Linear(dim, mlp_dim)
        self.down_proj = nn.Linear(mlp_dim, dim)
        self.ls2 = nn.Parameter(torch.ones(dim))

    def forward(self, x, cos, sin, n_prefix):
        B, T, C = x.shape
        hd = C // self.heads
        h = self.norm1(x)
        q = self.q_proj(h).view(B, T, self.heads, hd).transpose(1, 2)
        k = self.k_proj(h).view(B, T, self.heads, hd).transpose(1, 2)
        v = self.v_proj(h).view(B, T, self.heads, hd).transpose(1, 2)

        def rope(t):
            pre, pat = t[:, :, :n_prefix], t[:, :, n_prefix:]
            rot = torch.cat((-pat[..., hd // 2:], pat[..., : hd // 2]), dim=-1)
            return torch.cat((pre, pat * cos + rot * sin), dim=2)

        q, k = rope(q), rope(k)
        att = torch.softmax((q @ k.transpose(2, 3)) * (hd ** -0.5), dim=-1)
        o = (att @ v).transpose(1, 2).reshape(B, T, C)
        x = x + self.o_proj(o) * self.ls1
        x = x + self.down_proj(F.gelu(self.up_proj(self.norm2(x)))) * self.ls2
        return x


class DinoV3ViT(nn.Module):
    """Offers the two members DinoBackbone uses of a timm model: `.embed_dim` and `.forward_features(images)`."""

    def __init__(self, embed_dim: int = 384, depth: int = 12, heads: int = 6, mlp_dim: int = 1536, patch: int = 16,
                 n_register: int = 4, rope_theta: float = 100.0, eps: float = 1e-5):
        super().__init__()
        self.embed_dim, self.patch, self.n_register, self.heads = embed_dim, patch, n_register, heads
        self.rope_theta = rope_theta
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.register_tokens = nn.Parameter(torch.zeros(1, n_register, embed_dim))
        self.patch_embed = nn.Conv2d(3, embed_dim, kernel_size=patch, stride=patch)
        self.blocks = nn.ModuleList([_Block(embed_dim, heads, mlp_dim, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.trunc_normal_(self.register_tokens, std=0.02)

    def rope_tables(self, gh: int, gw: int, device, dtype=torch.float32):
        hd = self.embed_dim // self.heads
        inv_freq = 1.0 / self.rope_theta ** torch.arange(0, 1, 4 / hd, dtype=torch.float32, device=device)
        ys = (torch.arange(0.5, gh, dtype=torch.float32, device=device) / gh) * 2.0 - 1.0
        xs = (torch.arange(0.5, gw, dtype=torch.float32, device=device) / gw) * 2.0 - 1.0
        coords = torch.stack(torch.meshgrid(ys, xs, indexing="ij"), dim=-1).flatten(0, 1)      # (gh*gw, 2): (y, x)
        ang = (2 * math.pi * coords[:, :, None] * inv_freq[None, None, :]).flatten(1, 2).tile(2)  # (gh*gw, hd)
        return torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)

    def forward_features(self, images: torch.Tensor) -> torch.Tensor:
        """(B, 3, H, W) -> (B, 1 + n_register + (H/16)*(W/16), embed_dim), final-LayerNormed tokens."""
        B, _, H, W = images.shape
        gh, gw = H // self.patch, W // self.patch
        x = self.patch_embed(images).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(B, -1, -1), self.register_tokens.expand(B, -1, -1), x], dim=1)
        cos, sin = self.rope_tables(gh, gw, images.device, x.dtype)
        n_prefix = 1 + self.n_register
        for blk in self.blocks:
            x = blk(x, cos, sin, n_prefix)
        return self.norm(x)

    forward = forward_features

    # ------------------------------------------------------------------------------------------ weights
    def load_hf_state_dict(self, sd: dict):
        """Load weights keyed like `transformers.DINOv3ViTModel.state_dict()` (what the public DINOv3 safetensors
        checkpoints use), e.g. from a LOCAL file: `load_hf_state_dict(safetensors.torch.load_file(path))`."""
        m = {"cls_token": sd["embeddings.cls_token"], "register_tokens": sd["embeddings.register_tokens"],
             "patch_embed.weight": sd["embeddings.patch_embeddings.weight"],
             "patch_embed.bias": sd["embeddings.patch_embeddings.bias"],
             "norm.weight": sd["norm.weight"], "norm.bias": sd["norm.bias"]}
        for i in range(len(self.blocks)):
            s, d = f"model.layer.{i}.", f"blocks.{i}."
            if s + "norm1.weight" not in sd:
                s = f"layer.{i}."
            for a, b in [("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"),
                         ("attention.q_proj.weight", "q_proj.weight"), ("attention.q_proj.bias", "q_proj.bias"),
                         ("attention.k_proj.weight", "k_proj.weight"),
                         ("attention.v_proj.weight", "v_proj.weight"), ("attention.v_proj.bias", "v_proj.bias"),
                         ("attention.o_proj.weight", "o_proj.weight"), ("attention.o_proj.bias", "o_proj.bias"),
                         ("layer_scale1.lambda1", "ls1"), ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias"),
                         ("mlp.up_proj.weight", "up_proj.weight"), ("mlp.up_proj.bias", "up_proj.bias"),
                         ("mlp.down_proj.weight", "down_proj.weight"), ("mlp.down_proj.bias", "down_proj.bias"),
                         ("layer_scale2.lambda1", "ls2")]:
                m[d + b] = sd[s + a]
        missing, unexpected = self.load_state_dict(m, strict=True)
        return self

"""Stand-ins for third-party DINOv3 ViT modules (timm's, the upstream release's) in the tests of the weight conversion
(sslam_amd.vit.KEY_MAPS / convert_module): neither package is installed here, so the tests build state dicts with those
layouts - fused qkv, separate or fused-and-masked biases, other token names and shapes - from a random in-repo DinoV3ViT,
and a module (`ForeignViT`) that owns such parameters and computes forward_features with its OWN few lines of torch."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def random_vit(seed: int = 0, **kw):
    """A DinoV3ViT with every parameter family non-trivial (LayerScale, biases, norms, prefix tokens)."""
    from sslam_amd.vit import DinoV3ViT
    torch.manual_seed(seed)
    vit = DinoV3ViT(**kw).eval()
    with torch.no_grad():
        for n, p in vit.named_parameters():
            if n.endswith("ls1") or n.endswith("ls2"):
                p.copy_(0.5 + torch.rand_like(p))
            elif n.endswith("bias") or "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif "token" in n:
                p.copy_(0.5 * torch.randn_like(p))
            else:
                p.mul_(3.0)
    return vit


def foreign_state_dict(vit, layout: str) -> dict:
    """The weights of `vit` keyed and shaped like `layout` (a sslam_amd.vit.KEY_MAPS name)."""
    sd = {k: v.detach().clone() for k, v in vit.state_dict().items()}
    C = vit.embed_dim
    out = {}
    if layout.startswith("transformers"):
        pre = "model.layer.{i}." if layout == "transformers" else "layer.{i}."
        out.update({"embeddings.cls_token": sd["cls_token"], "embeddings.register_tokens": sd["register_tokens"],
                    "embeddings.mask_token": torch.zeros(1, 1, C),
                    "embeddings.patch_embeddings.weight": sd["patch_embed.weight"], "embeddings.patch_embeddings.bias": sd["patch_embed.bias"],
                    "norm.weight": sd["norm.weight"], "norm.bias": sd["norm.bias"]})
        names = {"norm1.weight": "norm1.weight", "norm1.bias": "norm1.bias", "q_proj.weight": "attention.q_proj.weight",
                 "q_proj.bias": "attention.q_proj.bias", "k_proj.weight": "attention.k_proj.weight", "v_proj.weight": "attention.v_proj.weight",
                 "v_proj.bias": "attention.v_proj.bias", "o_proj.weight": "attention.o_proj.weight", "o_proj.bias": "attention.o_proj.bias",
                 "ls1": "layer_scale1.lambda1", "norm2.weight": "norm2.weight", "norm2.bias": "norm2.bias", "up_proj.weight": "mlp.up_proj.weight",
                 "up_proj.bias": "mlp.up_proj.bias", "down_proj.weight": "mlp.down_proj.weight", "down_proj.bias": "mlp.down_proj.bias",
                 "ls2": "layer_scale2.lambda1"}
        for i in range(len(vit.blocks)):
            for a, b in names.items():
                out[pre.format(i=i) + b] = sd[f"blocks.{i}.{a}"]
        return out
    upstream = layout == "dinov3_upstream"
    assert upstream or layout == "timm_dinov3", layout
    out.update({"cls_token": sd["cls_token"], ("storage_tokens" if upstream else "reg_token"): sd["register_tokens"],
                "patch_embed.proj.weight": sd["patch_embed.weight"], "patch_embed.proj.bias": sd["patch_embed.bias"],
                "norm.weight": sd["norm.weight"], "norm.bias": sd["norm.bias"]})
    if upstream:
        out["mask_token"] = torch.zeros(1, C)
        out["rope_embed.periods"] = torch.ones(16)
    for i in range(len(vit.blocks)):
        b, d = f"blocks.{i}.", f"blocks.{i}."
        out[d + "attn.qkv.weight"] = torch.cat([sd[b + "q_proj.weight"], sd[b + "k_proj.weight"], sd[b + "v_proj.weight"]])
        if upstream:
            # a fused bias whose k part is NOT zero in the file but is masked at run time (bias_mask), as upstream stores it
            out[d + "attn.qkv.bias"] = torch.cat([sd[b + "q_proj.bias"], torch.full((C,), 0.37), sd[b + "v_proj.bias"]])
            out[d + "attn.qkv.bias_mask"] = torch.cat([torch.ones(C), torch.zeros(C), torch.ones(C)])
            out[d + "ls1.gamma"], out[d + "ls2.gamma"] = sd[b + "ls1"], sd[b + "ls2"]
        else:
            out[d + "attn.q_bias"], out[d + "attn.v_bias"] = sd[b + "q_proj.bias"], sd[b + "v_proj.bias"]
            out[d + "gamma_1"], out[d + "gamma_2"] = sd[b + "ls1"], sd[b + "ls2"]
        for a, f in [("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"), ("o_proj.weight", "attn.proj.weight"),
                     ("o_proj.bias", "attn.proj.bias"), ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias"),
                     ("up_proj.weight", "mlp.fc1.weight"), ("up_proj.bias", "mlp.fc1.bias"), ("down_proj.weight", "mlp.fc2.weight"),
                     ("down_proj.bias", "mlp.fc2.bias")]:
            out[d + f] = sd[b + a]
    return out


class _Attn(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.qkv = nn.Linear(C, 3 * C, bias=False)
        self.q_bias, self.v_bias = nn.Parameter(torch.zeros(C)), nn.Parameter(torch.zeros(C))
        self.proj = nn.Linear(C, C)


class _Mlp(nn.Module):
    def __init__(self, C, H):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(C, H), nn.Linear(H, C)


class _Blk(nn.Module):
    def __init__(self, C, H):
        super().__init__()
        self.norm1, self.attn, self.norm2, self.mlp = nn.LayerNorm(C, eps=1e-5), _Attn(C), nn.LayerNorm(C, eps=1e-5), _Mlp(C, H)
        self.gamma_1, self.gamma_2 = nn.Parameter(torch.ones(C)), nn.Parameter(torch.ones(C))


class _PE(nn.Module):
    def __init__(self, C, P):
        super().__init__()
        self.proj = nn.Conv2d(3, C, P, P)


class ForeignViT(nn.Module):
    """A module in the `timm_dinov3` parameter layout (fused qkv, q_bias / v_bias, gamma_1 / gamma_2, reg_token) with its own
    forward_features.  rope_theta != 100 makes it a model with the right parameter SHAPES but other arithmetic."""

    def __init__(self, C=384, depth=12, heads=6, H=1536, P=16, n_reg=4, rope_theta=100.0):
        super().__init__()
        self.embed_dim, self.heads, self.patch, self.rope_theta = C, heads, P, rope_theta
        self.cls_token, self.reg_token = nn.Parameter(torch.zeros(1, 1, C)), nn.Parameter(torch.zeros(1, n_reg, C))
        self.patch_embed = _PE(C, P)
        self.blocks = nn.ModuleList([_Blk(C, H) for _ in range(depth)])
        self.norm = nn.LayerNorm(C, eps=1e-5)

    @classmethod
    def from_vit(cls, vit, **kw):
        m = cls(vit.embed_dim, len(vit.blocks), vit.heads, vit.blocks[0].up_proj.out_features, vit.patch, vit.n_register, **kw)
        m.load_state_dict(foreign_state_dict(vit, "timm_dinov3"), strict=True)
        return m.eval()

    def forward_features(self, x):
        B, _, Hh, Ww = x.shape
        gh, gw, C, nh = Hh // self.patch, Ww // self.patch, self.embed_dim, self.heads
        hd = C // nh
        t = self.patch_embed.proj(x).flatten(2).transpose(1, 2)
        t = torch.cat([self.cls_token.expand(B, -1, -1), self.reg_token.expand(B, -1, -1), t], 1)
        npre = t.shape[1] - gh * gw
        inv = 1.0 / self.rope_theta ** torch.arange(0, 1, 4 / hd, dtype=torch.float32, device=x.device)
        ys = (torch.arange(0.5, gh, device=x.device) / gh) * 2 - 1
        xs = (torch.arange(0.5, gw, device=x.device) / gw) * 2 - 1
        co = torch.stack(torch.meshgrid(ys, xs, indexing="ij"), -1).flatten(0, 1)
        ang = (2 * math.pi * co[:, :, None] * inv[None, None, :]).flatten(1, 2).tile(2)
        cos, sin = ang.cos(), ang.sin()
        for b in self.blocks:
            h = b.norm1(t)
            bias = torch.cat([b.attn.q_bias, torch.zeros_like(b.attn.q_bias), b.attn.v_bias])
            q, k, v = F.linear(h, b.attn.qkv.weight, bias).view(B, -1, 3, nh, hd).permute(2, 0, 3, 1, 4)

            def rope(u):
                pre, pat = u[:, :, :npre], u[:, :, npre:]
                rot = torch.cat((-pat[..., hd // 2:], pat[..., :hd // 2]), -1)
                return torch.cat((pre, pat * cos + rot * sin), 2)
            o = F.scaled_dot_product_attention(rope(q), rope(k), v).transpose(1, 2).reshape(B, -1, C)
            t = t + b.attn.proj(o) * b.gamma_1
            t = t + b.mlp.fc2(F.gelu(b.mlp.fc1(b.norm2(t)))) * b.gamma_2
        return self.norm(t)

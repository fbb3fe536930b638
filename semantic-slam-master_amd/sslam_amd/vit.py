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
        name = "transformers" if "model.layer.0.norm1.weight" in sd else "transformers_flat"
        return self.load_mapped_state_dict(sd, KEY_MAPS[name])

    def load_mapped_state_dict(self, sd: dict, key_map: dict, fused_qkv: bool | None = None):
        """Load a state dict whose keys follow `key_map`: {canonical name -> foreign key}, per-layer entries with "{i}" for
        the layer index (KEY_MAPS holds the known layouts; the canonical names are this class's own parameter names with
        the `blocks.{i}.` prefix dropped).  fused_qkv (default: inferred from the map): the foreign model keeps one
        (3C, C) `qkv.weight` whose rows are [q; k; v], with its bias either fused too (`qkv.bias`, k part zero - DINOv3
        masks it) or as separate `q_bias` / `v_bias` vectors.  Shapes are checked; a non-zero k bias is refused (under
        RoPE it is not a softmax-invariant shift, and this architecture has none)."""
        if fused_qkv is None:
            fused_qkv = "qkv.weight" in key_map
        C = self.embed_dim

        def get(name, i=None):
            key = key_map[name].format(i=i) if i is not None else key_map[name]
            if key not in sd:
                raise KeyError(f"state dict has no {key!r} (canonical {name!r})")
            return sd[key]

        m, own = {}, self.state_dict()
        for name in TOP_KEYS:
            m[name] = get(name).reshape(own[name].shape)
        for i in range(len(self.blocks)):
            d = f"blocks.{i}."
            for name in LAYER_KEYS_COMMON:
                m[d + name] = get(name, i).reshape(own[d + name].shape)
            if fused_qkv:
                w = get("qkv.weight", i)
                if tuple(w.shape) != (3 * C, C):
                    raise ValueError(f"layer {i}: fused qkv weight {tuple(w.shape)}, expected {(3 * C, C)}")
                m[d + "q_proj.weight"], m[d + "k_proj.weight"], m[d + "v_proj.weight"] = w[:C], w[C:2 * C], w[2 * C:]
                if "qkv.bias" in key_map:
                    b = get("qkv.bias", i).reshape(3 * C)
                    mask_key = key_map.get("qkv.bias_mask")
                    if mask_key is not None and mask_key.format(i=i) in sd:
                        b = b * sd[mask_key.format(i=i)].reshape(3 * C).to(b.dtype)      # the official checkpoints mask the k part
                    qb, kb, vb = b[:C], b[C:2 * C], b[2 * C:]
                else:
                    qb, vb = get("q_bias", i).reshape(C), get("v_bias", i).reshape(C)
                    kb = sd.get(key_map["k_bias"].format(i=i)) if "k_bias" in key_map else None
                if kb is not None and bool((kb != 0).any()):
                    raise ValueError(f"layer {i}: non-zero k bias - not the DINOv3 attention this ViT implements")
                m[d + "q_proj.bias"], m[d + "v_proj.bias"] = qb, vb
            else:
                for name in LAYER_KEYS_QKV:
                    m[d + name] = get(name, i)
                if "k_proj.bias" in key_map and key_map["k_proj.bias"].format(i=i) in sd and \
                        bool((sd[key_map["k_proj.bias"].format(i=i)] != 0).any()):
                    raise ValueError(f"layer {i}: non-zero k bias - not the DINOv3 attention this ViT implements")
        self.load_state_dict({k: v.detach().to(torch.float32) for k, v in m.items()}, strict=True)
        return self

    @classmethod
    def from_state_dict(cls, sd: dict, key_map: dict | str | None = None, device=None) -> "DinoV3ViT":
        """A DinoV3ViT sized from the SHAPES in a foreign state dict and loaded from it.  key_map: a KEY_MAPS name, a map,
        or None = the first known layout whose keys are all present (detect_key_map).  Head dim 64, RoPE theta 100 and
        LayerNorm eps 1e-5 are DINOv3's constants (they leave no trace in a state dict); DinoBackbone verifies a converted
        MODULE numerically against its own forward before trusting it."""
        if key_map is None:
            name = detect_key_map(sd)
            if name is None:
                raise KeyError("no known DINOv3 key layout matches this state dict (KEY_MAPS: " + ", ".join(KEY_MAPS) + ")")
            key_map = KEY_MAPS[name]
        elif isinstance(key_map, str):
            key_map = KEY_MAPS[key_map]
        C = int(sd[key_map["norm.weight"]].numel())
        pw = sd[key_map["patch_embed.weight"]]
        depth = 0
        probe = key_map["norm1.weight"]
        while probe.format(i=depth) in sd:
            depth += 1
        if depth == 0 or C % 64 or pw.dim() != 4 or pw.shape[1] != 3 or pw.shape[2] != pw.shape[3]:
            raise ValueError("not a ViT with RGB square patches and 64-wide heads")
        vit = cls(embed_dim=C, depth=depth, heads=C // 64, mlp_dim=int(sd[key_map["up_proj.weight"].format(i=0)].shape[0]),
                  patch=int(pw.shape[2]), n_register=int(sd[key_map["register_tokens"]].numel()) // C)
        if device is not None:
            vit = vit.to(device)
        return vit.load_mapped_state_dict(sd, key_map).eval()


TOP_KEYS = ("cls_token", "register_tokens", "patch_embed.weight", "patch_embed.bias", "norm.weight", "norm.bias")
LAYER_KEYS_COMMON = ("norm1.weight", "norm1.bias", "o_proj.weight", "o_proj.bias", "ls1", "norm2.weight", "norm2.bias",
                     "up_proj.weight", "up_proj.bias", "down_proj.weight", "down_proj.bias", "ls2")
LAYER_KEYS_QKV = ("q_proj.weight", "q_proj.bias", "k_proj.weight", "v_proj.weight", "v_proj.bias")


def _hf_map(prefix: str) -> dict:
    m = {"cls_token": "embeddings.cls_token", "register_tokens": "embeddings.register_tokens",
         "patch_embed.weight": "embeddings.patch_embeddings.weight", "patch_embed.bias": "embeddings.patch_embeddings.bias",
         "norm.weight": "norm.weight", "norm.bias": "norm.bias", "k_proj.bias": prefix + "attention.k_proj.bias"}
    for ours, theirs in [("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"),
                         ("q_proj.weight", "attention.q_proj.weight"), ("q_proj.bias", "attention.q_proj.bias"),
                         ("k_proj.weight", "attention.k_proj.weight"),
                         ("v_proj.weight", "attention.v_proj.weight"), ("v_proj.bias", "attention.v_proj.bias"),
                         ("o_proj.weight", "attention.o_proj.weight"), ("o_proj.bias", "attention.o_proj.bias"),
                         ("ls1", "layer_scale1.lambda1"), ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias"),
                         ("up_proj.weight", "mlp.up_proj.weight"), ("up_proj.bias", "mlp.up_proj.bias"),
                         ("down_proj.weight", "mlp.down_proj.weight"), ("down_proj.bias", "mlp.down_proj.bias"),
                         ("ls2", "layer_scale2.lambda1")]:
        m[ours] = prefix + theirs
    return m


def _blocks_map(reg: str, ls1: str, ls2: str, bias: dict) -> dict:
    """The `blocks.{i}.attn.qkv / attn.proj / mlp.fc1 / mlp.fc2` family (fused qkv)."""
    m = {"cls_token": "cls_token", "register_tokens": reg, "patch_embed.weight": "patch_embed.proj.weight",
         "patch_embed.bias": "patch_embed.proj.bias", "norm.weight": "norm.weight", "norm.bias": "norm.bias",
         "norm1.weight": "blocks.{i}.norm1.weight", "norm1.bias": "blocks.{i}.norm1.bias",
         "qkv.weight": "blocks.{i}.attn.qkv.weight",
         "o_proj.weight": "blocks.{i}.attn.proj.weight", "o_proj.bias": "blocks.{i}.attn.proj.bias",
         "ls1": "blocks.{i}." + ls1, "norm2.weight": "blocks.{i}.norm2.weight", "norm2.bias": "blocks.{i}.norm2.bias",
         "up_proj.weight": "blocks.{i}.mlp.fc1.weight", "up_proj.bias": "blocks.{i}.mlp.fc1.bias",
         "down_proj.weight": "blocks.{i}.mlp.fc2.weight", "down_proj.bias": "blocks.{i}.mlp.fc2.bias", "ls2": "blocks.{i}." + ls2}
    m.update(bias)
    return m


# Known key layouts.  "transformers*" is pinned (tests/test_vit.py loads a real transformers.DINOv3ViTModel state dict
# through it).  The other two are written from the public layouts of the upstream DINOv3 release and of timm's DINOv3
# models and are UNVERIFIED against those packages - neither is installed here (parity unpinned); what IS tested is the
# mechanism (fused qkv, separate / fused / masked biases, token shapes) on state dicts of these shapes built from a random
# DinoV3ViT, and DinoBackbone checks every converted MODULE numerically against its own forward before using it.
KEY_MAPS = {
    "transformers": _hf_map("model.layer.{i}."),
    "transformers_flat": _hf_map("layer.{i}."),
    # upstream DINOv3 checkpoints (unverified): storage_tokens, fused qkv with a fused bias + bias_mask, ls{1,2}.gamma
    "dinov3_upstream": _blocks_map("storage_tokens", "ls1.gamma", "ls2.gamma",
                                   {"qkv.bias": "blocks.{i}.attn.qkv.bias", "qkv.bias_mask": "blocks.{i}.attn.qkv.bias_mask"}),
    # timm's vit_*_dinov3 (EVA-style blocks; unverified): reg_token, fused qkv weight, separate q_bias / v_bias, gamma_{1,2}
    "timm_dinov3": _blocks_map("reg_token", "gamma_1", "gamma_2",
                               {"q_bias": "blocks.{i}.attn.q_bias", "v_bias": "blocks.{i}.attn.v_bias", "k_bias": "blocks.{i}.attn.k_bias"}),
}


def detect_key_map(sd: dict) -> str | None:
    """Name of the first KEY_MAPS layout whose layer-0 and top-level keys are all in `sd`."""
    optional = ("k_bias", "k_proj.bias", "qkv.bias_mask")
    for name, km in KEY_MAPS.items():
        if all(v.format(i=0) in sd for k, v in km.items() if k not in optional):
            return name
    return None


def convert_module(module: nn.Module, probe_size: int = 64, tol: float = 1e-3):
    """A DinoV3ViT carrying the weights of `module` - any nn.Module whose state dict follows a KEY_MAPS layout and which
    offers forward_features(images) (timm's contract, dino_backbone.py:85) - or (None, reason).  The conversion is trusted
    only after a numeric check: both run one random (1, 3, probe_size, probe_size) image in fp32 (falling back to the
    module's own input size if it rejects that one) and their tokens must agree within `tol` relative - so a model with
    the same parameter shapes but other arithmetic (another RoPE convention, eps, activation) is never silently replaced."""
    try:
        sd = module.state_dict()
        ref = next(iter(sd.values()))
        vit = DinoV3ViT.from_state_dict(sd, device=ref.device)
    except (KeyError, ValueError, RuntimeError, StopIteration) as e:
        return None, f"{type(e).__name__}: {e}"
    was_training = module.training
    module.eval()
    try:
        with torch.no_grad():
            gen = torch.Generator(device="cpu").manual_seed(0)
            err = None
            for size in (probe_size, getattr(module, "img_size", None) or 448):
                size = size[0] if isinstance(size, (tuple, list)) else size
                x = torch.randn((1, 3, size, size), generator=gen).to(ref.device)
                try:
                    want = module.forward_features(x).float()
                except Exception as e:   # noqa: BLE001 - a fixed-size model rejects the small probe: try its own size
                    err = e
                    continue
                got = vit.forward_features(x)
                if got.shape != want.shape:
                    return None, f"token shape {tuple(want.shape)} vs {tuple(got.shape)}"
                rel = float((got - want).norm() / want.norm().clamp_min(1e-30))
                if not rel < tol:
                    return None, f"the module's tokens differ from the DINOv3 ViT definition by {rel:.2e} relative"
                return vit, f"verified on a {size}x{size} probe, rel {rel:.1e}"
            return None, f"forward_features failed on the probe: {err}"
    finally:
        module.train(was_training)

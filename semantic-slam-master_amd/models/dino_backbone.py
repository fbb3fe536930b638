"""Drop-in for the reference's `models.dino_backbone` (semantic-slam/models/dino_backbone.py).

Same class, constructor signature, attributes (`embed_dim, patch_size, grid_h, grid_w, num_patches,
n_storage_tokens, input_size, model_name`, sub-modules `dino`, `feature_norm`) and methods.

* `dino`: the reference fetches a pretrained timm model by name (dino_backbone.py:44-48) - third-party code and a
  remote download.  Here `timm` is used when it is importable; otherwise the in-repo DINOv3 ViT-S/16 definition
  (sslam_amd.vit.DinoV3ViT) is instantiated with random weights and a loud warning: load real weights from a LOCAL
  file with `backbone.dino.load_hf_state_dict(...)`.  A ready ViT can also be injected with `dino=`.
  Whatever module ends up in `self.dino` - timm's, a transformers-keyed one, any nn.Module with a known DINOv3 key
  layout (sslam_amd.vit.KEY_MAPS) - its WEIGHTS are converted to the in-repo definition at first use on a GPU, the
  conversion is verified numerically against the module's own forward_features, and A1 then runs on the HIP ViT
  (sslam_vit_forward).  Only when the conversion fails does the module's own eager forward run, with a warning that
  says A1 is not on the HIP kernels and why.  `self.dino` itself stays the caller's module.
* token drop + BatchNorm1d over tokens (dino_backbone.py:91-106) and the bilinear feature gather
  (extract_at_keypoints, :114-152) run as HIP kernels under `torch.no_grad()` on CUDA tensors, honouring
  `self.training` exactly as nn.BatchNorm1d does (the visualize_* scripts leave the backbone in train mode,
  the test/* scripts call .eval()); with autograd enabled or on CPU tensors they run as torch ops.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from sslam_amd import lib


class DinoBackbone(nn.Module):
    def __init__(self, model_name: str = "vit_small_patch16_dinov3.lvd1689m", input_size: int = 448, freeze: bool = True,
                 dino: nn.Module | None = None, vit_precision: str = "fp32"):
        """The first three arguments are the reference's (dino_backbone.py:25-30).  `vit_precision` says how the frozen
        ViT runs on a GPU under no_grad:
          "fp32"  (default - a script run unchanged keeps the reference's numerics) the HIP ViT-S/16 with fp32 operands on the
                  fp32 matrix pipe (sslam_vit_forward_f32): the REFERENCE'S numerics for A1 (its timm model is fp32), within
                  ~1e-6 relative of the eager torch evaluation at 2.3 x its rate (single frame: 1.8 ms against 4.9);
          "bf16"  the HIP ViT-S/16 with bf16 MFMA operands (fp32 accumulation, LayerNorm, softmax, residual): the throughput
                  form, 5 x the fp32 rate; tokens within rel 2.5e-2 / cos > 0.995 of the fp32 definition - keypoint and match
                  agreement with the fp32 path is MEASURED (tests/test_gpu_harness.py, bench.py
                  `with_vit.fp32_reference_numerics`: 99.8 % / 99.4 %), not bit-exact;
          "eager" the module's own torch forward (no HIP kernel for A1).
        Every entry point that needs tokens (forward(), harness.SequenceMatcher) goes through forward_tokens(), so they
        agree with each other."""
        super().__init__()
        if vit_precision not in ("bf16", "fp32", "eager"):
            raise ValueError(f"vit_precision must be 'bf16', 'fp32' or 'eager', got {vit_precision!r}")
        self.vit_precision = vit_precision
        self.model_name = model_name
        self.input_size = input_size
        self.patch_size = 16
        self.grid_h = input_size // self.patch_size
        self.grid_w = input_size // self.patch_size
        self.num_patches = self.grid_h * self.grid_w

        if dino is None:
            try:
                import timm  # third-party, optional
                dino = timm.create_model(model_name, pretrained=True, dynamic_img_size=True)
            except ImportError:
                from sslam_amd.vit import DinoV3ViT
                warnings.warn("timm is not installed: using the in-repo DINOv3 ViT-S/16 definition with RANDOM weights; "
                              "load pretrained weights from a local file via backbone.dino.load_hf_state_dict(...)")
                dino = DinoV3ViT()
        self.dino = dino
        self.embed_dim = self.dino.embed_dim
        self.n_storage_tokens = 4
        self.feature_norm = nn.BatchNorm1d(self.embed_dim, affine=True)
        if freeze:
            for p in self.dino.parameters():
                p.requires_grad = False
            self.dino.eval()

    def _is_frozen(self) -> bool:
        return not next(self.dino.parameters()).requires_grad

    # -------------------------------------------------------------------------------------------- forward
    def _hip_vit(self, images: torch.Tensor):
        """HIP execution of the ViT (bf16 or fp32 operands, by vit_precision); rebuilt when the ViT's parameters change.
        None for vit_precision == "eager" and unless the parameters live on the images' GPU (a CPU-resident module with CUDA images takes
        the eager path, which raises torch's usual device error).  `self.dino` is the in-repo definition, or ANY module
        whose weights convert to it (sslam_amd.vit.convert_module: known key layout, ViT-S/16 shapes, tokens verified
        against the module's own forward) - timm's model in the reference's setup (dino_backbone.py:44-48).  A module that
        does not convert runs its own eager forward, after one warning that names the reason."""
        from sslam_amd.vit import DinoV3ViT, convert_module
        from sslam_amd.vit_hip import HipViT, HipViTF32
        if self.vit_precision == "eager" or not images.is_cuda:
            return None
        ps = list(self.dino.parameters())
        if not ps or ps[0].device != images.device:
            return None
        key = (self.vit_precision,) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_hip_vit_key", None) != key:
            self._hip_vit_obj, self._hip_vit_key, self.hip_vit_status = None, key, None
            vit, why = (self.dino, "in-repo definition") if isinstance(self.dino, DinoV3ViT) else convert_module(self.dino)
            if vit is not None:
                try:
                    self._hip_vit_obj = (HipViT if self.vit_precision == "bf16" else HipViTF32)(vit, ps[0].device)
                except lib.SslamHipError as e:          # converts, but is not ViT-S/16 with 4 register tokens
                    why = str(e)
            self.hip_vit_status = why
            if self._hip_vit_obj is None:
                warnings.warn(f"DinoBackbone: A1 (the ViT) is NOT running on the HIP kernels - {type(self.dino).__name__} "
                              f"could not be converted to the in-repo DINOv3 ViT-S/16 ({why}); its own eager forward runs instead")
        return self._hip_vit_obj

    def forward_tokens(self, images: torch.Tensor) -> torch.Tensor:
        """(B, 3, H, W) -> (B, 1 + 4 + N, C) final-LayerNormed ViT tokens: the call at dino_backbone.py:85."""
        grad = self.training and not self._is_frozen()
        hv = None if (grad and torch.is_grad_enabled()) else self._hip_vit(images)
        if hv is not None:
            return hv.forward_features(images)
        with torch.set_grad_enabled(grad):
            return self.dino.forward_features(images)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """(B, 3, H, W) -> (B, grid_h, grid_w, embed_dim) patch features (dino_backbone.py:70-108)."""
        return self.tokens_to_features(self.forward_tokens(images))

    def tokens_to_features(self, features: torch.Tensor) -> torch.Tensor:
        """The part of forward() after the ViT call: (B, 1 + 4 + N, C) tokens -> (B, grid_h, grid_w, C)."""
        n_prefix = 1 + self.n_storage_tokens
        n_patch = features.shape[1] - n_prefix
        assert n_patch == self.num_patches, f"Expected {self.num_patches} patches, got {n_patch}"
        B = features.shape[0]
        bn = self.feature_norm
        needs_graph = torch.is_grad_enabled() and (features.requires_grad or bn.weight.requires_grad)
        if features.is_cuda and not needs_graph and self.embed_dim == lib.C_FEAT:
            tok = features.detach().contiguous().float()
            train = bn.training or bn.running_mean is None
            out, mean, var = lib.bn_tokens(tok, n_prefix, B, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                           bn.running_var, train, bn.eps)
            if bn.training and bn.track_running_stats:
                # what nn.BatchNorm1d does in train mode: momentum update with the UNBIASED batch variance
                n = B * n_patch
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_(mean[0], alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var[0] * (n / max(n - 1, 1)), alpha=mom)
            return out.reshape(B, self.grid_h, self.grid_w, self.embed_dim)
        patch_tokens = features[:, n_prefix:, :]
        B, N, C = patch_tokens.shape
        patch_tokens = bn(patch_tokens.reshape(B * N, C)).reshape(B, N, C)
        return patch_tokens.reshape(B, self.grid_h, self.grid_w, self.embed_dim)

    # ------------------------------------------------------------------------------------------- sampling
    def extract_at_keypoints(self, patch_features: torch.Tensor, keypoints: torch.Tensor) -> torch.Tensor:
        """(B, H, W, C) features, (B, N, 2) keypoints in PATCH coordinates -> (B, N, C) bilinear samples
        (grid_sample, align_corners=True, zero padding; dino_backbone.py:114-152)."""
        B, H, W, C = patch_features.shape
        needs_graph = torch.is_grad_enabled() and (patch_features.requires_grad or keypoints.requires_grad)
        if patch_features.is_cuda and not needs_graph and C == lib.C_FEAT and H == W:
            return lib.gather(patch_features.detach().contiguous().float(), keypoints.detach().contiguous().float())
        norm = keypoints.clone()
        norm[:, :, 0] = 2.0 * keypoints[:, :, 0] / (W - 1) - 1.0
        norm[:, :, 1] = 2.0 * keypoints[:, :, 1] / (H - 1) - 1.0
        sampled = torch.nn.functional.grid_sample(patch_features.permute(0, 3, 1, 2), norm.unsqueeze(1), mode="bilinear",
                                                  align_corners=True)
        return sampled.squeeze(2).permute(0, 2, 1)

    def patch_to_pixel(self, patch_coords: torch.Tensor) -> torch.Tensor:
        return patch_coords * self.patch_size + self.patch_size / 2      # dino_backbone.py:164

    def pixel_to_patch(self, pixel_coords: torch.Tensor) -> torch.Tensor:
        return (pixel_coords - self.patch_size / 2) / self.patch_size    # dino_backbone.py:177

"""Batched extraction + matching pipeline on one MI355X: the streaming counterpart of the reference's per-pair
harness (SequenceMatcher.extract + process_spacing, semantic-slam/visualize_matches_sequence.py:69-104, 272-357).

Differences from the reference harness, by design: every frame is extracted ONCE and its
descriptors reused for all pairs (as test/test_tracking.py:176-177 does); a whole chunk of frames goes through each
fused HIP stage in one launch; BatchNorm statistics stay per frame so results equal the reference's
B = 1 calls; nothing synchronises with the host until the caller reads the outputs.

PyTorch is plumbing only (device buffers, streams); all arithmetic runs in libsslam_hip.so via sslam_amd.lib.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import lib

PATCH = 16
N_PREFIX = 5   # CLS + 4 register tokens, dino_backbone.py:51,91
MAX_PAIRS_PER_LAUNCH = 65535


@dataclass
class ExtractorConfig:
    """Shapes and thresholds; defaults = semantic-slam/configs/train_config.yaml:5-17 and the CLI defaults of
    visualize_matches_sequence.py:381-388."""
    input_size: int = 448
    num_keypoints: int = 500
    nms_radius: int = 2
    min_score_percentile: float = 0.50
    bn_train_mode: bool = True          # visualize_* scripts never call backbone.eval()
    bn_eps: float = 1e-5
    saliency_weight: float = 0.3
    min_saliency: float = 0.5
    min_descriptor_sim: float = 0.7
    min_intensity: float = 0.15
    use_intensity: bool = True
    spacing: int = 1
    chunk_frames: int = 1024            # frames per launch group; 1024 x 28^2 x 384 fp32 = 1.2 GB of features
    precision: str = "fp32"             # "fp32": bit-exact vs the CPU reference (the product default and the parity claim);
                                        # "bf16": BASELINE configs[1] throughput mode - saliency CNN + descriptor MLP on bf16
                                        # MFMA (fp32 accumulate), everything else unchanged; NOT index-exact

    @property
    def grid(self) -> int:
        return self.input_size // PATCH

    def launch_group(self) -> int:
        """Frames per launch group: chunk_frames, but never more than one 32-bit buffer descriptor can span (the saliency CNN
        addresses the fp32 feature map through a single descriptor: < 4 GiB per launch; 2 965 frames at G = 40 are 7.3 GB)."""
        return max(1, min(self.chunk_frames, (2 ** 32 - 1) // (self.grid ** 2 * 384 * 4)))


def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


REFINER_ORDER_HEAD = ["input_proj.weight", "input_proj.bias"]
REFINER_ORDER_BLOCK = ["norm1.weight", "norm1.bias", "fc1.weight", "fc1.bias", "norm2.weight", "norm2.bias",
                       "fc2.weight", "fc2.bias"]
REFINER_ORDER_TAIL = ["output_proj.weight", "output_proj.bias"]


def refiner_weight_list(sd: dict):
    n_blocks = len({k.split(".")[1] for k in sd if k.startswith("residual_blocks.")})
    keys = list(REFINER_ORDER_HEAD)
    for i in range(n_blocks):
        keys += [f"residual_blocks.{i}.{k}" for k in REFINER_ORDER_BLOCK]
    keys += REFINER_ORDER_TAIL
    return [np.ascontiguousarray(_np(sd[k]), np.float32) for k in keys], n_blocks


class PackedSelector:
    """Device-resident, kernel-order copy of a KeypointSelector state_dict."""

    @staticmethod
    def supported(conv0_weight_shape) -> bool:
        """Shapes the saliency-CNN kernels are built for: Conv2d(384 -> 128 | 256, 3x3)."""
        sh = tuple(conv0_weight_shape)
        return len(sh) == 4 and sh[0] in (128, 256) and sh[1:] == (lib.C_FEAT, 3, 3)

    def __init__(self, sd: dict, device, bf16: bool = False):
        w1 = np.ascontiguousarray(_np(sd["conv.0.weight"]), np.float32)
        self.hidden = int(w1.shape[0])
        if not self.supported(w1.shape):
            raise lib.SslamHipError(f"selector shape {w1.shape} unsupported by the HIP kernels (hidden 128/256, C 384)")
        self.w1p = torch.from_numpy(lib.pack_conv3x3(w1)).to(device)
        self.b1 = torch.from_numpy(np.ascontiguousarray(_np(sd["conv.0.bias"]), np.float32)).to(device)
        self.w2 = torch.from_numpy(np.ascontiguousarray(_np(sd["conv.2.weight"]), np.float32).reshape(-1)).to(device)
        self.b2 = torch.from_numpy(np.ascontiguousarray(_np(sd["conv.2.bias"]), np.float32).reshape(-1)).to(device)
        self.w1p_bf16 = None
        if bf16:
            self.w1p_bf16 = torch.from_numpy(lib.pack_conv3x3_bf16(w1)).to(device).view(torch.bfloat16)

    @classmethod
    def empty(cls, hidden: int, device, bf16: bool = False) -> "PackedSelector":
        """Uninitialised device buffers of the packed shapes: the receiving side of the rank-0 weight broadcast (shard.py)."""
        self = cls.__new__(cls)
        self.hidden = int(hidden)
        if not cls.supported((hidden, lib.C_FEAT, 3, 3)):
            raise lib.SslamHipError(f"selector hidden {hidden} unsupported by the HIP kernels (128/256)")
        f32 = dict(dtype=torch.float32, device=device)
        self.w1p = torch.empty(9 * lib.C_FEAT * hidden, **f32)
        self.b1, self.w2, self.b2 = torch.empty(hidden, **f32), torch.empty(hidden, **f32), torch.empty(1, **f32)
        self.w1p_bf16 = torch.empty(9 * lib.C_FEAT * hidden, dtype=torch.bfloat16, device=device) if bf16 else None
        return self

    def tensors(self) -> list:
        """Every device buffer the kernels read, in a fixed order (what broadcast_weights sends)."""
        return [t for t in (self.w1p, self.b1, self.w2, self.b2, self.w1p_bf16) if t is not None]


class PackedRefiner:
    """Device-resident packed DescriptorRefiner weights (one buffer, sslam_refiner_layout order)."""

    @staticmethod
    def supported(input_dim: int, hidden_dim: int, output_dim: int, n_blocks: int) -> bool:
        """Shapes the fused descriptor-MLP kernel is built for: 384 -> 384 -> 128 with up to 8 residual blocks."""
        return (input_dim, hidden_dim, output_dim) == (lib.C_FEAT, lib.HID, lib.D_OUT) and 0 <= n_blocks <= 8

    def __init__(self, sd: dict, device, bf16: bool = False):
        ws, self.n_blocks = refiner_weight_list(sd)
        if ws[0].shape != (lib.HID, lib.C_FEAT) or ws[-2].shape != (lib.D_OUT, lib.HID):
            raise lib.SslamHipError("refiner shape unsupported by the HIP kernels (384 -> 384 -> 128)")
        self.packed = torch.from_numpy(lib.pack_refiner(ws, self.n_blocks)).to(device)
        self.packed_bf16 = torch.from_numpy(lib.pack_refiner_bf16(ws, self.n_blocks)).to(device) if bf16 else None

    @classmethod
    def empty(cls, n_blocks: int, device, bf16: bool = False) -> "PackedRefiner":
        """Uninitialised packed buffers for `n_blocks` residual blocks (receiving side of the weight broadcast)."""
        self = cls.__new__(cls)
        self.n_blocks = int(n_blocks)
        self.packed = torch.empty(int(lib.refiner_layout(n_blocks).total), dtype=torch.float32, device=device)
        self.packed_bf16 = (torch.empty(int(lib.lib().sslam_refiner_bf16_bytes(n_blocks)), dtype=torch.uint8, device=device)
                            if bf16 else None)
        return self

    def tensors(self) -> list:
        return [t for t in (self.packed, self.packed_bf16) if t is not None]


class ResampleTables:
    """Pillow coefficient tables on the device, cached per (height, width, size, filter)."""

    def __init__(self, device):
        self.device = device
        self._cache = {}

    def get(self, h: int, w: int, size: int, bicubic: bool):
        key = (h, w, size, bicubic)
        if key not in self._cache:
            tabs = []
            for n_in in (w, h):
                b, c, k = lib.resample_table(n_in, size, bicubic)
                tabs.append((torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device), k))
            self._cache[key] = tuple(tabs)
        return self._cache[key]


class SequencePipeline:
    def __init__(self, cfg: ExtractorConfig, selector_state: dict | None, refiner_state: dict | None, bn_state: dict | None = None,
                 device="cuda", vit=None, empty_shapes: tuple | None = None, vit_precision: str = "bf16"):
        """vit: optional sslam_amd.vit.DinoV3ViT, or any module whose weights convert to it (vit.KEY_MAPS) - enables
        run(images, tokens=None): images -> A0 -> HIP ViT (A1) -> ...; vit_precision "bf16" (throughput form, bf16 MFMA
        operands) or "fp32" (the reference's numerics for A1: fp32 operands on the fp32 matrix pipe, csrc/vit_f32.hip)
        selector_state / refiner_state None + empty_shapes=(selector hidden, refiner blocks): uninitialised packed buffers,
        to be filled by the rank-0 weight broadcast (shard.pipeline_from_rank0)."""
        self.cfg = cfg
        self.device = torch.device(device)
        lib.lib()   # fail loudly if the HIP library is not built
        if cfg.precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {cfg.precision!r}")
        if vit_precision not in ("bf16", "fp32"):          # checked whether or not a ViT is given: a typo must not pass silently
            raise ValueError(f"vit_precision must be 'bf16' or 'fp32', got {vit_precision!r}")
        self.bf16 = cfg.precision == "bf16"
        if selector_state is None or refiner_state is None:
            if empty_shapes is None:
                raise ValueError("state dicts or empty_shapes=(hidden, n_blocks) required")
            self.selector = PackedSelector.empty(empty_shapes[0], self.device, self.bf16)
            self.refiner = PackedRefiner.empty(empty_shapes[1], self.device, self.bf16)
        else:
            self.selector = PackedSelector(selector_state, self.device, self.bf16)
            self.refiner = PackedRefiner(refiner_state, self.device, self.bf16)
        self._ws = None             # caller-owned scratch handed to the *_ws entries (the library never allocates)
        self._stage = {}            # pipeline-owned stage temporaries (features, their bf16 copy, the A0 image): see _stage_buffer
        c = lib.C_FEAT
        bn = bn_state or {}
        f32 = dict(dtype=torch.float32, device=self.device)
        self.bn_gamma = torch.as_tensor(_np(bn.get("weight", np.ones(c))), **f32).contiguous()
        self.bn_beta = torch.as_tensor(_np(bn.get("bias", np.zeros(c))), **f32).contiguous()
        self.bn_mean = torch.as_tensor(_np(bn.get("running_mean", np.zeros(c))), **f32).contiguous()
        self.bn_var = torch.as_tensor(_np(bn.get("running_var", np.ones(c))), **f32).contiguous()
        self.tables = ResampleTables(self.device)
        self.vit_hip = None
        if vit is not None:
            from .vit import DinoV3ViT, convert_module
            from .vit_hip import HipViT
            if not isinstance(vit, DinoV3ViT):
                # a third-party module (timm's, transformers-keyed, ...): its weights, converted and verified (vit.convert_module);
                # this pipeline has no eager path, so a module that does not convert is an error here
                conv, why = convert_module(vit)
                if conv is None:
                    raise lib.SslamHipError(f"{type(vit).__name__} cannot run on the HIP ViT: {why}")
                vit = conv
            # the packers copy every tensor to the device themselves: the caller's module is NOT moved (nn.Module.to works in place)
            if vit_precision == "fp32":
                from .vit_hip import HipViTF32
                self.vit_hip = HipViTF32(vit, self.device)
            else:
                self.vit_hip = HipViT(vit, self.device)
        self.vit_precision = vit_precision

    def weight_tensors(self) -> list:
        """Every device buffer of packed weights / BatchNorm state, in a fixed order (6.7 MB fp32 at the shipped shapes)."""
        return self.selector.tensors() + self.refiner.tensors() + [self.bn_gamma, self.bn_beta, self.bn_mean, self.bn_var]

    def workspace(self, n_frames: int, n_pairs: int) -> torch.Tensor | None:
        """The pipeline's scratch tensor, grown to sslam_workspace_bytes(n_frames, G, K, n_pairs): one buffer for all
        stages of a step (they run in stream order on the caller's stream)."""
        need = lib.workspace_bytes(max(1, n_frames), self.cfg.grid, self.cfg.num_keypoints, max(0, n_pairs))
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _stage_buffer(self, name: str, shape: tuple, dtype) -> torch.Tensor:
        """A pipeline-owned temporary of a stage, allocated once and reused by every launch group of every pass (grown by
        replacement, never shrunk): the multi-GB intermediates - a launch group's fp32 features are 2.5 GB at G = 40 - do not
        go back to the caching allocator between groups, whose state otherwise decided whether a pass ran at 52 k or 96 k
        frames/s (bf16 mode, 2 965 frames).  Stages run in stream order on the caller's stream, so a buffer is free again by the
        time the next group's producer writes it; results a CALLER keeps are never placed here (reuse=False paths allocate)."""
        need = 1
        for d in shape:
            need *= int(d)
        t = self._stage.get(name)
        if t is None or t.numel() < need or t.dtype != dtype:
            t = torch.empty(need, dtype=dtype, device=self.device)
            self._stage[name] = t
        return t[:need].view(shape)

    # ---------------------------------------------------------------------------------------------- stages
    def preprocess(self, images_u8: torch.Tensor, reuse: bool = False) -> torch.Tensor:
        """A0: (N, H, W, 3) uint8 -> (N, 3, S, S) fp32, Pillow-exact (the ViT input).
        reuse: write into the pipeline's own A0 buffer (valid until the next reuse=True call) instead of a fresh tensor."""
        n, h, w, _ = images_u8.shape
        size = self.cfg.input_size
        th, tv = self.tables.get(h, w, size, False)
        out = self._stage_buffer("a0_image", (n, 3, size, size), torch.float32) if reuse else None
        return lib.preprocess_u8(images_u8, size, th, tv, out=out)

    def tokens_from_images(self, images_u8: torch.Tensor, vit_chunk: int | None = None, out: torch.Tensor | None = None,
                           batch_frames: int | None = None) -> torch.Tensor:
        """A0 + A1: (N, H, W, 3) uint8 -> (N, 5 + G*G, 384) fp32 tokens via the HIP ViT, `vit_chunk` frames at a time.
        Default chunk: HipViT.chunk_frames - whole rounds of the ViT's row-tile workgroups (82 frames at 448 x 448), alternating between two streams.
        batch_frames: the length of the sequence these frames are a piece of, when the caller feeds it in pieces (the streaming
        harness): the fp32 ViT picks its attention form by the batch, so that a frame's tokens do not depend on the cuts."""
        if self.vit_hip is None:
            raise lib.SslamHipError("this pipeline was built without a ViT: pass tokens, or construct it with vit=")
        if vit_chunk is None:
            vit_chunk = self.vit_hip.chunk_frames(self.cfg.input_size)
        shape = (images_u8.shape[0], N_PREFIX + self.cfg.grid ** 2, lib.C_FEAT)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous fp32 tensor of shape {shape}")
        # A0 runs over 16 launch groups at a time (bounds the ViT input: 1.6 GB of bf16 patch rows at 448 x 448); the ViT then
        # alternates those groups between its two streams.  A0 writes the patch-embedding operand directly (bf16 rows of 768 per
        # patch): no fp32 image, no im2patch pass; resampling ratios its tiled kernel does not cover take the fp32 image.
        span = 16 * vit_chunk
        size = self.cfg.input_size
        n, h, w, _ = images_u8.shape
        th, tv = self.tables.get(h, w, size, False)
        for a in range(0, n, span):
            b = min(a + span, n)
            # the fp32 ViT consumes the fp32 image (its patch embedding is an fp32 contraction too); bf16: the patch rows
            patches = lib.preprocess_u8_patches(images_u8[a:b], size, th, tv) if self.vit_precision == "bf16" else None
            if patches is not None:
                self.vit_hip.forward_features(None, out=out[a:b], chunk=vit_chunk, patches=patches, size=size)
            else:
                self.vit_hip.forward_features(self.preprocess(images_u8[a:b], reuse=self.vit_hip.n_streams < 2), out=out[a:b], chunk=vit_chunk,
                                              batch_frames=max(n, batch_frames or 0))
        return out

    def preprocess_patches(self, images_u8: torch.Tensor):
        """A0 as the ViT consumes it: (N, H, W, 3) uint8 -> (N, G*G, 768) bf16 patch rows (None: ratio not covered, see lib)."""
        n, h, w, _ = images_u8.shape
        th, tv = self.tables.get(h, w, self.cfg.input_size, False)
        return lib.preprocess_u8_patches(images_u8, self.cfg.input_size, th, tv)

    def features(self, tokens: torch.Tensor, bf16_copy: bool = False, reuse: bool = False):
        """A2: (N, 5 + G*G, 384) ViT tokens -> (N, G, G, 384) per-frame-normalised patch features
        (bf16_copy: also their bf16 copy, written in the same pass, for the bf16-mode saliency CNN).
        reuse: write into the pipeline's own feature buffers (valid until the next reuse=True call) - what extract() does."""
        g = self.cfg.grid
        if tokens.shape[1] != N_PREFIX + g * g:
            raise AssertionError(f"Expected {g * g} patches, got {tokens.shape[1] - N_PREFIX}")   # dino_backbone.py:94
        flat = (tokens.shape[0], g * g, lib.C_FEAT)
        o32 = self._stage_buffer("features", flat, torch.float32) if reuse else None
        o16 = self._stage_buffer("features_bf16", flat, torch.bfloat16) if (reuse and bf16_copy) else None
        r = lib.bn_tokens(tokens, N_PREFIX, 1, self.bn_gamma, self.bn_beta, self.bn_mean, self.bn_var,
                          self.cfg.bn_train_mode, self.cfg.bn_eps, out=o32, want_stats=False, bf16_copy=bf16_copy, out_bf16=o16)
        shape = (tokens.shape[0], g, g, lib.C_FEAT)
        return (r[0].view(shape), r[3].view(shape)) if bf16_copy else r[0].view(shape)

    def launch_group(self) -> int:
        """Frames per launch group (ExtractorConfig.launch_group: chunk_frames, capped by the 4 GiB a buffer descriptor spans)."""
        return self.cfg.launch_group()

    def alloc_extract(self, n: int, with_intensity: bool) -> dict:
        """Output buffers of extract() for n frames (every launch group writes its slice: nothing is concatenated)."""
        cfg, dev = self.cfg, self.device
        g, K = cfg.grid, cfg.num_keypoints
        f32 = dict(dtype=torch.float32, device=dev)
        out = dict(saliency=torch.empty((n, g, g), **f32), keypoints_patch=torch.empty((n, K, 2), **f32),
                   keypoints_pixel=torch.empty((n, K, 2), **f32), scores=torch.empty((n, K), **f32),
                   idx=torch.empty((n, K), dtype=torch.int32, device=dev), descriptors=torch.empty((n, K, lib.D_OUT), **f32),
                   status=torch.empty((n,), dtype=torch.int32, device=dev))
        if with_intensity:
            out["intensity"] = torch.empty((n, K), **f32)
        return out

    def extract(self, tokens: torch.Tensor, images_u8: torch.Tensor | None = None, out: dict | None = None,
                images_ready: "torch.cuda.Event | None" = None) -> dict:
        """A2..A9 for any number of frames (launch groups of `launch_group()` frames).  Returns device tensors; no host
        synchronisation unless num_keypoints exceeds the number of grid cells (the only case `status` can be set).
        out: buffers from alloc_extract (or row slices of them) to write into - the sharded runner and the streaming
        scheduler pass slices of sequence-sized buffers, so nothing is copied afterwards.
        images_ready: event after which images_u8 holds the frames (an upload in flight on another stream): the current
        stream waits for it in front of the FIRST kernel that reads pixels - A9, the last stage - so A2..A7, which read
        only the tokens, run while the upload is still in flight."""
        n, step = tokens.shape[0], self.launch_group()
        if out is None:
            out = self.alloc_extract(n, images_u8 is not None)
        for a in range(0, n, step):
            b = min(a + step, n)
            self._extract_group(tokens[a:b], None if images_u8 is None else images_u8[a:b], {k: v[a:b] for k, v in out.items()},
                                images_ready if a == 0 else None)
        if self.cfg.num_keypoints > self.cfg.grid ** 2 and bool(out["status"].any()):
            # torch.topk raises there in the reference (keypoint_selector.py:160 / :176)
            raise RuntimeError("selected index k out of range")
        return out

    def _extract_group(self, tokens: torch.Tensor, images_u8: torch.Tensor | None, out: dict, images_ready=None) -> None:
        cfg, s = self.cfg, self.selector
        ws = self.workspace(tokens.shape[0], 0)
        if self.bf16:
            feat, feat_bf = self.features(tokens, bf16_copy=True, reuse=True)
            lib.selector_saliency_bf16(feat_bf, s.w1p_bf16, s.b1, s.w2, s.b2, s.hidden, out=out["saliency"])
            del feat_bf
        else:
            feat = self.features(tokens, reuse=True)
            lib.selector_saliency(feat, s.w1p, s.b1, s.w2, s.b2, s.hidden, out=out["saliency"], workspace=ws)
        lib.select_keypoints(out["saliency"], cfg.num_keypoints, cfg.nms_radius, cfg.min_score_percentile,
                             out=(out["keypoints_patch"], out["scores"], out["idx"], out["keypoints_pixel"], out["status"]))
        if self.bf16:
            lib.gather_refine_bf16(feat, out["keypoints_patch"], self.refiner.packed_bf16, self.refiner.n_blocks, out=out["descriptors"])
        else:
            lib.gather_refine(feat, out["keypoints_patch"], self.refiner.packed, self.refiner.n_blocks, out=out["descriptors"])
        if images_ready is not None:
            torch.cuda.current_stream(self.device).wait_event(images_ready)
        if images_u8 is not None and "intensity" in out:
            n, h, w, _ = images_u8.shape
            th, tv = self.tables.get(h, w, cfg.input_size, True)
            lib.keypoint_intensity(images_u8, cfg.input_size, th, tv, out["keypoints_pixel"], out=out["intensity"])

    def alloc_match(self, n_pairs: int, k: int | None = None) -> dict:
        """Output buffers of match() for n_pairs pairs (fixed capacity K per pair + device-side count)."""
        k = self.cfg.num_keypoints if k is None else k
        dev = self.device
        return dict(matches=torch.empty((n_pairs, k, 2), dtype=torch.int64, device=dev),
                    quality=torch.empty((n_pairs, k), dtype=torch.float32, device=dev),
                    match_count=torch.empty((n_pairs,), dtype=torch.int32, device=dev))

    def match(self, desc, scores, intensity=None, spacing: int | None = None, out: dict | None = None) -> dict:
        """M1 for all pairs (i, i + spacing) inside the batch.  desc (N, K, 128), scores (N, K), intensity (N, K).
        out: row slices of alloc_match buffers to write into (the streaming scheduler passes slices of sequence-sized ones)."""
        cfg = self.cfg
        sp = cfg.spacing if spacing is None else spacing
        n, k = desc.shape[0], desc.shape[1]
        n_pairs = n - sp
        if n_pairs <= 0:
            z = torch.zeros
            return dict(matches=z((0, k, 2), dtype=torch.int64, device=desc.device),
                        quality=z((0, k), dtype=torch.float32, device=desc.device),
                        match_count=z((0,), dtype=torch.int32, device=desc.device))
        use_int = cfg.use_intensity and intensity is not None
        res = dict(out) if out is not None else self.alloc_match(n_pairs, k)
        aux = []
        for a in range(0, n_pairs, MAX_PAIRS_PER_LAUNCH):       # the pair index is a 16-bit grid dimension
            m = min(MAX_PAIRS_PER_LAUNCH, n_pairs - a)
            d1, d2 = desc[a:a + m], desc[a + sp:a + sp + m]
            nn12, s12, nn21, _, _ = lib.sim_argmax(d1, k * lib.D_OUT, k, d2, k * lib.D_OUT, k, m, workspace=self.workspace(0, m))
            lib.match_finalize(nn12, s12, nn21, k, k, m, scores[a:], k, scores[a + sp:], k,
                               intensity[a:] if use_int else None, intensity[a + sp:] if use_int else None,
                               1.0 - cfg.saliency_weight, cfg.saliency_weight, cfg.min_saliency,
                               cfg.min_descriptor_sim, cfg.min_intensity,
                               out=(res["matches"][a:a + m], res["quality"][a:a + m], res["match_count"][a:a + m]))
            aux.append((nn12, nn21, s12))
        for i, key in enumerate(("nn12", "nn21", "sim")):       # the arg-max arrays (diagnostics): one launch in practice
            res[key] = aux[0][i] if len(aux) == 1 else torch.cat([x[i] for x in aux])
        return res

    def run(self, images_u8: torch.Tensor | None, tokens: torch.Tensor | None = None, with_preprocess: bool = False) -> dict:
        """One pass of the hot path over a frame sequence: extract every frame once, match (i, i+spacing).
        tokens=None: compute them from the images with the HIP ViT (A0 + A1)."""
        cfg = self.cfg
        if tokens is None:
            tokens = self.tokens_from_images(images_u8)
            with_preprocess = False
        vit_in = None
        if with_preprocess and images_u8 is not None:
            step = self.launch_group()
            for a in range(0, images_u8.shape[0], step):
                vit_in = self.preprocess(images_u8[a:a + step], reuse=True)      # A0: would feed the ViT (A1)
        out = dict(self.extract(tokens, images_u8))
        out.update(self.match(out["descriptors"], out["scores"], out.get("intensity")))
        if vit_in is not None:
            out["vit_input_last_chunk"] = vit_in
        return out

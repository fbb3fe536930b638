"""HIP execution of the DINOv3 ViT-S/16 forward (sslam_vit_forward, csrc/vit.hip) for a `sslam_amd.vit.DinoV3ViT`
module: packs its Parameters into device buffers (matrices bf16, vectors fp32), builds the RoPE tables, owns the
workspace.  PyTorch is plumbing; no torch op takes part in the forward."""
from __future__ import annotations


import torch

from . import lib
from .vit import DinoV3ViT


class HipViT:
    def __init__(self, vit: DinoV3ViT, device="cuda"):
        if (vit.embed_dim, vit.heads, vit.patch, vit.n_register, len(vit.blocks)) != (384, 6, 16, 4, 12):
            raise lib.SslamHipError("the HIP ViT is built for ViT-S/16 with 4 register tokens (384 / 6 heads / 12 layers)")
        if vit.blocks[0].up_proj.out_features != 1536:
            raise lib.SslamHipError("MLP width must be 1536")
        self.vit, self.device = vit, torch.device(device)
        self._keep = []
        self._rope = {}
        self.n_streams = 2          # launch groups alternate between two streams (1: single stream; tools/vit_streams.py A/B)
        self._side = None            # two side streams + their workspaces (forward_features with several launch groups)
        self._side_ws = [None, None]
        self.w = lib.VitWeights()
        bf, f32 = self._frag, self._f32
        self.w.patch_w = bf(vit.patch_embed.weight.reshape(384, 768))
        self.w.patch_b = f32(vit.patch_embed.bias)
        self.w.prefix = f32(torch.cat([vit.cls_token[0], vit.register_tokens[0]], dim=0))
        qs = 0.125 * 1.4426950408889634          # 1/sqrt(64) and the exp2 domain of the softmax, folded into the q rows
        for i, b in enumerate(vit.blocks):
            ly = self.w.layer[i]
            ly.ln1_g, ly.ln1_b = f32(b.norm1.weight), f32(b.norm1.bias)
            ly.wqkv = bf(torch.cat([b.q_proj.weight * qs, b.k_proj.weight, b.v_proj.weight], dim=0))
            ly.bqkv = f32(torch.cat([b.q_proj.bias * qs, torch.zeros_like(b.q_proj.bias), b.v_proj.bias]))
            # LayerScale folded into the projections that feed the residual stream: ls * (W a + b) = (ls W) a + ls b
            ly.wo, ly.bo = bf(b.o_proj.weight * b.ls1[:, None]), f32(b.o_proj.bias * b.ls1)
            ly.ln2_g, ly.ln2_b = f32(b.norm2.weight), f32(b.norm2.bias)
            ly.wup, ly.bup = bf(b.up_proj.weight), f32(b.up_proj.bias)
            ly.wdown, ly.bdown = bf(b.down_proj.weight * b.ls2[:, None]), f32(b.down_proj.bias * b.ls2)
            ly.wmlp = self._hold(torch.from_numpy(lib.pack_vit_mlp(b.up_proj.weight.detach().float().cpu().numpy(),
                                                                  b.down_proj.weight.detach().float().cpu().numpy(),
                                                                  b.ls2.detach().float().cpu().numpy())).to(self.device))
        self.w.norm_g, self.w.norm_b = f32(vit.norm.weight), f32(vit.norm.bias)

    def _hold(self, t):
        self._keep.append(t)
        return t.data_ptr()

    def _frag(self, w):
        """(N, K) nn.Linear weight -> bf16 in the streaming order of csrc/vit.hip gemm_rt_kernel (packed by the library)."""
        packed = lib.pack_vit_linear(w.detach().float().cpu().numpy())
        return self._hold(torch.from_numpy(packed).to(self.device))

    def _f32(self, t):
        return self._hold(t.detach().to(self.device, torch.float32).contiguous())

    @staticmethod
    def chunk_frames(size: int) -> int:
        """Frames per launch group: just under 2 x 253 row tiles of 128 tokens.  The GEMM launches run one workgroup per (row
        tile, column half) at two per CU and the fused MLP one per row tile, so a multiple of 253 tiles is whole rounds for all
        of them (448 x 448: 41 frames = 1 round 14.0 k frames/s, 82 = 2 rounds 15.0 k, 164 = 4 rounds 15.2 k on one stream;
        64 frames = a partial round: 13.3 k).  Several groups alternate between two streams (forward_features)."""
        return max(1, (506 * 128) // (5 + (size // 16) ** 2))

    def forward_features(self, images: torch.Tensor | None, out: torch.Tensor | None = None, chunk: int | None = None,
                         patches: torch.Tensor | None = None, size: int | None = None) -> torch.Tensor:
        """(B, 3, S, S) fp32 cuda -> (B, 5 + (S/16)^2, 384) fp32 tokens (final-LayerNormed), `chunk` frames per launch group.
        patches (with size = S): the bf16 patch rows of lib.preprocess_u8_patches (B, (S/16)^2, 768) instead of the image -
        no fp32 image, no im2patch pass, the same tokens bit for bit."""
        if patches is not None:
            n, s = patches.shape[0], int(size)
            assert patches.is_cuda and patches.dtype == torch.bfloat16 and patches.shape[1:] == ((s // 16) ** 2, 768)
            dev_of = patches
        else:
            n, _, s, s2 = images.shape
            assert s == s2 and s % 16 == 0 and images.is_cuda
            dev_of = images
        g = s // 16
        if g not in self._rope:
            cos, sin = self.vit.rope_tables(g, g, self.device)
            self._rope[g] = (cos.float().contiguous(), sin.float().contiguous())
        self.w.rope_cos, self.w.rope_sin = self._rope[g][0].data_ptr(), self._rope[g][1].data_ptr()
        step = chunk or self.chunk_frames(s)
        need = lib.vit_workspace_bytes(min(n, step), s)
        x = patches.contiguous() if patches is not None else images.detach().float().contiguous()
        if out is None:
            out = torch.empty((n, 5 + g * g, lib.C_FEAT), dtype=torch.float32, device=dev_of.device)

        def launch(a, ws):
            if patches is not None:
                lib.vit_forward_patches(x[a:a + step], s, self.w, ws, out=out[a:a + step])
            else:
                lib.vit_forward(x[a:a + step], self.w, ws, out=out[a:a + step])

        starts = list(range(0, n, step))
        if len(starts) >= 2 and self.n_streams >= 2:
            # Launch groups alternate between two side streams: every launch has a ramp and an uneven tail (mean workgroup lifetime x
            # workgroups / slots explains only 60 of the QKV launch's 86 us), and the other group's kernels fill them
            # (tools/vit_streams.py: +3-5 % at 82 frames per group; a third stream adds nothing).  The caller's stream semantics are
            # kept: the side streams start after the current stream's work and the current stream waits for them.
            dev = dev_of.device
            cur = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            for i in (0, 1):          # one workspace per stream; [0] doubles as the single-stream workspace (never both at once)
                if self._side_ws[i] is None or self._side_ws[i].numel() < need:
                    self._side_ws[i] = torch.empty(need, dtype=torch.uint8, device=dev)
            ready = cur.record_event()
            for i, a in enumerate(starts):
                st = self._side[i & 1]
                if i < 2:
                    st.wait_event(ready)
                with torch.cuda.stream(st):
                    launch(a, self._side_ws[i & 1])
            for st in self._side:
                x.record_stream(st)
                out.record_stream(st)
                cur.wait_stream(st)
            return out
        if self._side_ws[0] is None or self._side_ws[0].numel() < need:
            self._side_ws[0] = torch.empty(need, dtype=torch.uint8, device=self.device)
        for a in starts:
            launch(a, self._side_ws[0])
        return out


class HipViTF32:
    """The same forward with the REFERENCE'S numerics (fp32 operands on the fp32 matrix pipe, sslam_vit_forward_f32,
    csrc/vit_f32.hip): what DinoBackbone(vit_precision="fp32") runs on a GPU.  The weights stay fp32 with nothing folded into them;
    q / k / v are concatenated and the four per-layer matrices re-ordered into the fragment order their GEMM streams."""

    def __init__(self, vit: DinoV3ViT, device="cuda"):
        if (vit.embed_dim, vit.heads, vit.patch, vit.n_register, len(vit.blocks)) != (384, 6, 16, 4, 12):
            raise lib.SslamHipError("the HIP ViT is built for ViT-S/16 with 4 register tokens (384 / 6 heads / 12 layers)")
        if vit.blocks[0].up_proj.out_features != 1536:
            raise lib.SslamHipError("MLP width must be 1536")
        self.vit, self.device = vit, torch.device(device)
        self._keep, self._rope = [], {}
        # launch groups on ONE stream: alternating them between two (as the bf16 HipViT does, where it gains 3-5 %) costs 3 % here -
        # two groups in flight double the working set (1.6 GB against a 256 MB MALL) and the fp32 kernels have no idle pipe to fill
        # (613 frames: 2 598 frames/s on one stream, 2 513 on two)
        self._ws, self._side, self.n_streams = [None, None], None, 1
        self.w = lib.VitWeightsF32()
        f, pk = self._f32, self._packed
        self.w.patch_w, self.w.patch_b = f(vit.patch_embed.weight.reshape(384, 768)), f(vit.patch_embed.bias)
        self.w.prefix = f(torch.cat([vit.cls_token[0], vit.register_tokens[0]], dim=0))
        for i, b in enumerate(vit.blocks):
            ly = self.w.layer[i]
            ly.ln1_g, ly.ln1_b = f(b.norm1.weight), f(b.norm1.bias)
            ly.wqkv = pk(torch.cat([b.q_proj.weight, b.k_proj.weight, b.v_proj.weight], dim=0))
            ly.bqkv = f(torch.cat([b.q_proj.bias, torch.zeros_like(b.q_proj.bias), b.v_proj.bias]))
            ly.wo, ly.bo, ly.ls1 = pk(b.o_proj.weight), f(b.o_proj.bias), f(b.ls1)
            ly.ln2_g, ly.ln2_b = f(b.norm2.weight), f(b.norm2.bias)
            ly.wup, ly.bup = pk(b.up_proj.weight), f(b.up_proj.bias)
            ly.wdown, ly.bdown, ly.ls2 = pk(b.down_proj.weight), f(b.down_proj.bias), f(b.ls2)
        self.w.norm_g, self.w.norm_b = f(vit.norm.weight), f(vit.norm.bias)

    def _f32(self, t):
        t = t.detach().to(self.device, torch.float32).contiguous()
        self._keep.append(t)
        return t.data_ptr()

    def _packed(self, w):
        """(N, K) nn.Linear weight -> fp32, the same values in the fragment order of the per-layer GEMM (packed by the library)."""
        t = torch.from_numpy(lib.pack_vit_f32_linear(w.detach().float().cpu().numpy())).to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    @staticmethod
    def chunk_frames(size: int) -> int:
        """Frames per launch group: 512 row tiles of 128 tokens.  The per-layer GEMMs run one workgroup per (row tile, 128 columns)
        at three per CU (768 slots), so 512 row tiles are whole rounds for all four of them (N = 384: 2 rounds, 1 152: 6, 1 536: 8);
        64 frames = 395 row tiles left the N = 384 GEMMs at 1.54 rounds (a quarter of their time idle).  83 frames at 448 x 448;
        also bounds the workspace (13.7 KB per token: 0.9 GB)."""
        return max(1, (512 * 128) // (5 + (size // 16) ** 2))

    def forward_features(self, images: torch.Tensor, out: torch.Tensor | None = None, chunk: int | None = None,
                         batch_frames: int | None = None) -> torch.Tensor:
        """(B, 3, S, S) fp32 cuda -> (B, 5 + (S/16)^2, 384) fp32 tokens (final-LayerNormed), launch groups of `chunk` frames one after
        the other on the caller's stream (n_streams = 2: alternating between two side streams as HipViT does - measured slower
        here, see __init__); the caller's stream semantics are kept either way.
        The attention's launch form follows the size of the BATCH, not of a launch group: up to 8 frames (the reference's own
        callers: B = 1, B = 4) the few-frame form (include/sslam_hip.h: key-split attention, K-quartered down projection; 1.83 -> 1.11 ms
        for one frame), above it the one-pass
        form for every group, a short last one included - so a frame's tokens do not depend on where a batch is cut.
        batch_frames: the size of the batch these frames belong to when the caller itself hands it over in pieces
        (SequencePipeline.tokens_from_images); default: this call's frame count."""
        n, _, s, s2 = images.shape
        assert s == s2 and s % 16 == 0 and images.is_cuda
        g = s // 16
        if g not in self._rope:
            cos, sin = self.vit.rope_tables(g, g, self.device)
            cos, sin = cos.float().contiguous(), sin.float().contiguous()
            if not (torch.equal(cos[:, :32], cos[:, 32:]) and torch.equal(sin[:, :32], sin[:, 32:])):
                raise ValueError("sslam_vit_forward_f32 reads columns 0..31 of the RoPE tables (DINOv3 tiles its 32 angles twice); "
                                 "these tables differ between d and d + 32")
            self._rope[g] = (cos, sin)
        self.w.rope_cos, self.w.rope_sin = self._rope[g][0].data_ptr(), self._rope[g][1].data_ptr()
        step = chunk or self.chunk_frames(s)
        need = lib.vit_f32_workspace_bytes(min(n, step), s)
        x = images.detach().float().contiguous()
        if out is None:
            out = torch.empty((n, 5 + g * g, lib.C_FEAT), dtype=torch.float32, device=images.device)
        starts = list(range(0, n, step))
        dev = images.device
        # None: the library's own rule for a launch of <= 8 frames (key split; the test-only knob can switch it off)
        form = None if (batch_frames or n) <= lib.ATTN_KEY_SPLIT_MAX_FRAMES else lib.ATTN_ONE_PASS
        for i in (0, 1) if len(starts) >= 2 and self.n_streams >= 2 else (0,):
            if self._ws[i] is None or self._ws[i].numel() < need:
                self._ws[i] = torch.empty(need, dtype=torch.uint8, device=dev)
        if len(starts) >= 2 and self.n_streams >= 2:
            cur = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            ready = cur.record_event()
            for i, a in enumerate(starts):
                st = self._side[i & 1]
                if i < 2:
                    st.wait_event(ready)
                with torch.cuda.stream(st):
                    lib.vit_forward_f32(x[a:a + step], self.w, self._ws[i & 1], out=out[a:a + step], attention_form=form)
            for st in self._side:
                x.record_stream(st)
                out.record_stream(st)
                cur.wait_stream(st)
            return out
        for a in starts:
            lib.vit_forward_f32(x[a:a + step], self.w, self._ws[0], out=out[a:a + step], attention_form=form)
        return out

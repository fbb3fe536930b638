"""CPU tests of the drop-in boundary: the reference's class / method / state_dict surface, the
autograd-capable torch-op path of the modules (used by train.py) against the reference's golden vectors,
and the C-ABI library's export table.  No GPU compute here."""
import os
import re

import numpy as np
import pytest
import torch

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class TokenDino(torch.nn.Module):
    """stand-in ViT that returns the tokens it was handed (same device as timm would be used in the reference)"""

    def __init__(self):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1))
        self.embed_dim = 384
        self.tokens = None

    def forward_features(self, images):
        return self.tokens


def test_library_exports_every_declared_symbol():
    from sslam_amd import lib
    hdr = open(os.path.join(ROOT, "include", "sslam_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(sslam_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    L = lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/sslam_hip.h but not exported"
    assert sorted(lib.EXPORTS) == declared
    assert L.sslam_arch() == b"gfx950" and L.sslam_version() >= 100


def test_host_side_packing_and_tables():
    from sslam_amd import lib
    from oracle import ora
    w = synth.selector_state(0)["conv.0.weight"]
    p = lib.pack_conv3x3(w).reshape(12, 9, 4, 256, 8)
    # stage (chunk, tap) holds [k-group][n][8 k] with k permuted (0,2,4,6,1,3,5,7): the MFMA B-fragment order
    perm = np.array([0, 2, 4, 6, 1, 3, 5, 7])
    for tap, chunk, n in [(0, 0, 0), (4, 7, 100), (8, 11, 255)]:
        want = w[n, chunk * 32:(chunk + 1) * 32, tap // 3, tap % 3].reshape(4, 8)[:, perm]
        assert np.array_equal(p[chunk, tap, :, n], want)
    ws = ora.refiner_weight_list(synth.refiner_state(0), 2)
    packed = lib.pack_refiner(ws, 2)
    lay = lib.refiner_layout(2)
    assert packed.size == lay.total == 791552                      # 791 552 parameters
    assert np.array_equal(packed[lay.in_b:lay.in_b + 384], ws[1])
    assert np.array_equal(packed[lay.out_b:lay.out_b + 128], ws[-1])
    # Pillow coefficient tables: identical to what the oracle derives (checked end-to-end against PIL goldens there)
    b, c, k = lib.resample_table(640, 448, False)
    assert k == 5 and b.shape == (896,) and c.shape == (448 * 5,)
    assert c.reshape(448, 5).sum(axis=1).min() >= (1 << 22) - 3 and c.reshape(448, 5).sum(axis=1).max() <= (1 << 22) + 3
    b, c, k = lib.resample_table(480, 480, True)                   # same size: identity taps
    assert all(c.reshape(480, k)[i, b[2 * i + 1] - 1 if b[2 * i] + b[2 * i + 1] == 480 and i > 470 else (i - b[2 * i])] == 1 << 22
               for i in (0, 5, 200, 479))
    with pytest.raises(ValueError):
        lib.resample_table(0, 10, False)


def test_state_dict_surface_matches_reference():
    from models.descriptor_refiner import DescriptorRefiner
    from models.keypoint_selector import KeypointSelector
    sel = KeypointSelector(384, 256)
    assert list(sel.state_dict().keys()) == ["conv.0.weight", "conv.0.bias", "conv.2.weight", "conv.2.bias"]
    assert sel.state_dict()["conv.0.weight"].shape == (256, 384, 3, 3)
    assert KeypointSelector().conv[0].out_channels == 128          # class default, keypoint_selector.py:25
    ref = DescriptorRefiner(384, 384, 128)
    keys = list(ref.state_dict().keys())
    want = ["input_proj.weight", "input_proj.bias"]
    for i in range(2):
        for m in ("norm1", "fc1", "norm2", "fc2"):
            want += [f"residual_blocks.{i}.{m}.weight", f"residual_blocks.{i}.{m}.bias"]
    want += ["output_proj.weight", "output_proj.bias"]
    assert keys == want
    assert sum(p.numel() for p in ref.parameters()) == 791552
    assert sum(p.numel() for p in sel.parameters()) == 885249
    # loads a checkpoint written with the reference's key names (train.py:582-592)
    sel.load_state_dict({k: t(v) for k, v in synth.selector_state(0).items()})
    ref.load_state_dict({k: t(v) for k, v in synth.refiner_state(0).items()})


def _backbone(grid):
    from models.dino_backbone import DinoBackbone
    bb = DinoBackbone(input_size=grid * 16, freeze=True, dino=TokenDino())
    assert (bb.embed_dim, bb.patch_size, bb.grid_h, bb.grid_w, bb.num_patches, bb.n_storage_tokens) == (384, 16, grid, grid, grid * grid, 4)
    return bb


def test_backbone_batchnorm_modes_cpu_path():
    g = gold("bn_tokens")
    tok = synth.tokens(0, 28, batch=2)
    bb = _backbone(28)
    assert bb._is_frozen()
    bb.dino.tokens = t(tok[:1])
    with torch.no_grad():
        y = bb(torch.zeros(1, 3, 448, 448))                        # train mode by default, like the visualize_* scripts
    assert y.shape == (1, 28, 28, 384)
    assert np.abs(y.numpy().reshape(784, 384)[::13] - g["train_b1_f0_sub"][0]).max() < 1e-5
    assert np.abs(bb.feature_norm.running_var.numpy() - g["train_b1_f0_running_var"]).max() < 1e-5
    bb2 = _backbone(28).eval()
    bb2.dino.tokens = t(tok)
    with torch.no_grad():
        y2 = bb2(torch.zeros(2, 3, 448, 448))
    assert np.array_equal(y2.numpy().reshape(2, 784, 384)[:, ::13], g["eval_b2_sub"])
    bb.dino.tokens = t(tok[:, :700])
    with pytest.raises(AssertionError):
        bb(torch.zeros(2, 3, 448, 448))                            # dino_backbone.py:94


def test_selector_and_refiner_cpu_path_against_golden():
    from models.descriptor_refiner import DescriptorRefiner
    from models.keypoint_selector import KeypointSelector
    from oracle import ora
    gs, gr, gc = gold("selector"), gold("gather_refine"), gold("select_cases")
    feat = t(ora.bn_tokens(synth.tokens(1, 28))[0].reshape(1, 28, 28, 384))
    sel = KeypointSelector(384, 256).eval()
    sel.load_state_dict({k: t(v) for k, v in synth.selector_state(0).items()})
    with torch.no_grad():
        sal = sel(feat)
        assert sal.shape == (1, 28, 28, 1)
        assert np.abs(sal[0, :, :, 0].numpy() - gs["g28_saliency"]).max() < 2e-6
        kp, sc = sel.select_keypoints(t(gs["g28_saliency"]).reshape(1, 28, 28, 1), 500)
        assert kp.shape == (1, 500, 2) and kp.dtype == torch.float32 and sc.shape == (1, 500)
        assert np.array_equal(kp[0].numpy(), gs["g28_kp"]) and np.array_equal(sc[0].numpy(), gs["g28_scores"])
        assert np.array_equal(sel._apply_nms(t(gs["g28_saliency"])[None], 2)[0].numpy(), gs["g28_nms"])
        for tag in bytes(gc["tags"]).decode().split(","):
            m = gc[tag + "_map"]
            kp, sc = sel.select_keypoints(t(m).reshape(1, *m.shape, 1), int(gc[tag + "_K"]), int(gc[tag + "_radius"]),
                                          float(gc[tag + "_pct"]))
            assert np.array_equal(kp[0].numpy(), gc[tag + "_kp"]), tag
            assert np.array_equal(sc[0].numpy(), gc[tag + "_scores"]), tag
        with pytest.raises(RuntimeError):
            sel.select_keypoints(t(gc["Belse_r2_map"]).reshape(1, 28, 28, 1), 28 * 28 + 200)
    ref = DescriptorRefiner(384, 384, 128).eval()
    ref.load_state_dict({k: t(v) for k, v in synth.refiner_state(0).items()})
    bb = _backbone(28)
    with torch.no_grad():
        samp = bb.extract_at_keypoints(feat, t(gs["g28_kp"])[None])
        assert np.abs(samp[0, ::10].numpy() - gr["g28_sampled_sub"]).max() < 2e-5
        desc = ref(samp)
        assert desc.shape == (1, 500, 128)
        assert np.abs(desc[0].numpy() - gr["g28_desc"]).max() < 5e-6
        assert np.array_equal(bb.patch_to_pixel(t(gs["g28_kp"])).numpy(), gr["g28_pix"])
        assert np.array_equal(bb.pixel_to_patch(t(gr["g28_pix"])).numpy(), gr["g28_pix_back"])


def test_training_step_runs_through_the_drop_in_modules():
    """What train.py does (train.py:299-329, 239-244): grads reach selector and refiner, AdamW steps."""
    from models.descriptor_refiner import DescriptorRefiner
    from models.keypoint_selector import KeypointSelector
    torch.manual_seed(0)
    bb = _backbone(28)
    sel, ref = KeypointSelector(384, 256), DescriptorRefiner(384, 384, 128)
    opt = torch.optim.AdamW(list(sel.parameters()) + list(ref.parameters()), lr=1e-3)
    bb.dino.tokens = t(synth.tokens(5, 28, batch=2))
    with torch.no_grad():
        feat = bb(torch.zeros(2, 3, 448, 448))
    sal = sel(feat)
    kp, sc = sel.select_keypoints(sal, 64)
    desc = ref(bb.extract_at_keypoints(feat, kp))
    loss = sal.mean() + desc.var()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(sel.parameters(), 1.0)
    assert sel.conv[0].weight.grad is not None and ref.input_proj.weight.grad is not None
    before = ref.output_proj.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, ref.output_proj.weight)


# ------------------------------------------------------------------------------------------ device / shape guards
class _FakeCuda:
    """just enough of a tensor for lib.common_device: .is_cuda and .device"""

    def __init__(self, index):
        self.is_cuda, self.device = True, torch.device("cuda", index)


def test_lib_rejects_cpu_and_mixed_device_arguments():
    from sslam_amd import lib
    a, b = _FakeCuda(0), _FakeCuda(1)
    assert lib.common_device(a, None, _FakeCuda(0)) == torch.device("cuda", 0)
    with pytest.raises(ValueError, match="different devices"):
        lib.common_device(a, b)
    with pytest.raises(ValueError, match="CUDA tensors only"):
        lib.common_device(a, torch.zeros(3))
    with pytest.raises(ValueError):
        lib.common_device(None)
    # a wrapper handed CPU tensors raises a clean Python error before anything reaches the C ABI
    before = lib.launch_count()
    with pytest.raises(ValueError):
        lib.gather(torch.zeros(1, 4, 4, 384), torch.zeros(1, 3, 2))
    with pytest.raises(ValueError):
        lib.select_keypoints(torch.zeros(1, 4, 4), 3)
    assert lib.launch_count() == before


def test_vit_f32_weight_pack_is_a_permutation_in_fragment_order():
    """sslam_vit_f32_pack_linear_host (host code, no GPU): element (n, k) of an nn.Linear weight lands at
    [n/32][k/8][(k%8)/4][n%32][k%4] - every value exactly once; shapes the kernel cannot tile are refused."""
    from sslam_amd import lib
    rng = np.random.default_rng(0)
    for n_out, k_in in ((1152, 384), (384, 1536), (128, 96)):
        w = rng.standard_normal((n_out, k_in)).astype(np.float32)
        p = lib.pack_vit_f32_linear(w).reshape(n_out // 32, k_in // 8, 2, 32, 4)
        n, k = rng.integers(0, n_out, 200), rng.integers(0, k_in, 200)
        assert np.array_equal(p[n // 32, k // 8, (k % 8) // 4, n % 32, k % 4], w[n, k])
        assert np.array_equal(np.sort(p.ravel()), np.sort(w.ravel()))
    with pytest.raises(ValueError):
        lib.pack_vit_f32_linear(np.zeros((100, 96), np.float32))
    with pytest.raises(ValueError):
        lib.pack_vit_f32_linear(np.zeros((128, 48), np.float32))


def test_fp32_vit_launch_groups_are_whole_rounds_of_workgroups():
    """HipViTF32.chunk_frames: the per-layer GEMMs run one workgroup per (128-row tile, 128 columns) at three per CU - the launch
    group is the largest frame count whose row tiles (padded to the 8 XCDs) fill whole rounds of 768 workgroups for N = 384, 1152
    and 1536 alike (64 frames left the N = 384 GEMMs at 1.54 rounds)."""
    from sslam_amd.vit_hip import HipViTF32
    for size, want in ((448, 83), (640, 40), (960, 18)):
        n = HipViTF32.chunk_frames(size)
        assert n == want
        t = 5 + (size // 16) ** 2
        tiles = -(-(n * t) // 128)
        assert tiles <= 512 < -(-((n + 1) * t) // 128)              # the largest frame count within 512 row tiles
        for cols in (3, 9, 12):
            rounds = -(-tiles // 8) * 8 * cols / 768.0
            assert rounds <= round(rounds) + 1e-9 and (round(rounds) - rounds) / round(rounds) < 0.05     # < 5 % of the launch idle


def test_backbone_vit_precision_flag_and_device_guard():
    from models.dino_backbone import DinoBackbone
    from sslam_amd.vit import DinoV3ViT
    with pytest.raises(ValueError, match="vit_precision"):
        DinoBackbone(dino=TokenDino(), vit_precision="fp16")
    torch.manual_seed(0)
    vit = DinoV3ViT(depth=1)
    bb = DinoBackbone(input_size=32, dino=vit, vit_precision="bf16")
    x = torch.randn(1, 3, 32, 32)
    assert bb._hip_vit(x) is None                      # CPU images: never the HIP ViT
    # a CUDA image with a CPU-resident module must not hand host pointers to the kernels: _hip_vit says None and the
    # eager path raises torch's own device error (checked with a stand-in that only carries .is_cuda / .device)
    class _Img(_FakeCuda):
        pass
    assert bb._hip_vit(_Img(0)) is None
    assert DinoBackbone(input_size=32, dino=vit, vit_precision="fp32")._hip_vit(_Img(0)) is None
    assert DinoBackbone(input_size=32, dino=vit, vit_precision="eager")._hip_vit(_Img(0)) is None
    with torch.no_grad():
        f = bb(x)                                      # eager fp32 definition on CPU, all precisions agree there
        f32 = DinoBackbone(input_size=32, dino=vit, vit_precision="fp32")(x)
        fe = DinoBackbone(input_size=32, dino=vit, vit_precision="eager")(x)
    assert f.shape == (1, 2, 2, 384) and torch.equal(f, f32) and torch.equal(f, fe)


def test_unsupported_dims_fall_back_to_eager_on_cpu_too():
    from models.descriptor_refiner import DescriptorRefiner
    from models.keypoint_selector import KeypointSelector
    from sslam_amd.pipeline import PackedRefiner, PackedSelector
    assert PackedSelector.supported((256, 384, 3, 3)) and PackedSelector.supported((128, 384, 3, 3))
    assert not PackedSelector.supported((64, 384, 3, 3)) and not PackedSelector.supported((256, 256, 3, 3))
    assert PackedRefiner.supported(384, 384, 128, 2) and not PackedRefiner.supported(384, 256, 128, 2)
    assert not PackedRefiner.supported(384, 384, 64, 2)
    sel, ref = KeypointSelector(384, 64).eval(), DescriptorRefiner(384, 256, 64).eval()
    with torch.no_grad():
        assert sel(torch.randn(1, 6, 6, 384)).shape == (1, 6, 6, 1)
        assert ref(torch.randn(1, 5, 384)).shape == (1, 5, 64)


def test_streaming_scheduler_pair_bookkeeping():
    """The reference's pair set (visualize_matches_sequence.py:298-300) and the ring arithmetic, without a GPU."""
    from sslam_amd.harness import StreamingSequence
    assert StreamingSequence.reference_pairs(613, 5)[:3] == [0, 5, 10]
    assert StreamingSequence.reference_pairs(613, 5)[-1] == 605 and len(StreamingSequence.reference_pairs(613, 20)) == 30
    assert StreamingSequence.reference_pairs(613, 20, max_pairs=1) == [0]
    assert StreamingSequence.reference_pairs(3, 5) == []

    class _Pipe:                                   # records what the scheduler asks for; frames are their own index
        class cfg:
            spacing = 1

        def alloc_extract(self, n, with_intensity):
            return {"descriptors": torch.full((n,), -1.0), "scores": torch.full((n,), -1.0)}

        def alloc_match(self, n_pairs):
            return {"match_count": torch.full((n_pairs,), -1, dtype=torch.int32), "pair": torch.full((n_pairs, 2), -1.0)}

        def extract(self, tokens, images=None, out=None, images_ready=None):
            if out is None:
                return {"descriptors": tokens.clone(), "scores": tokens.clone()}
            out["descriptors"][:] = tokens
            out["scores"][:] = tokens
            return out

        def match(self, desc, scores, intensity=None, spacing=1, out=None):
            n = desc.shape[0] - spacing
            res = out if out is not None else self.alloc_match(n)
            res["match_count"][:] = 0
            res["pair"][:] = torch.stack([desc[:n], desc[spacing:]], 1)
            return dict(res)

    n = 31
    frames = torch.arange(n, dtype=torch.float32)
    for chunk in (None, 1, 4, 13, 40):
        # a sequence of known length: sequence-sized buffers written in place (every pair exactly once, whatever the chunking)
        seq = StreamingSequence(_Pipe(), (1, 5, 10, 15, 20))
        res = seq.run(frames, None, chunk=chunk)
        assert seq._ring is None and torch.equal(res["frames"]["descriptors"], frames)
        # an unbounded stream: the ring of the last max(spacings) frames
        ring = StreamingSequence(_Pipe(), (1, 5, 10, 15, 20))
        step = n if chunk is None else chunk
        outs = [ring.push(frames[a:a + step]) for a in range(0, n, step)]
        assert ring._ring["descriptors"].shape[0] == min(20, n)
        for s in (1, 5, 10, 15, 20):
            want = torch.stack([frames[:n - s], frames[s:]], 1)
            assert torch.equal(res[s]["pair"], want), (chunk, s)
            assert res[s]["first"].tolist() == list(range(n - s))
            assert int(res[s]["match_count"].min()) == 0                        # every row was written
            assert torch.equal(torch.cat([o[s]["pair"] for o in outs if s in o]), want), (chunk, s)
    with pytest.raises(ValueError):
        seq = StreamingSequence(_Pipe(), (1,))
        seq.reset(capacity=3)
        seq.push(frames[:4])


def test_chunk_bounds_and_feeder_arguments():
    from sslam_amd.harness import FrameFeeder, chunk_bounds
    assert chunk_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert chunk_bounds(10, 4, 2) == [(0, 2), (2, 6), (6, 10)]
    assert chunk_bounds(3, 8, 16) == [(0, 3)]
    assert chunk_bounds(0, 8) == []
    with pytest.raises(ValueError):
        FrameFeeder(4, 2, 2, "cpu", [(0, 4)])                                   # neither fill nor pinned_source


def test_spacing_summary_is_the_reference_loop_over_its_pair_subset():
    """harness.spacing_summary == the statistics process_spacing accumulates (visualize_matches_sequence.py:294-356): pairs
    (i, i + s) for i = 0, s, 2 s, ... up to max_pairs, all their match qualities pooled - restated here as the plain Python loop."""
    import numpy as np
    import torch
    from sslam_amd.harness import spacing_summary
    rng = np.random.default_rng(5)
    n, K = 23, 40
    res = {}
    for s in (1, 5, 10):
        p = n - s
        cnt = rng.integers(0, K + 1, p).astype(np.int32)
        cnt[::4] = 0
        q = rng.random((p, K)).astype(np.float32)
        q *= (np.arange(K)[None, :] < cnt[:, None])
        res[s] = dict(quality=torch.from_numpy(q), match_count=torch.from_numpy(cnt), matches=torch.zeros((p, K, 2), dtype=torch.int64))
    for s in (1, 5, 10):
        for max_pairs in (None, 1, 3, 100):
            pool, pairs = [], 0
            for i in range(0, n - s, s):                              # the reference's loop
                if max_pairs is not None and pairs >= max_pairs:
                    break
                c = int(res[s]["match_count"][i])
                pool.extend(res[s]["quality"][i, :c].tolist())
                pairs += 1
            got = spacing_summary(res, s, max_pairs)
            assert got["pairs"] == pairs and got["matches"] == len(pool)
            if pool:
                assert abs(got["mean_quality"] - float(np.mean(pool))) < 1e-6
                assert got["min_quality"] == np.float32(min(pool)) and got["max_quality"] == np.float32(max(pool))
                assert got["high_quality"] == sum(v > 0.8 for v in pool)
            else:
                assert got["mean_quality"] is None and got["high_quality"] == 0
    assert spacing_summary(res, 20)["pairs"] == 0                     # a spacing the result does not hold


def test_launch_groups_stay_below_one_buffer_descriptor():
    """The saliency CNN addresses a launch's fp32 feature map through ONE 32-bit buffer descriptor: a launch group must stay below
    4 GiB.  BASELINE configs[2] at its full length (2 965 frames at G = 40: 7.3 GB of features) is therefore cut into launch
    groups by ExtractorConfig.launch_group(), which pipeline.extract, the sharded runner and bench.py's staged extraction all
    iterate by (round 4 hit 'selector_saliency: unsupported shape' before they did)."""
    import inspect

    import bench
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    for name, (n, h, w, size, K) in bench.WORKLOADS.items():
        cfg = ExtractorConfig(input_size=size, num_keypoints=K)
        lg = cfg.launch_group()
        assert 1 <= lg <= cfg.chunk_frames
        assert lg * cfg.grid ** 2 * 384 * 4 < 2 ** 32, name
    assert ExtractorConfig(input_size=640, num_keypoints=1024).launch_group() == 1024        # 1024 x 1600 x 1536 B = 2.5 GB
    assert ExtractorConfig(input_size=960, num_keypoints=2048, chunk_frames=4096).launch_group() == 776   # capped: 776 x 3600 x 1536 B < 4 GiB
    assert "launch_group()" in inspect.getsource(SequencePipeline.extract)
    src = inspect.getsource(bench.main)
    assert "step_ = pipe.launch_group()" in src and "range(0, t.shape[0], step_)" in src


def test_pipeline_rejects_an_unknown_vit_precision_even_without_a_vit():
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    with pytest.raises(ValueError, match="vit_precision"):
        SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cpu", vit_precision="fp23")

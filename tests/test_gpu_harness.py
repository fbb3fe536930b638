"""GPU tests of the reference-shaped harness (SURVEY §8f-3 and the glue around the hot path):

* `StreamingSequence`: >= 25 frames, spacings (1, 5, 10, 15, 20) as at visualize_matches_sequence.py:369 - every pair's
  matches / quality bit-exact vs the oracle's match_with_quality on the same descriptors, for one-shot and chunked
  pushes alike; the reference's own pair set (process_spacing, :298-300) is a row subset of the result.
* `SequenceMatcher.extract(path)`: PNG on disk -> PIL decode -> HIP A0 -> ViT -> A2..A9, the reference's dict keys and
  dtypes (visualize_matches_sequence.py:97-104), values vs the oracle chain on the same tokens.
* ViT precision: keypoint-set and match agreement of the bf16 HIP ViT path against the fp32 definition (reported and
  bounded - it is NOT bit-exact and never claimed to be).
* Drop-in shape fallback: KeypointSelector(384, 64) / DescriptorRefiner(hidden 256) on cuda run the eager path.
"""
import numpy as np
import pytest

import synth
from oracle import ora

pytestmark = pytest.mark.gpu

SPACINGS = (1, 5, 10, 15, 20)
CLI = dict(saliency_weight=0.3, min_saliency=0.5, min_descriptor_sim=0.7, min_intensity=0.15)   # :381-388


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def pipe(T):
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    return SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cuda")


def _oracle_pair(fr, i, j):
    return ora.match_with_quality(fr["descriptors"][i], fr["descriptors"][j], fr["scores"][i], fr["scores"][j],
                                  CLI["saliency_weight"], CLI["min_saliency"], CLI["min_descriptor_sim"],
                                  fr["intensity"][i], fr["intensity"][j], CLI["min_intensity"])


def _check_spacing(res, s, n, fr):
    mm = {k: v.cpu().numpy() for k, v in res[s].items()}
    assert mm["first"].tolist() == list(range(n - s)), s
    assert mm["match_count"].shape == (n - s,)
    total = 0
    for row, i in enumerate(mm["first"]):
        want_m, want_q = _oracle_pair(fr, i, i + s)
        c = int(mm["match_count"][row])
        assert c == len(want_m), (s, i, c, len(want_m))
        assert mm["matches"].dtype == np.int64 and np.array_equal(mm["matches"][row, :c], want_m), (s, i)
        assert np.array_equal(mm["quality"][row, :c].view(np.uint32), want_q.view(np.uint32)), (s, i)
        total += c
    return total


@pytest.mark.parametrize("chunk", [None, 7, 1])
def test_streaming_sequence_all_spacings_vs_oracle(T, pipe, chunk):
    from sslam_amd.harness import StreamingSequence
    n = 27
    toks, imgs = synth.token_sequence(n, 28), synth.image_sequence(n)
    seq = StreamingSequence(pipe, SPACINGS)
    res = seq.run(T.from_numpy(toks).cuda(), T.from_numpy(imgs).cuda(), chunk=chunk)
    fr = {k: res["frames"][k].cpu().numpy() for k in ("descriptors", "scores", "intensity", "idx")}
    # extraction itself: bit-exact vs the oracle chain (spot: three frames)
    feat = ora.bn_tokens(toks[[0, 13, 26]])[0].reshape(3, 28, 28, 384)
    _, _, oidx, _ = ora.select_keypoints(ora.selector_saliency(feat, synth.selector_state(0)), 500)
    assert np.array_equal(fr["idx"][[0, 13, 26]], oidx)
    totals = {s: _check_spacing(res, s, n, fr) for s in SPACINGS}
    assert totals[1] > 0, "consecutive synthetic frames must have mutual matches"
    # the reference's process_spacing visits rows 0, s, 2s, ... (and stops after max_pairs)
    assert StreamingSequence.reference_pairs(n, 5) == [0, 5, 10, 15, 20]
    assert StreamingSequence.reference_pairs(n, 5, max_pairs=1) == [0]
    assert StreamingSequence.reference_pairs(n, 20) == [0]
    assert seq.n_seen == n and seq._ring is None            # a sequence of known length lives in sequence-sized buffers
    # the unbounded-stream mode (ring of the last 20 frames) gives the same rows
    ring = StreamingSequence(pipe, SPACINGS)
    outs = [ring.push(T.from_numpy(toks[a:a + 9]).cuda(), T.from_numpy(imgs[a:a + 9]).cuda()) for a in range(0, n, 9)]
    assert ring.n_seen == n and ring._ring["descriptors"].shape[0] == 20
    for s in SPACINGS:
        for key in ("matches", "quality", "match_count", "first"):
            got = T.cat([o[s][key] for o in outs if s in o])
            assert T.equal(got, res[s][key]), (s, key)


def test_streaming_short_sequence_skips_long_spacings(T, pipe):
    from sslam_amd.harness import StreamingSequence
    toks = synth.token_sequence(6, 28)
    res = StreamingSequence(pipe, SPACINGS).run(T.from_numpy(toks).cuda(), None, chunk=4)
    assert set(res) == {"frames", 1, 5}
    assert res[5]["match_count"].shape == (1,) and res[1]["match_count"].shape == (5,)


def test_sequence_matcher_extract_from_png(T, tmp_path):
    """The reference-shaped entry point end to end: a PNG on disk -> the dict of visualize_matches_sequence.py:97-104."""
    from PIL import Image

    from models.dino_backbone import DinoBackbone
    from sslam_amd import lib
    from sslam_amd.harness import SequenceMatcher
    from sslam_amd.vit import DinoV3ViT
    T.manual_seed(11)
    imgs = synth.image_sequence(2)
    paths = []
    for i in range(2):
        paths.append(str(tmp_path / f"{i:04d}.png"))
        Image.fromarray(imgs[i]).save(paths[-1])
    bb = DinoBackbone(input_size=448, dino=DinoV3ViT().eval())
    ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
    sm = SequenceMatcher(bb, ssd, rsd, device="cuda")
    before = lib.launch_count()
    f1, f2 = sm.extract(paths[0]), sm.extract(paths[1])
    assert lib.launch_count() >= before + 2 * 60, "the HIP ViT and the HIP stages must have served extract()"
    assert set(f1) == {"image", "saliency", "keypoints_pixel", "scores", "intensity", "descriptors"}
    assert f1["saliency"].shape == (28, 28) and f1["keypoints_pixel"].shape == (500, 2) and f1["scores"].shape == (500,)
    assert f1["intensity"].shape == (500,) and f1["descriptors"].shape == (500, 128)
    assert all(f1[k].dtype == np.float32 for k in ("saliency", "keypoints_pixel", "scores", "intensity", "descriptors"))
    assert f1["image"].size == (640, 480)
    # values: the oracle chain on the tokens the backbone produced for the same file
    for path, f, im in ((paths[0], f1, imgs[0]), (paths[1], f2, imgs[1])):
        with T.no_grad():
            tok = bb.forward_tokens(sm.pipe.preprocess(T.from_numpy(im[None]).cuda())).cpu().numpy()
        feat = ora.bn_tokens(tok)[0].reshape(1, 28, 28, 384)
        sal = ora.selector_saliency(feat, ssd)
        kp, sc, idx, _ = ora.select_keypoints(sal, 500)
        desc = ora.refine(ora.gather(feat, kp), rsd)
        assert np.array_equal(f["saliency"].view(np.uint32), sal[0].view(np.uint32))
        assert np.array_equal(f["keypoints_pixel"], ora.patch_to_pixel(kp[0]))
        assert np.array_equal(f["scores"].view(np.uint32), sc[0].view(np.uint32))
        assert np.array_equal(f["descriptors"].view(np.uint32), desc[0].view(np.uint32))
        assert np.array_equal(f["intensity"], ora.intensity(im, 448, ora.patch_to_pixel(kp[0])))
    m, q = sm.match_with_quality(f1["descriptors"], f2["descriptors"], f1["scores"], f2["scores"], intensity1=f1["intensity"],
                                 intensity2=f2["intensity"], **CLI)
    wm, wq = ora.match_with_quality(f1["descriptors"], f2["descriptors"], f1["scores"], f2["scores"], 0.3, 0.5, 0.7,
                                    f1["intensity"], f2["intensity"], 0.15)
    assert m.dtype == np.int64 and q.dtype == np.float32 and np.array_equal(m, wm) and np.array_equal(q, wq)


@pytest.mark.parametrize("tag", ["e2e_g40", "e2e_g60"])
def test_reference_shaped_harness_on_larger_grids(T, tag):
    """The reference's own call shape - SequenceMatcher.extract frame by frame, then match_with_quality pair by pair with the CLI
    thresholds (visualize_matches_sequence.py:69-104, 106-197, 381-388) - through the drop-in DinoBackbone (input_size 640 /
    960), harness.SequenceMatcher and matching.match_with_quality, against the reference's end-to-end goldens at G = 40 / K = 1024
    and G = 60 / K = 2048 (tests/e2e_check.py holds the bars: sets, order up to near-tie swaps, values by cell, matches as cell
    pairs and - where both frames kept their order - as indices)."""
    import e2e_check
    import matching
    from models.dino_backbone import DinoBackbone
    from sslam_amd.harness import SequenceMatcher
    from sslam_amd.pipeline import ExtractorConfig
    from test_models_api import TokenDino
    g = e2e_check.gold(tag)
    grid, K, n = int(g["grid"]), int(g["K"]), int(g["n_frames"])
    toks = synth.token_sequence(n, grid)
    imgs = synth.image_sequence(n, int(g["height"]), int(g["width"]))
    bb = DinoBackbone(input_size=16 * grid, dino=TokenDino(), vit_precision="eager")      # tokens are the input here (a stand-in ViT)
    sm = SequenceMatcher(bb, synth.selector_state(0), synth.refiner_state(0), cfg=ExtractorConfig(input_size=16 * grid, num_keypoints=K))
    frames = []
    for i in range(n):
        bb.dino.tokens = T.from_numpy(toks[i:i + 1]).cuda()
        out = sm.extract_batch(imgs[i:i + 1])
        frames.append({k: v[0].cpu().numpy() for k, v in out.items()})
    px = np.stack([f["keypoints_pixel"] for f in frames])
    cell = ((px - 8) / 16).astype(np.int64)
    idx = cell[..., 1] * grid + cell[..., 0]
    sc = np.stack([f["scores"] for f in frames])
    desc = np.stack([f["descriptors"] for f in frames])
    inten = np.stack([f["intensity"] for f in frames])

    def match(a, b):
        return matching.match_with_quality(desc[a], desc[b], sc[a], sc[b], intensity1=inten[a], intensity2=inten[b], **e2e_check.CLI)
    rep = e2e_check.check_sequence(tag, idx, sc, desc, inten, match)
    assert len(rep["pairs"]) >= 8 and all(p["cells_equal"] for p in rep["pairs"])


def _assert_same(T, a: dict, b: dict, keys):
    for k in keys:
        assert a[k].dtype == b[k].dtype and T.equal(a[k], b[k]), k


def test_run_directory_equals_resident_run(T, pipe, tmp_path):
    """Directory -> matches (TUMSequence -> PNG decode on a thread pool -> pinned double buffer -> H2D on a side stream ->
    StreamingSequence) is bit-equal to the same frames already resident in HBM, for every spacing, whatever the chunking
    (reference: main -> process_spacing -> extract(path) -> match, visualize_matches_sequence.py:272-357, 360-448)."""
    from sslam_amd.harness import StreamingSequence, run_directory, run_frames
    n = 23
    toks_h, imgs_h = synth.token_sequence(n, 28), synth.image_sequence(n)
    names = synth.write_tum_rgb_sequence(str(tmp_path / "seq"), imgs_h)
    toks, imgs = T.from_numpy(toks_h).cuda(), T.from_numpy(imgs_h).cuda()
    want = StreamingSequence(pipe, SPACINGS).run(toks, imgs)
    want = {k: ({kk: vv.clone() for kk, vv in v.items()}) for k, v in want.items()}
    one = pipe.run(imgs, toks)                              # the plain resident pass (spacing 1)
    for chunk in (5, 64):
        got = run_directory(str(tmp_path), "seq", SPACINGS, pipe=pipe, tokens_fn=lambda a, b: toks[a:b], chunk=chunk, decode_workers=4)
        assert got["files"] == names and len(got["timestamps"]) == n
        _assert_same(T, got["frames"], want["frames"], ("idx", "descriptors", "scores", "intensity", "keypoints_pixel"))
        for s in SPACINGS:
            _assert_same(T, got[s], want[s], ("matches", "quality", "match_count"))
        _assert_same(T, got[1], one, ("matches", "quality", "match_count"))
    # the per-spacing statistics process_spacing prints (:345-356), on device tensors, against the oracle on the reference's pairs
    from sslam_amd.harness import spacing_summary
    fr = {k: v.cpu().numpy() for k, v in got["frames"].items()}          # (equal to the oracle's: checked frame by frame above)
    for s, max_pairs in ((5, 2), (10, None)):
        pool = []
        first = StreamingSequence.reference_pairs(n, s, max_pairs)
        for i in first:
            pool.extend(_oracle_pair(fr, i, i + s)[1].tolist())
        sm = spacing_summary(got, s, max_pairs)
        assert sm["pairs"] == len(first) and sm["matches"] == len(pool) and sm["high_quality"] == sum(v > 0.8 for v in pool)
        if pool:
            assert abs(sm["mean_quality"] - float(np.mean(pool))) < 1e-6 and sm["max_quality"] == np.float32(max(pool))
    # host-resident frames (pinned, uploaded straight from the array; and pageable through the staging buffers)
    pinned = T.from_numpy(imgs_h).pin_memory()
    got = run_frames(pipe, n, 480, 640, spacings=(1,), tokens=toks, pinned_source=pinned, chunk=6, first_chunk=2)
    _assert_same(T, got[1], one, ("matches", "quality", "match_count"))
    got = run_frames(pipe, n, 480, 640, spacings=(1,), tokens=toks, chunk=7,
                     fill=lambda dst, a, b: np.copyto(dst, imgs_h[a:b]))
    _assert_same(T, got[1], one, ("matches", "quality", "match_count"))
    _assert_same(T, got["frames"], one, ("idx", "descriptors", "intensity"))
    # long / unbounded sequences: the device side is a ring of chunk slots reused behind the compute (release events)
    for kw in (dict(pinned_source=pinned), dict(fill=lambda dst, a, b: np.copyto(dst, imgs_h[a:b]))):
        got = run_frames(pipe, n, 480, 640, spacings=(1, 5), tokens=toks, chunk=3, first_chunk=1, preprocess_too=True,
                         feeder_kw=dict(max_bytes=0, ring=3), **kw)
        _assert_same(T, got[1], one, ("matches", "quality", "match_count"))
        _assert_same(T, got[5], want[5], ("matches", "quality", "match_count"))
        _assert_same(T, got["frames"], one, ("idx", "descriptors", "intensity"))


def test_back_to_back_feeds_share_the_staging_buffer_safely(T, pipe):
    """Two run_frames(fill=...) calls on the same device with NO host synchronisation between them: the pinned staging buffer
    is cached per device, so the second feeder must not overwrite halves whose last uploads (the first sequence's last chunks)
    are still in flight - the upload events live beside the buffer, not in the feeder instance.  Whole-buffer and ring mode,
    different chunk sizes (the halves of the two feeders overlap differently), results bit-equal to the resident passes."""
    from sslam_amd.harness import run_frames
    n = 21
    toks_h, imgs_h = synth.token_sequence(n, 28), synth.image_sequence(n)
    imgs_r = np.ascontiguousarray(imgs_h[::-1])
    toks, toks_r = T.from_numpy(toks_h).cuda(), T.from_numpy(np.ascontiguousarray(toks_h[::-1])).cuda()
    one = {k: v.clone() for k, v in pipe.run(T.from_numpy(imgs_h).cuda(), toks).items()}
    rev = {k: v.clone() for k, v in pipe.run(T.from_numpy(imgs_r).cuda(), toks_r).items()}
    T.cuda.synchronize()
    for kw in (dict(max_bytes=0, ring=2), dict()):
        a = run_frames(pipe, n, 480, 640, spacings=(1,), tokens=toks, chunk=8, fill=lambda dst, lo, hi: np.copyto(dst, imgs_h[lo:hi]),
                       feeder_kw=dict(kw))
        b = run_frames(pipe, n, 480, 640, spacings=(1,), tokens=toks_r, chunk=5, fill=lambda dst, lo, hi: np.copyto(dst, imgs_r[lo:hi]),
                       feeder_kw=dict(kw))
        c = run_frames(pipe, n, 480, 640, spacings=(1,), tokens=toks, chunk=8, fill=lambda dst, lo, hi: np.copyto(dst, imgs_h[lo:hi]),
                       feeder_kw=dict(kw))
        for got, want in ((a, one), (b, rev), (c, one)):
            _assert_same(T, got[1], want, ("matches", "quality", "match_count"))
            _assert_same(T, got["frames"], want, ("idx", "descriptors", "intensity"))


def test_run_directory_with_the_hip_vit(T, tmp_path):
    """images on disk -> A0 -> HIP ViT (A1) -> A2 .. M1: equal to the resident pass over the same launch groups."""
    from sslam_amd.harness import run_directory
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    from sslam_amd.vit import DinoV3ViT
    T.manual_seed(3)
    n = 12
    imgs_h = synth.image_sequence(n)
    synth.write_tum_rgb_sequence(str(tmp_path / "seq"), imgs_h)
    pv = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cuda", vit=DinoV3ViT().cuda().eval())
    got = run_directory(str(tmp_path / "seq"), "", (1, 5), pipe=pv, chunk=12)       # first chunk: min(chunk, 16) = 12 frames
    imgs = T.from_numpy(imgs_h).cuda()
    want = pv.run(imgs)
    _assert_same(T, got["frames"], want, ("idx", "descriptors", "intensity"))
    _assert_same(T, got[1], want, ("matches", "quality", "match_count"))
    assert int(got[1]["match_count"].sum()) > 0 and got[5]["match_count"].shape == (n - 5,)


def test_vit_precision_agreement(T, pipe):
    """bf16 HIP ViT vs the fp32 definition (eager torch) on the same weights and frames: how many keypoints / matches agree.
    (Random ViT weights give a flat, noise-like saliency field - the hardest case for a discontinuous selector.)
    And the fp32-operand HIP ViT (vit_precision="fp32", the reference's numerics on the HIP kernels) against the same
    eager evaluation: tokens within 1e-4, keypoint sets and matches (as cell pairs) >= 99.9 % identical."""
    from models.dino_backbone import DinoBackbone
    from sslam_amd.vit import DinoV3ViT
    T.manual_seed(5)
    vit = DinoV3ViT().eval()
    imgs = T.from_numpy(synth.image_sequence(6)).cuda()
    outs = {}
    for prec in ("bf16", "fp32", "eager"):
        bb = DinoBackbone(input_size=448, dino=vit, vit_precision=prec).cuda()
        with T.no_grad():
            tok = bb.forward_tokens(pipe.preprocess(imgs)).float().contiguous()
        outs[prec] = (tok, {k: v.clone() for k, v in pipe.run(imgs, tok).items()})
    hip32 = outs["fp32"]
    outs["fp32"] = outs["eager"]               # the reference evaluation below is the eager one
    rel32 = float((hip32[0] - outs["eager"][0]).norm() / outs["eager"][0].norm())
    i_h, i_e = hip32[1]["idx"].cpu().numpy(), outs["eager"][1]["idx"].cpu().numpy()
    kp32 = np.mean([len(set(a) & set(b)) / len(set(b)) for a, b in zip(i_h, i_e)])
    print(f"\nViT fp32 HIP vs eager: token rel err {rel32:.2e}, keypoint-set agreement {kp32:.4f}")
    assert rel32 <= 1e-4 and kp32 >= 0.999
    tb, tf = outs["bf16"][0], outs["fp32"][0]
    rel = float((tb - tf).norm() / tf.norm())
    ib, i32 = outs["bf16"][1]["idx"].cpu().numpy(), outs["fp32"][1]["idx"].cpu().numpy()
    kp_agree = np.mean([len(set(a) & set(b)) / len(set(b)) for a, b in zip(ib, i32)])
    mb, mf = outs["bf16"][1], outs["fp32"][1]
    agree = tot = 0
    for p in range(5):
        cb, cf = int(mb["match_count"][p]), int(mf["match_count"][p])
        sb = {(int(ib[p][a]), int(ib[p + 1][b])) for a, b in mb["matches"][p, :cb].cpu().numpy()}     # as grid cells
        sf = {(int(i32[p][a]), int(i32[p + 1][b])) for a, b in mf["matches"][p, :cf].cpu().numpy()}
        agree += len(sb & sf)
        tot += max(len(sf), 1)
    print(f"\nViT bf16 vs fp32: token rel err {rel:.2e}, keypoint-set agreement {kp_agree:.3f}, "
          f"match agreement (cell pairs) {agree / tot:.3f}")
    assert rel < 2.5e-2
    assert kp_agree > 0.80, kp_agree
    assert agree / tot > 0.90, agree / tot        # the default bf16 ViT keeps >= 90 % of the fp32 definition's matches (measured 0.99)


def test_unsupported_dims_take_the_eager_path(T):
    from models.descriptor_refiner import DescriptorRefiner
    from models.keypoint_selector import KeypointSelector
    from sslam_amd import lib
    T.manual_seed(0)
    sel = KeypointSelector(384, 64).cuda().eval()
    ref = DescriptorRefiner(384, 256, 64).cuda().eval()
    feat = T.randn(2, 28, 28, 384, device="cuda")
    before = lib.launch_count()
    with T.no_grad(), pytest.warns(UserWarning, match="eager torch path"):
        sal = sel(feat)
    with T.no_grad(), pytest.warns(UserWarning, match="eager torch path"):
        d = ref(T.randn(2, 50, 384, device="cuda"))
    assert lib.launch_count() == before, "unsupported shapes must not reach the HIP kernels"
    want = T.sigmoid(sel.conv(feat.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    assert sal.shape == (2, 28, 28, 1) and T.allclose(sal, want, atol=1e-6)
    assert d.shape == (2, 50, 64) and T.allclose(d.norm(dim=-1), T.ones(2, 50, device="cuda"), atol=1e-5)
    # the keypoint selection on that saliency still runs on the HIP kernel (shape independent)
    kp, sc = sel.select_keypoints(sal, 100)
    assert kp.shape == (2, 100, 2) and lib.launch_count() == before + 1


def test_pipeline_raises_when_k_cannot_be_served(T):
    """input_size 224 (196 cells) with K = 500: torch.topk raises in the reference; the batched pipeline does too."""
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    p14 = SequencePipeline(ExtractorConfig(input_size=224, num_keypoints=500), synth.selector_state(0), synth.refiner_state(0),
                           device="cuda")
    toks = T.from_numpy(synth.token_sequence(2, 14)).cuda()
    with pytest.raises(RuntimeError, match="out of range"):
        p14.extract(toks)


def test_two_rank_rehearsal_on_one_gpu(T):
    """The whole N-rank path of bench.py - it starts its own ranks, rank 0 packs and broadcasts the weights, boundary frames go
    first, halo exchange, compacted gather to rank 0 - rehearsed with two ranks sharing this GPU over gloo (RCCL refuses two ranks
    on one GPU; device buffers are staged through host memory): the gathered result equals the single-process pass pair for
    pair, and the line carries no value (a rehearsal is not a measurement)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--frames", "21",
                        "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["value"] is None and d["n_gpus"] == 2 and d["n_ranks_seen"] == 2
    assert d["frame_ranges"] == [[0, 21], [21, 42]]
    # the gate of an N-rank line: every rank's whole block against the oracle (21 frames, 20 pairs inside each), the pair that crosses
    # the shard boundary (frames 20 | 21, regenerated on rank 0 from the seed) against the oracle through the GATHERED arrays, and the
    # gathered rows of every rank against a digest of what that rank computed
    par = d["parity"]
    assert par["bit_exact"] and par["first_mismatch"] is None
    assert par["ranks_checked"] == 2 and par["frames_checked_vs_oracle"] == 42
    assert par["boundaries"] == 1 and par["boundary_pairs_checked"] == 1 and par["boundary_matches_checked"] > 0
    assert par["pairs_checked"] == 41 and par["pairs_total"] == 41 and par["gathered_rows_equal_every_ranks_local_result"]
    # the N > 1 legs a scaling run reports beside `value`: the per-rank host feed and the sharded step with the HIP ViT inside
    assert d["with_upload"]["equal_to_resident_pass_on_every_rank"] and d["with_upload"]["value"] > 0
    assert d["with_vit"]["pairs_gathered_on_rank0"] == 41 and d["with_vit"]["value"] > 0
    reh = d["rehearsal"]
    assert reh["sharded_equals_single_process"] and reh["pairs"] == 41 and reh["pairs_per_rank"] == [21, 20] and reh["matches"] > 0


def test_rehearsal_gate_refuses_a_corrupted_boundary_pair(T):
    """The same rehearsal with ONE match index of the cross-rank pair flipped in the gathered result before the gate: no value,
    exit code 1, and the mismatch is attributed to the boundary (the rank-0-local check of round 4 would not have seen it)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--frames", "9",
                        "--steps", "1", "--warmup", "1", "--test-corrupt-gathered"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["value"] is None and not d["parity"]["bit_exact"]
    assert "boundary of rank 1" in d["parity"]["first_mismatch"] or "gathered rows" in d["parity"]["first_mismatch"]


_RCCL_ONE_RANK = r'''
import os, sys
root = sys.argv[1]
for p in (root, os.path.join(root, "semantic-slam-master_amd"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import torch, torch.distributed as dist
import synth
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
from sslam_amd.shard import ShardedSequenceRunner, _P2P, _host_staged, pipeline_from_rank0

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and not _host_staged()
cfg = ExtractorConfig()
ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
pipe = pipeline_from_rank0(cfg, ssd, rsd, dev)                    # header + packed weights through ncclBroadcast
ref = SequencePipeline(cfg, ssd, rsd, device=dev)
for a, b in zip(pipe.weight_tensors(), ref.weight_tensors()):
    assert torch.equal(a, b)
n = 9
toks = torch.from_numpy(synth.token_sequence(n, 28)).to(dev)
imgs = torch.from_numpy(synth.image_sequence(n)).to(dev)
# device buffers handed to RCCL as they are: a self-addressed batch of the same shapes and dtypes the halo exchange and the
# record gather use (fp32 descriptors / scores, int32 records; sends and receives posted in ONE batch_isend_irecv)
ex = pipe.extract(toks, imgs)
halo = {k: torch.empty_like(ex[k][:1]) for k in ("descriptors", "scores", "intensity")}
rec = torch.arange(4 * 37, dtype=torch.int32, device=dev).reshape(37, 4)
rec_in = torch.empty_like(rec)
p2p = _P2P()
for k, v in halo.items():
    p2p.send(ex[k][:1], 0)
    p2p.recv(v, 0)
p2p.send(rec, 0)
p2p.recv(rec_in, 0)
# ... and of the padded gather: int64 match slots, int32 counts, received into a slice of a larger buffer
mt = torch.arange(3 * 500 * 2, dtype=torch.int64, device=dev).reshape(3, 500, 2)
cn = torch.tensor([7, 0, 500], dtype=torch.int32, device=dev)
mt_all, cn_all = torch.zeros((5, 500, 2), dtype=torch.int64, device=dev), torch.zeros(5, dtype=torch.int32, device=dev)
p2p.send(mt, 0)
p2p.recv(mt_all[2:5], 0)
p2p.send(cn, 0)
p2p.recv(cn_all[2:5], 0)
p2p.post().wait()
torch.cuda.synchronize()
assert torch.equal(mt_all[2:5], mt) and torch.equal(cn_all[2:5], cn) and int(mt_all[:2].abs().sum()) == 0
for k, v in halo.items():
    assert torch.equal(v, ex[k][:1]), k
assert torch.equal(rec_in, rec)
sizes = torch.tensor([5, 7], dtype=torch.int64, device=dev)
got = [torch.zeros_like(sizes)]
dist.all_gather(got, sizes)
assert got[0].tolist() == [5, 7]
out = ShardedSequenceRunner(pipe.extract, pipe.match, spacing=1).run(toks, imgs)
one = ref.run(imgs, toks)
for k in ("idx", "descriptors", "matches", "quality", "match_count"):
    assert torch.equal(out[k], one[k]), k
dist.barrier()
dist.destroy_process_group()
print("rccl-one-rank ok")
'''


def test_rccl_group_of_one_rank(T):
    """What CAN be run of the RCCL path on a one-GPU box: a process group on backend "nccl" (= RCCL) with one rank.  The weight
    header + packed buffers go through the broadcast, the size exchange through all_gather, and device buffers of the halo /
    record shapes and dtypes through one batch_isend_irecv addressed to the rank itself - no host staging (that is the gloo
    rehearsal's).  Two ranks on two GPUs over xGMI remain the driver's round-end run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl-one-rank ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("use_graph", [False, True])
def test_online_stepper_tokens_in_equals_the_batched_run(T, pipe, use_graph):
    """The online caller's loop (test/test_tracking.py:146-178: one frame at a time, the previous frame's descriptors kept):
    FrameStepper's per-frame outputs and its matches against the previous frame are bit-identical to the same frames going
    through SequencePipeline.run as one batch - as ordinary launches and replayed from a captured HIP graph (0 library calls
    per frame after the capture)."""
    from sslam_amd import lib
    from sslam_amd.online import FrameStepper
    n = 6
    imgs = T.from_numpy(synth.image_sequence(n)).cuda()
    toks = T.from_numpy(synth.token_sequence(n, 28)).cuda()
    want = pipe.run(imgs, tokens=toks)
    st = FrameStepper(pipe, 480, 640, use_graph=use_graph, tokens_in=True)
    with pytest.raises(ValueError, match="tokens"):
        st.step(imgs[0])
    st.reset()
    for rnd in range(2):                       # a second pass after reset(): the first frame has no previous one again
        for i in range(n):
            n0 = lib.launch_count()
            o = st.step(imgs[i].cpu() if i == 2 else imgs[i], toks[i])        # a host-resident frame is accepted too
            calls = lib.launch_count() - n0
            if use_graph and (rnd or i):
                assert calls == 0, "a replayed step issues no library call"
            for k in ("idx", "descriptors", "intensity", "scores", "saliency", "keypoints_pixel"):
                assert T.equal(o[k], want[k][i]), (k, i)
            if i == 0:
                assert o["matches"] is None and o["match_count"] is None
            else:
                c = int(o["match_count"])
                assert c == int(want["match_count"][i - 1]) and c > 0
                assert T.equal(o["matches"][:c], want["matches"][i - 1][:c]) and T.equal(o["quality"][:c], want["quality"][i - 1][:c])
        st.reset()


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_online_stepper_with_the_hip_vit(T, prec):
    """The same loop with A0 + A1 inside the step (images in): graph replay == ordinary launches == the batched run, bit for bit,
    for both ViT forms (a frame's tokens do not depend on what else is in the launch group)."""
    from sslam_amd.online import FrameStepper
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    from sslam_amd.vit import DinoV3ViT
    T.manual_seed(3)
    n = 4
    imgs = T.from_numpy(synth.image_sequence(n)).cuda()
    pv = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cuda",
                          vit=DinoV3ViT().cuda().eval(), vit_precision=prec)
    want = pv.run(imgs)
    for use_graph in (False, True):
        st = FrameStepper(pv, 480, 640, use_graph=use_graph)
        for i in range(n):
            o = st.step(imgs[i])
            for k in ("idx", "descriptors", "intensity", "scores"):
                assert T.equal(o[k], want[k][i]), (use_graph, k, i)
            if i:
                c = int(o["match_count"])
                assert c == int(want["match_count"][i - 1]) and T.equal(o["matches"][:c], want["matches"][i - 1][:c])
    with pytest.raises(Exception, match="without a ViT"):
        FrameStepper(SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cuda"), 480, 640)


@pytest.mark.parametrize("mode", ["tokens", "bf16", "fp32"])
def test_online_stepper_graph_survives_the_pipeline_replacing_its_buffers(T, mode):
    """A captured step bakes in the addresses of buffers the stepper does not own (the pipeline's scratch, the ViT's workspace), and
    their owners grow them BY REPLACEMENT.  Capture at one frame, then push a 20-frame batch through the SAME pipeline (19 pairs: the
    batched matcher's key scratch; 20 frames: a larger ViT workspace), fill whatever the allocator got back with junk, and keep
    stepping: the replayed graph must still give the batched result (the stepper holds its own references from the capture on)."""
    from sslam_amd.online import FrameStepper
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    from sslam_amd.vit import DinoV3ViT
    T.manual_seed(5)
    n = 20
    imgs = T.from_numpy(synth.image_sequence(n)).cuda()
    toks = T.from_numpy(synth.token_sequence(n, 28)).cuda() if mode == "tokens" else None
    kw = {} if mode == "tokens" else dict(vit=DinoV3ViT().cuda().eval(), vit_precision=mode)
    pv = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0), device="cuda", **kw)
    st = FrameStepper(pv, 480, 640, use_graph=True, tokens_in=(mode == "tokens"))
    tk = (lambda i: toks[i]) if mode == "tokens" else (lambda i: None)
    first = [{k: v.clone() for k, v in st.step(imgs[i], tk(i)).items() if v is not None} for i in range(2)]
    ws_bytes = 0 if pv._ws is None else pv._ws.numel()            # (no reference kept here: the stepper's own must be what keeps it alive)
    big = pv.run(imgs, tokens=toks) if mode == "tokens" else pv.run(imgs)
    assert pv._ws.numel() > ws_bytes, "the batch was meant to outgrow the scratch"
    del big
    # the reference result: the first 8 frames as one batch (the ViTs' few-frame launch forms - two-launch MLP of the bf16 form,
    # key-split attention of the fp32 form - cover 1..8 frames, so a one-frame step and this batch agree bit for bit)
    want = pv.run(imgs[:8], tokens=toks[:8]) if mode == "tokens" else pv.run(imgs[:8])
    T.cuda.synchronize()
    junk = [T.full((sz,), 0xA5, dtype=T.uint8, device="cuda") for sz in (1 << 12, 1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 26) for _ in range(3)]
    T.cuda.synchronize()
    for i in range(2):
        for k in ("idx", "descriptors", "scores"):
            assert T.equal(first[i][k], want[k][i]), (mode, k, i)
    for i in range(2, 8):
        o = st.step(imgs[i], tk(i))
        for k in ("idx", "descriptors", "intensity", "scores"):
            assert T.equal(o[k], want[k][i]), (mode, k, i)
        c = int(o["match_count"])
        assert c == int(want["match_count"][i - 1]) and T.equal(o["matches"][:c], want["matches"][i - 1][:c]), (mode, i)
    assert all(int(j[0]) == 0xA5 and int(j[-1]) == 0xA5 for j in junk), "the replay wrote into memory it no longer owns"

"""CPU test of the frame-sharded N > 1 path (sslam_amd/shard.py) with torch.distributed on the gloo backend,
world_size 2, 3, 4 and 8: block partition, early halo exchange of boundary-frame descriptors (boundary frames extracted as
their own launch group), gather of COMPACTED match records to rank 0 only.
The compute functions are injected (here: the CPU oracle on small inputs; in production the HIP pipeline), so the
test checks exactly the distributed logic: the sharded result must equal the single-process result pair for pair."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_inputs(n_frames, K=96, d=128):
    desc = np.stack([synth.unit_descriptors(300 + i, K, d, dup=6) for i in range(n_frames)])
    # make consecutive frames related so that matches survive the thresholds
    for i in range(1, n_frames):
        rng = np.random.Generator(np.random.PCG64(i))
        mix = desc[i - 1][rng.permutation(K)] + 0.2 * desc[i]
        desc[i] = (mix / np.linalg.norm(mix, axis=1, keepdims=True)).astype(np.float32)
    sc = np.stack([np.random.Generator(np.random.PCG64(50 + i)).random(K).astype(np.float32) for i in range(n_frames)])
    inten = np.stack([np.random.Generator(np.random.PCG64(90 + i)).random(K).astype(np.float32) for i in range(n_frames)])
    return desc, sc, inten


def _extract(tokens, images):
    # "tokens" carries (desc | scores | intensity) packed per frame: extraction itself is not under test here.
    # Like SequencePipeline.extract, the intensity exists only when pixels were passed.
    t = tokens
    ex = dict(descriptors=t[:, :, :128].contiguous(), scores=t[:, :, 128].contiguous())
    if images is not None:
        ex["intensity"] = t[:, :, 129].contiguous()
    return ex


def _match(desc, sc, inten, sp):
    from oracle import ora
    n, K = desc.shape[0], desc.shape[1]
    p = n - sp
    mt = torch.zeros((max(p, 0), K, 2), dtype=torch.int64)
    q = torch.zeros((max(p, 0), K), dtype=torch.float32)
    cnt = torch.zeros((max(p, 0),), dtype=torch.int32)
    for i in range(p):
        m, qq = ora.match_with_quality(desc[i].numpy(), desc[i + sp].numpy(), sc[i].numpy(), sc[i + sp].numpy(), 0.3, 0.3, 0.5,
                                       None if inten is None else inten[i].numpy(), None if inten is None else inten[i + sp].numpy(), 0.1)
        mt[i, :len(m)] = torch.from_numpy(m)
        q[i, :len(m)] = torch.from_numpy(qq)
        cnt[i] = len(m)
    return dict(matches=mt, quality=q, match_count=cnt)


def _pack(desc, sc, inten):
    return torch.from_numpy(np.concatenate([desc, sc[..., None], inten[..., None]], axis=-1))


def _worker(rank, world, port, n_frames, spacing, q, in_place=True, gather="records", halo="early", with_images=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
        from sslam_amd.shard import ShardedSequenceRunner, pipeline_from_rank0, shard_bounds
        desc, sc, inten = _make_inputs(n_frames)
        lo, hi = shard_bounds(n_frames, world, rank)
        # only rank 0 holds the state dicts; the others receive the PACKED selector / refiner buffers
        ssd, rsd = synth.selector_state(0, hidden=128), synth.refiner_state(0, n_blocks=1)
        pipe = pipeline_from_rank0(ExtractorConfig(), ssd if rank == 0 else None, rsd if rank == 0 else None, "cpu")
        want = SequencePipeline(ExtractorConfig(), ssd, rsd, device="cpu")
        assert pipe.selector.hidden == 128 and pipe.refiner.n_blocks == 1
        got_w, want_w = pipe.weight_tensors(), want.weight_tensors()
        assert len(got_w) == len(want_w) == 9
        for a, b in zip(got_w, want_w):
            assert a.shape == b.shape and torch.equal(a.view(torch.uint8), b.view(torch.uint8))
        calls = []

        def extract(tokens, images, out=None):
            calls.append(tokens.shape[0])
            ex = _extract(tokens, images)
            if out is None:
                return ex
            assert in_place, "out= passed to an extract function that does not take it"
            for k, v in ex.items():
                out[k][:] = v
            return out

        def extract_plain(tokens, images):
            return extract(tokens, images)

        K = desc.shape[1]

        def alloc(rows):
            # like SequencePipeline.alloc_extract(rows, True): it does not know whether pixels will be passed.  NaN poison: an
            # 'intensity' buffer nothing wrote must never reach the exchange or the matcher (shard.py)
            return dict(descriptors=torch.empty((rows, K, 128)), scores=torch.empty((rows, K)), intensity=torch.full((rows, K), float("nan")))

        runner = ShardedSequenceRunner(extract if in_place else extract_plain, _match, spacing=spacing,
                                       alloc_fn=alloc if halo == "late" else None, halo=halo)
        # no frame offset from the caller; "padded+sizes": the blocks' frame counts are passed (no size exchange)
        kw = dict(gather="records") if gather == "records" else dict(gather="padded")
        if gather == "padded+sizes":
            kw["frames_per_rank"] = [b - a for a, b in (shard_bounds(n_frames, world, r) for r in range(world))]
        pixels = torch.zeros((hi - lo, 1), dtype=torch.uint8) if with_images else None      # stands for this block's frames
        out = runner.run(_pack(desc[lo:hi], sc[lo:hi], inten[lo:hi]), pixels, **kw)
        assert out["descriptors"].shape[0] == hi - lo and torch.equal(out["descriptors"], _extract(_pack(desc[lo:hi], sc[lo:hi], inten[lo:hi]), None)["descriptors"])
        assert ("intensity" in out) == with_images
        if halo == "late" and in_place:
            assert calls == [hi - lo], calls           # the block is ONE launch group; the exchange follows it
        else:
            # the boundary frames go first, as their own group, whenever the block is longer than the halo
            assert calls == ([spacing, hi - lo - spacing] if hi - lo > spacing else [hi - lo]), calls
        if rank == 0:
            rec = (out["records"].numpy(), out["records_per_rank"]) if gather == "records" else (None, None)
            assert ("records" in out) == (gather == "records")
            q.put((out["all_match_count"].numpy(), out["all_matches"].numpy(), out["all_quality"].numpy(), out["pairs_per_rank"]) + rec)
        else:
            assert "all_matches" not in out and "records" not in out      # payload goes to rank 0 only
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,spacing,in_place,gather,halo,with_images",
                         [(2, 9, 1, True, "records", "early", True), (3, 11, 2, True, "padded+sizes", "late", True),
                          (2, 4, 2, True, "padded", "late", True), (4, 14, 1, True, "padded+sizes", "late", True),
                          (3, 10, 1, False, "records", "late", True), (3, 11, 2, True, "records", "early", True),
                          (2, 9, 1, False, "padded", "early", True), (4, 14, 1, True, "padded+sizes", "early", True),
                          # tokens only (no pixels): no intensity anywhere - the late path must drop alloc_fn's buffer for it
                          (3, 11, 1, True, "padded+sizes", "late", False), (2, 7, 2, True, "records", "early", False),
                          # BASELINE configs[4] arithmetic: 8 ranks, a frame count 8 does not divide, spacing 1, the defaults of
                          # bench.py --gpus 8 (late halo, padded gather with the ranks' frame counts: no size exchange)
                          (8, 43, 1, True, "padded+sizes", "late", True),
                          # 8 ranks, spacing 5: a halo of five frames per boundary (blocks of 6 and 7 frames)
                          (8, 53, 5, True, "padded+sizes", "late", True), (8, 53, 5, True, "records", "early", True)])
def test_sharded_equals_single_process(world, n_frames, spacing, in_place, gather, halo, with_images):
    from sslam_amd.shard import shard_bounds
    # partition covers every frame exactly once, contiguously
    edges = [shard_bounds(n_frames, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n_frames and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, spacing, q, in_place, gather, halo, with_images)) for r in range(world)]
    for p in procs:
        p.start()
    cnt, mt, qual, per_rank, records, rec_per_rank = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, sc, inten = _make_inputs(n_frames)
    ref = _match(torch.from_numpy(desc), torch.from_numpy(sc), torch.from_numpy(inten) if with_images else None, spacing)
    assert sum(per_rank) == n_frames - spacing                     # every pair exactly once, boundary pairs included
    assert np.array_equal(cnt, ref["match_count"].numpy())
    assert np.array_equal(mt, ref["matches"].numpy())
    assert np.array_equal(qual.view(np.uint32), ref["quality"].numpy().view(np.uint32))
    assert cnt.sum() > 0
    if gather != "records":
        return          # padded form: the arrays above ARE what travelled, received in place
    # the wire format: one 16-byte record per match, pairs ascending, idx1 ascending inside a pair - nothing padded travels
    assert records.shape == (int(cnt.sum()), 4) and records.dtype == np.int32 and sum(rec_per_rank) == records.shape[0]
    assert np.array_equal(records[:, 0], np.repeat(np.arange(n_frames - spacing), cnt))
    want = np.concatenate([ref["matches"].numpy()[i, :c] for i, c in enumerate(cnt)])
    assert np.array_equal(records[:, 1:3], want)
    assert records.nbytes == 16 * int(cnt.sum()) < mt.nbytes + qual.nbytes      # padded arrays: 20 bytes per SLOT


def test_expand_records_rejects_malformed_records():
    """Records that compact_records cannot have produced (pair numbers out of order / out of range, more than K per pair)
    raise a Python error instead of an out-of-bounds indexed write (shard.py)."""
    from sslam_amd.shard import compact_records, expand_records
    K = 4
    mt = torch.zeros((3, K, 2), dtype=torch.int64)
    q = torch.rand((3, K))
    cnt = torch.tensor([2, 0, 3], dtype=torch.int32)
    for i in range(3):
        mt[i, :, 0] = torch.arange(K) + 10 * i
        mt[i, :, 1] = torch.arange(K) + 100 * i
    rec, nv = compact_records(mt, q, cnt)
    rec = rec[: int(nv)]
    back = expand_records(rec, 3, K)
    assert torch.equal(back["all_match_count"], cnt)
    for i in range(3):
        c = int(cnt[i])
        assert torch.equal(back["all_matches"][i, :c], mt[i, :c]) and torch.equal(back["all_quality"][i, :c], q[i, :c])
    shuffled = rec.flip(0)
    with pytest.raises(ValueError):
        expand_records(shuffled, 3, K)
    with pytest.raises(ValueError):
        expand_records(rec, 2, K)                          # pair index 2 out of range
    dup = torch.cat([rec[:2], rec[:2], rec[:2]])           # 6 records for pair 0 > K... sorted, in range
    with pytest.raises(ValueError):
        expand_records(dup, 3, K)


def test_bench_launch_plan():
    """bench.py --gpus N decides, before touching a GPU, between running as one rank, starting the N ranks itself, and
    failing: it never reports a smaller job as N GPUs."""
    import bench
    assert bench.launch_plan(1, {}, 1)[0] == "run"
    assert bench.launch_plan(1, {}, 0)[0] == "run"                      # the GPU assertion comes later, with its own message
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, 8)[0] == "run"     # under torch.distributed.run
    assert bench.launch_plan(2, {}, 8)[0] == "spawn"
    assert bench.launch_plan(8, {}, 8)[0] == "spawn"
    act, msg = bench.launch_plan(2, {}, 1)                              # a 1-GPU box asked for 2
    assert act == "error" and "only 1 GPU" in msg
    assert bench.launch_plan(4, {"WORLD_SIZE": "2"}, 8)[0] == "error"
    assert bench.launch_plan(2, {"WORLD_SIZE": "1"}, 8)[0] == "error"   # the old escape hatch (world == 1) is gone
    assert bench.launch_plan(0, {}, 8)[0] == "error"


def test_bench_exits_nonzero_when_gpus_are_missing():
    """`python bench.py --gpus 2` where fewer than 2 GPUs are visible (here: none) exits non-zero with a message and
    prints no JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr and "{" not in r.stdout

"""CPU test of the frame-sharded N > 1 path (sslam_amd/shard.py) with torch.distributed on the gloo backend,
world_size 2, 3 and 4: block partition, early halo exchange of boundary-frame descriptors (boundary frames extracted as
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
    # "tokens" carries (desc | scores | intensity) packed per frame: extraction itself is not under test here
    K = 96
    t = tokens
    return dict(descriptors=t[:, :, :128].contiguous(), scores=t[:, :, 128].contiguous(), intensity=t[:, :, 129].contiguous())


def _match(desc, sc, inten, sp):
    from oracle import ora
    n, K = desc.shape[0], desc.shape[1]
    p = n - sp
    mt = torch.zeros((max(p, 0), K, 2), dtype=torch.int64)
    q = torch.zeros((max(p, 0), K), dtype=torch.float32)
    cnt = torch.zeros((max(p, 0),), dtype=torch.int32)
    for i in range(p):
        m, qq = ora.match_with_quality(desc[i].numpy(), desc[i + sp].numpy(), sc[i].numpy(), sc[i + sp].numpy(), 0.3, 0.3, 0.5,
                                       inten[i].numpy(), inten[i + sp].numpy(), 0.1)
        mt[i, :len(m)] = torch.from_numpy(m)
        q[i, :len(m)] = torch.from_numpy(qq)
        cnt[i] = len(m)
    return dict(matches=mt, quality=q, match_count=cnt)


def _pack(desc, sc, inten):
    return torch.from_numpy(np.concatenate([desc, sc[..., None], inten[..., None]], axis=-1))


def _worker(rank, world, port, n_frames, spacing, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sslam_amd.shard import ShardedSequenceRunner, broadcast_weights, shard_bounds
        desc, sc, inten = _make_inputs(n_frames)
        lo, hi = shard_bounds(n_frames, world, rank)
        w = torch.full((7,), float(rank))
        broadcast_weights([w], src=0)
        assert float(w.sum()) == 0.0
        calls = []

        def extract(tokens, images):
            calls.append(tokens.shape[0])
            return _extract(tokens, images)

        runner = ShardedSequenceRunner(extract, _match, spacing=spacing)
        out = runner.run(_pack(desc[lo:hi], sc[lo:hi], inten[lo:hi]), first_frame=lo)
        # the boundary frames go first, as their own group, whenever the block is longer than the halo
        assert calls == ([spacing, hi - lo - spacing] if hi - lo > spacing else [hi - lo]), calls
        if rank == 0:
            q.put((out["all_match_count"].numpy(), out["all_matches"].numpy(), out["all_quality"].numpy(), out["pairs_per_rank"],
                   out["records"].numpy(), out["records_per_rank"]))
        else:
            assert "all_matches" not in out and "records" not in out      # payload goes to rank 0 only
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,spacing", [(2, 9, 1), (3, 11, 2), (2, 4, 2), (4, 14, 1)])
def test_sharded_equals_single_process(world, n_frames, spacing):
    from sslam_amd.shard import shard_bounds
    # partition covers every frame exactly once, contiguously
    edges = [shard_bounds(n_frames, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n_frames and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, spacing, q)) for r in range(world)]
    for p in procs:
        p.start()
    cnt, mt, qual, per_rank, records, rec_per_rank = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, sc, inten = _make_inputs(n_frames)
    ref = _match(torch.from_numpy(desc), torch.from_numpy(sc), torch.from_numpy(inten), spacing)
    assert sum(per_rank) == n_frames - spacing                     # every pair exactly once, boundary pairs included
    assert np.array_equal(cnt, ref["match_count"].numpy())
    assert np.array_equal(mt, ref["matches"].numpy())
    assert np.array_equal(qual.view(np.uint32), ref["quality"].numpy().view(np.uint32))
    assert cnt.sum() > 0
    # the wire format: one 16-byte record per match, pairs ascending, idx1 ascending inside a pair - nothing padded travels
    assert records.shape == (int(cnt.sum()), 4) and records.dtype == np.int32 and sum(rec_per_rank) == records.shape[0]
    assert np.array_equal(records[:, 0], np.repeat(np.arange(n_frames - spacing), cnt))
    want = np.concatenate([ref["matches"].numpy()[i, :c] for i, c in enumerate(cnt)])
    assert np.array_equal(records[:, 1:3], want)
    assert records.nbytes == 16 * int(cnt.sum()) < mt.nbytes + qual.nbytes      # padded arrays: 20 bytes per SLOT

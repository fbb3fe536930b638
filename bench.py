#!/usr/bin/env python3
"""bench.py - frames/s of the extraction + matching hot path on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one frame sequence per GPU: A0 (Pillow-exact preprocessing), A2 (token
BatchNorm), A3 (saliency CNN), A4/A5 (NMS + top-k), A6+A7 (gather + descriptor MLP), A9 (intensity), M1 for every
consecutive pair, and for N > 1 the halo exchange + gather of match records (sslam_amd/shard.py).
Workload at N = 1: BASELINE.json configs[1] restated synthetically: 613 frames of 640x480 RGB +
ViT-S/16 token grids (28x28), 500 keypoints, all-fp32 exact mode.  The third-party ViT (A1) is NOT inside the timed
region: tokens are an input.  Inputs are resident in HBM before the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# dmabuf IPC: RCCL / device-buffer sharing between the rank processes of one node fails on this driver with the legacy IPC mode
# (hipIpcGetMemHandle: invalid argument); must be in the environment before the HIP runtime loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

WORKLOADS = {
    # name: (frames per GPU, image h, w, input_size, keypoints)
    "fr1_desk_613": (613, 480, 640, 448, 500),
    "fr1_xyz_50": (50, 480, 640, 448, 500),
    "fr2_desk_1024kp": (2965, 480, 640, 640, 1024),          # configs[2] at its full length (three launch groups of <= 1024 frames)
    "synthetic_2048kp": (128, 960, 1280, 960, 2048),
    # the multi-GPU configs of BASELINE.json at their own per-GPU share (use with --gpus 4 / --gpus 8):
    "fr3_long_office_4gpu": (647, 480, 640, 448, 500),        # configs[3]: 2 585 frames over 4 GPUs, RCCL gather of match pairs
    "synthetic_2048kp_8gpu": (512, 960, 1280, 960, 2048),     # configs[4]: 4 096 frames of 1280 x 960 @ 2 048 keypoints over 8 GPUs
}
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
METRIC = "frames/sec extract+match on TUM 640x480; match-index bit-exact vs CPU ref"


def synth_sequence(n_total, lo, hi, h, w, grid, device, seed):
    """Frames [lo, hi) of ONE device-side synthetic sequence of n_total frames: smooth-plus-texture uint8 frames
    related by small shifts, and N(0.5, 3^2) token fields sliding over a larger field (+ per-frame noise) so that real mutual
    matches exist.  The canvas, the token field and the random walk depend on `seed` only and the per-frame noise on
    (seed, frame index), so every rank of a sharded run generates ITS block of the same sequence: the pairs that straddle a
    shard boundary are consecutive frames like any other pair."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    pad, tp, n = 32, 6, hi - lo
    base = torch.randn((1, 3, h + 2 * pad, w + 2 * pad), generator=g, device=device)
    low = base
    for _ in range(3):
        low = torch.nn.functional.avg_pool2d(low, 25, stride=1, padding=12)
    low = low / low.std()
    canvas = 128.0 + 60.0 * low + 25.0 * torch.randn(base.shape, generator=g, device=device)
    field = torch.randn((grid + 2 * tp, grid + 2 * tp, 384), generator=g, device=device) * 3.0 + 0.5
    steps, tsteps = rng.integers(-4, 5, size=(n_total, 2)), rng.integers(-1, 2, size=(n_total, 2))
    imgs = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    toks = torch.empty((n, 5 + grid * grid, 384), dtype=torch.float32, device=device)
    ox = oy = pad
    tx = ty = tp
    gf = torch.Generator(device=device)
    for i in range(hi):
        ox = int(np.clip(ox + steps[i, 0], 0, 2 * pad))
        oy = int(np.clip(oy + steps[i, 1], 0, 2 * pad))
        tx = int(np.clip(tx + tsteps[i, 0], 0, 2 * tp))
        ty = int(np.clip(ty + tsteps[i, 1], 0, 2 * tp))
        if i < lo:
            continue
        gf.manual_seed(seed * 1000003 + i)
        win = canvas[0, :, oy:oy + h, ox:ox + w] + torch.randint(-6, 7, (3, h, w), generator=gf, device=device)
        imgs[i - lo] = win.clamp(0, 255).round().permute(1, 2, 0).to(torch.uint8)
        toks[i - lo, :5] = torch.randn((5, 384), generator=gf, device=device) * 3.0 + 0.5
        toks[i - lo, 5:] = (field[ty:ty + grid, tx:tx + grid] + 0.35 * torch.randn((grid, grid, 384), generator=gf, device=device)).reshape(-1, 384)
    return imgs, toks


def cpu_baseline(imgs, toks, ssd, rsd, size, K, cfg, budget_s=12.0):
    """The CPU oracle (oracle/sslam_oracle.c, a port of the reference's algorithm) timed on this host's cores on a
    bounded sample of the same workload: extract every frame once + one match per consecutive pair (oracle_check.oracle_block
    with A0 inside - the same function the parity gate runs).  Returns (the JSON object, the outputs of one timed run): the gate
    compares those outputs with the GPU pass too, so what is timed is what is checked."""
    from oracle import ora
    from oracle_check import oracle_block
    nthr = ora.host_threads()       # affinity mask capped by the cgroup CPU quota
    ora.set_num_threads(nthr)
    kept = {}

    def run(n):
        t0 = time.perf_counter()
        o = oracle_block(imgs[:n], toks[:n], ssd, rsd, size, K, cfg, with_a0=True)
        t = time.perf_counter() - t0
        if n == len(imgs):
            kept["out"] = o
        return t

    run(4)                                       # warm-up: thread pool, page faults
    nmax = len(imgs)
    t1 = run(nmax)
    reps = int(max(1, min(40, round(budget_s / max(t1, 1e-3)))))
    ts = [t1] + [run(nmax) for _ in range(reps - 1)]
    t = float(np.mean(ts))
    # the same port on ONE thread (SURVEY 8d asks for both), on a shorter sample of the same frames
    ora.set_num_threads(1)
    n1 = min(nmax, 16)
    run(2)
    t_1 = run(n1)
    ora.set_num_threads(nthr)
    return dict(value=round(nmax / t, 2), unit="frames/s", cores=nthr, kind="port", value_1core=round(n1 / t_1, 2),
                median=round(nmax / float(np.median(ts)), 2),
                sample=f"{nmax} frames of the same workload (each extracted once + {nmax - 1} consecutive-pair matches), "
                       f"repeated {len(ts)}x = {sum(ts):.1f} s of CPU work; oracle/sslam_oracle.c (AVX2+FMA, OpenMP) on "
                       f"{nthr} threads; mean of repeats"), kept["out"]


def measured_mfma_peak():
    """The matrix pipe's own rate on THIS box (tools/microbench/mfma_peak.hip: every wave issues independent
    v_mfma_f32_32x32x2_f32 chains, no memory traffic), as ~6 ms bursts - the duty cycle of the conv inside the pass.
    It sits below the nominal 157.3 TFLOP/s (2.4 GHz): under matrix load the clock is about 2.27 GHz."""
    import ctypes
    so = os.path.join(ROOT, "tools", "microbench", "libmfma_peak.so")
    if not os.path.exists(so):
        return None
    L = ctypes.CDLL(so)
    L.mfma_peak_tflops.restype = ctypes.c_double
    L.mfma_peak_tflops.argtypes = [ctypes.c_int, ctypes.c_int]
    L.mfma_peak_tflops(0, 200)                                   # module load
    burst = [L.mfma_peak_tflops(0, 3080) for _ in range(3)]
    return {"f32_burst_tflops": round(float(np.median(burst)), 1),
            "source": "tools/microbench/mfma_peak.hip (pure MFMA loop, 2 waves/SIMD x 4 independent accumulators, ~6 ms)"}


def path_roofline(fps, grid, K, hidden, h, w):
    cells = grid * grid
    flop = cells * hidden * 3456 * 2 + cells * hidden * 2 + K * 1572864 + K * 384 * 8 + K * K * 128 * 2
    bytes_io = h * w * 3 + (5 + cells) * 384 * 4 + K * (8 + 4 + 4 + 512) + K * 20
    return {"flop_per_frame": flop, "mfma_tflops": round(fps * flop / 1e12, 2), "frac_of_fp32_matrix_peak": round(fps * flop / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
            "compulsory_bytes_per_frame": bytes_io, "hbm_tb_s": round(fps * bytes_io / 1e12, 3), "frac_of_hbm_8tb_s": round(fps * bytes_io / 8e12, 4)}


def index_parity_vs_torch(grid):
    """What is known about index parity against TORCH itself (the reference's own arithmetic) at this grid, read from the
    committed measurement (tools/order_swap_rate.py on the reference goldens) - the bench line's bit_exact is against the oracle."""
    import glob
    src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_order_swap_rate.json")))
    if grid <= 28:
        return {"index_parity_vs_torch": {"level": "exact: keypoint indices and match pairs identical to the reference's on every G = 28 golden",
                                          "source": "tests/golden/e2e.npz, selector.npz, select_wide.npz, match_wide.npz (tests/test_oracle_golden.py)"}}
    level = {"level": "set-level: keypoint SET and matches as (cell, cell) pairs identical to the reference's; keypoint ORDER (hence raw match "
                      "indices) only up to swaps between saliencies the reference itself holds closer than its conv's summation-order noise",
             "source": "tests/e2e_check.py on tests/golden/e2e_g40.npz / e2e_g60.npz / order_g60.npz"}
    if src:
        try:
            d = json.load(open(src[-1]))
            key = "e2e_g40" if grid == 40 else "e2e_g60"
            level["measured"] = {"fixture": key, **d[key]["summary"], "pairs": d[key]["pairs"],
                                 "reference_vs_itself": {k: v for k, v in d.get("order_g60", {}).get("summary", {}).items() if k.startswith("reference_vs_itself")}}
            level["source"] += "; profiles/" + os.path.basename(src[-1])
        except Exception:
            pass
    return {"index_parity_vs_torch": level,
            "note": "bit_exact is against the CPU oracle; against torch itself see index_parity_vs_torch (set-level at G >= 40)"}


def launch_plan(gpus: int, env: dict, device_count: int, shared_gpu: bool = False):
    """What `bench.py --gpus N` does, decided BEFORE anything touches a GPU.  Returns (action, message):
      "run"   - this process is one rank (WORLD_SIZE == N, the torch.distributed.run contract), or N == 1;
      "spawn" - N > 1 and no WORLD_SIZE: start N fresh rank processes with torch.distributed.run and pass rank 0's line through;
      "error" - WORLD_SIZE != N, or fewer than N GPUs visible: exit non-zero instead of reporting a smaller job as N GPUs.
    shared_gpu (--rehearse-shared-gpu): N ranks may share fewer GPUs - a functional rehearsal over gloo that reports no value."""
    if gpus < 1:
        return "error", f"--gpus {gpus}: need at least one GPU"
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            return "error", f"--gpus {gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {gpus} (or drop WORLD_SIZE and let bench.py start the ranks)"
        return "run", ""
    if gpus == 1:
        return "run", ""
    if device_count < gpus and not (shared_gpu and device_count >= 1):
        return "error", f"--gpus {gpus} but only {device_count} GPU(s) visible on this node: refusing to report a smaller job as {gpus} GPUs"
    return "spawn", ""


def spawn_ranks(gpus: int, argv: list) -> int:
    """Start `gpus` rank processes of this script (one per GPU, RCCL rendezvous on 127.0.0.1) from a parent that has not
    initialised the GPU, wait for them, and return their exit code; rank 0's JSON line goes to our stdout unchanged."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL / tensor sharing across processes on this driver)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fr1_desk_613", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain", type=float, default=10.0, help="N = 1: seconds of back-to-back steps for the `sustained` rate (0: skip)")
    ap.add_argument("--frames", type=int, default=0, help="override frames per GPU (profiling runs)")
    ap.add_argument("--parity-frames", type=int, default=-1,
                    help="frames per rank compared with the CPU oracle after the timed region (-1, the default: the whole block; "
                         "a smaller count: that many frames in three blocks from both ends and the middle)")
    ap.add_argument("--no-vit", action="store_true", help="skip the additional end-to-end leg that includes the HIP ViT (A1)")
    ap.add_argument("--no-vit-fp32", action="store_true", help="skip the fp32-operand HIP ViT leg (reference numerics for A1) of the ViT-inside pass")
    ap.add_argument("--no-bf16", action="store_true", help="skip the additional bf16 throughput-mode leg (BASELINE configs[1])")
    ap.add_argument("--no-upload", action="store_true", help="skip the host-resident-frames leg (overlapped H2D feed)")
    ap.add_argument("--tum-root", default=None, help="a TUM RGB-D sequence directory (rgb/*.png ...) for the directory -> matches leg; "
                    "absent or not given: a synthetic directory of the workload's first frames is written to a temp dir")
    ap.add_argument("--tum-frames", type=int, default=128, help="frames of the directory leg (synthetic directory size / max_frames)")
    ap.add_argument("--no-directory", action="store_true", help="skip the directory -> matches leg")
    ap.add_argument("--halo", choices=("late", "early"), default="late",
                    help="N > 1: extract the block as one launch group and exchange the boundary frames afterwards (one small transfer "
                         "exposed), or extract the boundary frames first as their own group (transfer hidden, +0.26 ms of small launches)")
    ap.add_argument("--gather", choices=("padded", "records"), default="padded",
                    help="N > 1: how the matches reach rank 0 - fixed-capacity arrays received in place (no device work, no host "
                         "synchronisation) or compacted 16-byte records (half the bytes, a size exchange and an expansion on rank 0)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="functional rehearsal of the N-rank path on fewer than N GPUs: ranks share the visible GPU(s), torch.distributed "
                         "runs on gloo with device buffers staged through host memory, rank 0 checks sharded == single-process; "
                         "prints a line with value null (NOT a measurement)")
    ap.add_argument("--test-corrupt-gathered", action="store_true",
                    help="TEST ONLY (N > 1): flip one match index of the first cross-rank pair in the gathered result before the parity "
                         "gate - the run must then report no value and exit 1")
    args = ap.parse_args()

    # nothing above this line and nothing in launch_plan touches a GPU (torch.cuda.device_count() only counts devices)
    action, msg = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), args.rehearse_shared_gpu)
    if action == "error":
        print(f"bench.py: {msg}", file=sys.stderr)
        sys.exit(2)
    if action == "spawn":
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # stdout carries rank 0's ONE JSON line and nothing else: whatever a library writes to file descriptor 1 from here on (gloo's
    # connection notes, a runtime's warnings) is sent to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    rehearsal = bool(args.rehearse_shared_gpu and world > 1)
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")           # RCCL refuses two ranks on one GPU: the rehearsal stages through host memory
        else:
            import datetime
            # a rendezvous or collective that does not complete fails the run after 5 minutes instead of hanging it
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks", file=sys.stderr)
            sys.exit(2)

    import synth
    from sslam_amd import lib
    from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
    from sslam_amd.shard import ShardedSequenceRunner, pipeline_from_rank0, shard_bounds

    n, h, w, size, K = WORKLOADS[args.workload]
    if args.frames:
        n = args.frames
    grid = size // 16
    cfg = ExtractorConfig(input_size=size, num_keypoints=K)
    ssd, rsd = synth.selector_state(0), synth.refiner_state(0)
    if world > 1:
        # rank 0 alone holds the checkpoint and packs it; the other ranks receive the packed buffers over RCCL
        pipe = pipeline_from_rank0(cfg, ssd if rank == 0 else None, rsd if rank == 0 else None, dev)
    else:
        pipe = SequencePipeline(cfg, ssd, rsd, device=dev)
    # one sequence of n * world frames, cut into contiguous blocks (shard_bounds); this rank generates its own block
    lo, hi = shard_bounds(n * world, world, rank)
    imgs, toks = synth_sequence(n * world, lo, hi, h, w, grid, dev, seed=1234)
    runner = ShardedSequenceRunner(pipe.extract, pipe.match, spacing=cfg.spacing, alloc_fn=lambda rows: pipe.alloc_extract(rows, True),
                                   halo=args.halo)
    ranges = [[lo, hi]]
    if world > 1:
        rg = [None] * world
        dist.all_gather_object(rg, [lo, hi])
        ranges = rg

    # per-stage HIP events on the launch stream (torch's current stream is the one handed to the C ABI)
    ev = {}

    def timed(name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        ev.setdefault(name, []).append((a, b))
        return r

    def extract_staged(t, im, out=None):
        """pipe.extract stage by stage (same calls, same buffers, the same launch groups of pipe.launch_group() frames), each
        bracketed by HIP events on the launch stream; a stage's time per step is the sum over its launch groups"""
        s = pipe.selector
        o = out if out is not None else pipe.alloc_extract(t.shape[0], True)
        step_ = pipe.launch_group()
        for a_ in range(0, t.shape[0], step_):
            tt, ii = t[a_:a_ + step_], im[a_:a_ + step_]
            og = {k_: v_[a_:a_ + step_] for k_, v_ in o.items()}
            vit_in = timed("A0_preprocess", lambda: pipe.preprocess(ii, reuse=True))
            feat = timed("A2_bn_tokens", lambda: pipe.features(tt, reuse=True))
            ws = pipe.workspace(tt.shape[0], 0)
            timed("A3_selector_saliency", lambda: lib.selector_saliency(feat, s.w1p, s.b1, s.w2, s.b2, s.hidden, out=og["saliency"], workspace=ws))
            timed("A45_select_keypoints",
                  lambda: lib.select_keypoints(og["saliency"], cfg.num_keypoints, cfg.nms_radius, cfg.min_score_percentile,
                                               out=(og["keypoints_patch"], og["scores"], og["idx"], og["keypoints_pixel"], og["status"])))
            timed("A67_gather_refine", lambda: lib.gather_refine(feat, og["keypoints_patch"], pipe.refiner.packed, pipe.refiner.n_blocks, out=og["descriptors"]))
            th, tv = pipe.tables.get(h, w, size, True)
            timed("A9_intensity", lambda: lib.keypoint_intensity(ii, size, th, tv, og["keypoints_pixel"], out=og["intensity"]))
            del vit_in, feat
        return o

    def match_staged(desc, sc, inten, sp):
        return timed("M1_match", lambda: pipe.match(desc, sc, inten, sp))

    runner.extract_fn, runner.match_fn = extract_staged, match_staged

    # every rank's block size is known from shard_bounds: the padded gather then needs no size exchange (no host synchronisation)
    frames_per_rank = [b - a for a, b in (shard_bounds(n * world, world, r) for r in range(world))]

    def step():
        return runner.run(toks, imgs, gather=args.gather, frames_per_rank=frames_per_rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    ev.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # additional leg (N = 1): the SUSTAINED rate.  `value` is `steps` passes of ~11 ms, i.e. a burst on a chip whose clock is
    # power-managed; here the same step loops for `--sustain` seconds (default 10) and the rate - and the dominant kernel's
    # TFLOP/s from its HIP events - is taken over the LAST half of that time, when clocks and temperature have settled.
    sustained = None
    if world == 1 and args.sustain > 0:
        keep = dict(ev)                  # the timed region's stage events
        ev.clear()
        marks, a3_seen = [], []           # event at the end of each step; A3 launches recorded up to it (launch groups per step)
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        k_ = 0
        while True:
            out = step()
            e_ = torch.cuda.Event(enable_timing=True)
            e_.record()
            marks.append(e_)
            a3_seen.append(len(ev["A3_selector_saliency"]))
            k_ += 1
            if k_ % 4 == 0:
                marks[-3].synchronize()    # keep the launch queue a few steps deep, not seconds deep
                if time.perf_counter() - ts0 >= args.sustain:
                    break
        torch.cuda.synchronize()
        total_s = time.perf_counter() - ts0
        ms_from_first = [0.0] + [marks[0].elapsed_time(m_) for m_ in marks[1:]]
        span = ms_from_first[-1]
        i0 = next(i for i, t_ in enumerate(ms_from_first) if t_ >= span / 2)        # first step that ends in the last half
        win_ms = span - ms_from_first[i0]
        nwin = len(marks) - 1 - i0
        a3 = float(np.sum([a.elapsed_time(b) for a, b in ev["A3_selector_saliency"][a3_seen[i0]:]])) / max(nwin, 1)   # ms per step
        cells_s = grid * grid
        conv_flop_s = n * cells_s * pipe.selector.hidden * (9 * 384) * 2 + n * cells_s * pipe.selector.hidden * 2
        sustained = {"seconds": round(total_s, 2), "steps": len(marks), "window_s": round(win_ms * 1e-3, 2), "window_steps": nwin,
                     "value": round(n * nwin / (win_ms * 1e-3), 2), "unit": "frames/s",
                     "a3_ms_per_step": round(a3, 4), "a3_tflops": round(conv_flop_s / (a3 * 1e-3) / 1e12, 2),
                     "a3_frac_of_fp32_matrix_peak": round(conv_flop_s / (a3 * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                     "first_half_frames_s": round(n * i0 / (ms_from_first[i0] * 1e-3), 2) if i0 else None,
                     "ratio_to_burst_value": round((n * nwin / (win_ms * 1e-3)) / (n * args.steps / dt), 4)}
        ev.clear()
        ev.update(keep)

    # additional leg (N = 1): the same pass with the tokens computed on the GPU by the HIP ViT-S/16 (A1, random weights
    # of the DINOv3 architecture - pretrained weights are a remote fetch): images -> A0 -> A1 -> A2 .. M1
    vit_leg = None
    if world == 1 and not args.no_vit:
        from sslam_amd.vit import DinoV3ViT
        torch.manual_seed(0)
        pipe_v = SequencePipeline(cfg, ssd, rsd, device=dev, vit=DinoV3ViT().to(dev).eval())
        for _ in range(max(1, args.warmup)):
            pipe_v.run(imgs)
        torch.cuda.synchronize()
        tv = time.perf_counter()
        nv = max(1, min(args.steps, 3))
        for _ in range(nv):
            ov = pipe_v.run(imgs)
        torch.cuda.synchronize()
        dtv = (time.perf_counter() - tv) / nv
        cells_ = grid * grid
        t_ = cells_ + 5
        vit_flop = 12 * (t_ * 384 * 1152 * 2 + 2 * 6 * t_ * t_ * 64 * 2 + t_ * 384 * 384 * 2 + 2 * t_ * 384 * 1536 * 2) + cells_ * 768 * 384 * 2
        # the ViT alone, timed on the launch stream with HIP events: its roofline is the dense bf16 MFMA peak
        tok_buf = pipe_v.tokens_from_images(imgs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nv):
            pipe_v.tokens_from_images(imgs)
        e1.record()
        torch.cuda.synchronize()
        vit_ms = e0.elapsed_time(e1) / nv
        del tok_buf
        # the same pass with the frames starting in (pinned) host memory: chunked H2D on a side stream under the ViT
        vit_up = None
        if not args.no_upload:
            from sslam_amd.harness import run_frames
            pin_v = imgs.cpu().pin_memory()
            run_frames(pipe_v, n, h, w, spacings=(cfg.spacing,), pinned_source=pin_v)
            torch.cuda.synchronize()
            tvu = time.perf_counter()
            for _ in range(nv):
                ouv = run_frames(pipe_v, n, h, w, spacings=(cfg.spacing,), pinned_source=pin_v)
            torch.cuda.synchronize()
            dtvu = (time.perf_counter() - tvu) / nv
            vit_up = {"value": round(n / dtvu, 2), "unit": "frames/s", "ms_per_step": round(dtvu * 1e3, 3),
                      "frac_of_resident": round(dtv / dtvu, 4),
                      "equal_to_resident_pass": bool(torch.equal(ouv[cfg.spacing]["matches"], ov["matches"]) and
                                                     torch.equal(ouv["frames"]["descriptors"], ov["descriptors"]))}
            del ouv, pin_v
        def agreement(oa, ob):
            """keypoint sets per frame and matches as (cell, cell) pairs of pass ob against pass oa (same frames)"""
            ia, ib = oa["idx"].cpu().numpy(), ob["idx"].cpu().numpy()
            same = float(np.mean([np.intersect1d(a_, b_).size / float(np.unique(a_).size) for a_, b_ in zip(ia, ib)]))
            ma, mb = oa["matches"].cpu().numpy(), ob["matches"].cpu().numpy()
            ca, cb = oa["match_count"].cpu().numpy(), ob["match_count"].cpu().numpy()
            hit_ = tot_ = 0
            for p_ in range(n - 1):
                sa = set(zip(ia[p_][ma[p_, :ca[p_], 0]].tolist(), ia[p_ + 1][ma[p_, :ca[p_], 1]].tolist()))
                sb = set(zip(ib[p_][mb[p_, :cb[p_], 0]].tolist(), ib[p_ + 1][mb[p_, :cb[p_], 1]].tolist()))
                hit_ += len(sa & sb)
                tot_ += len(sa)
            return same, hit_, tot_

        # the whole path in its throughput forms: bf16 HIP ViT + bf16 conv stack / descriptor MLP (BASELINE configs[1]'s mode with A1
        # inside), fp32 matcher; agreement against the pass above (bf16 ViT, exact fp32 stages)
        all_bf16 = None
        if not args.no_bf16:
            import dataclasses as _dc
            pipe_vb = SequencePipeline(_dc.replace(cfg, precision="bf16"), ssd, rsd, device=dev, vit=pipe_v.vit_hip.vit)
            pipe_vb.run(imgs)
            torch.cuda.synchronize()
            tvb = time.perf_counter()
            for _ in range(nv):
                ovb = pipe_vb.run(imgs)
            torch.cuda.synchronize()
            dtvb = (time.perf_counter() - tvb) / nv
            kpb, hitb, totb = agreement(ov, ovb)
            all_bf16 = {"value": round(n / dtvb, 2), "unit": "frames/s", "ms_per_step": round(dtvb * 1e3, 3),
                        "what": "images -> A0 -> bf16 HIP ViT -> A2 -> bf16 A3 -> A4/A5 -> bf16 A6+A7 -> A9 -> fp32 M1",
                        "keypoint_set_agreement_vs_exact_stages": round(kpb, 4), "match_agreement_vs_exact_stages": round(hitb / max(totb, 1), 4)}
            del ovb, pipe_vb
        # reference numerics for A1 (the reference's timm ViT is fp32, dino_backbone.py:85): the same pass with the tokens from
        # the eager fp32 torch definition of the same weights (what DinoBackbone(vit_precision="fp32") runs), its rate, and the
        # agreement of the bf16 HIP-ViT pass with it on THIS workload: keypoint sets per frame, matches as (cell, cell) pairs
        fp32_leg = None
        if not args.no_vit_fp32:
            vit_mod = pipe_v.vit_hip.vit
            pipe_32 = SequencePipeline(cfg, ssd, rsd, device=dev, vit=vit_mod, vit_precision="fp32")     # sslam_vit_forward_f32
            pipe_32.run(imgs)                          # warm-up at the full size: workspaces, the fp32 image buffer, the output buffers
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            n32 = max(1, min(args.steps, 2))
            for _ in range(n32):
                o32 = pipe_32.run(imgs)
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t32) / n32
            tok32 = pipe_32.tokens_from_images(imgs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pipe_32.tokens_from_images(imgs, out=tok32)
            e1.record()
            torch.cuda.synchronize()
            vit32_ms = e0.elapsed_time(e1)
            # one frame at a time (the reference's callers): the few-frame forms of the fp32 GEMMs
            x1 = pipe_32.preprocess(imgs[:1])
            for _ in range(3):
                pipe_32.vit_hip.forward_features(x1)
            e0.record()
            for _ in range(20):
                pipe_32.vit_hip.forward_features(x1)
            e1.record()
            torch.cuda.synchronize()
            vit32_b1_ms = e0.elapsed_time(e1) / 20
            # beside it: the eager torch evaluation of the same weights (rocBLAS / hipBLASLt fp32 GEMMs + torch ops) on a sample
            ne = min(n, 128)
            fchunk = max(1, min(32, (64 * 789) // (5 + grid * grid)))
            toke = torch.empty((ne, 5 + grid * grid, 384), dtype=torch.float32, device=dev)

            def tokens_eager():
                with torch.no_grad():
                    for a_ in range(0, ne, fchunk):
                        toke[a_:a_ + fchunk] = vit_mod.forward_features(pipe_v.preprocess(imgs[a_:a_ + fchunk]))

            tokens_eager()
            torch.cuda.synchronize()
            te = time.perf_counter()
            tokens_eager()
            torch.cuda.synchronize()
            eager_ms = (time.perf_counter() - te) * 1e3
            rel_hip_eager = float((tok32[:ne] - toke).norm() / toke.norm())
            tok16 = pipe_v.tokens_from_images(imgs)
            tok_rel = float((tok16 - tok32).norm() / tok32.norm())
            kp_same, hit, tot = agreement(o32, ov)
            tf32 = n * vit_flop / (vit32_ms * 1e-3) / 1e12
            fp32_leg = {"value": round(n / dt32, 2), "unit": "frames/s", "ms_per_step": round(dt32 * 1e3, 3),
                        "what": "images -> A0 -> HIP ViT-S/16 with fp32 operands on the fp32 matrix pipe (sslam_vit_forward_f32: the "
                                "reference's numerics for A1) -> A2..A9 -> M1; same weights as the bf16 HIP-ViT pass beside it",
                        "roofline": {"bound": "mfma", "kernel": "sslam_vit_forward_f32 (per layer: gemm_f32_rows_kernel x 4 + attn_f32_kernel + 2 LayerNorms)",
                                     "achieved": round(tf32, 1), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(tf32 / FP32_MATRIX_PEAK_TFLOPS, 4), "launch_ms": round(vit32_ms, 2)},
                        "eager_torch_fp32": {"vit_frames_s": round(ne / (eager_ms * 1e-3), 1), "frames": ne,
                                             "vit_tflops": round(ne * vit_flop / (eager_ms * 1e-3) / 1e12, 1),
                                             "hip_fp32_vs_eager_tokens_rel_err": float(f"{rel_hip_eager:.3e}")},
                        "vit_batch1_ms": round(vit32_b1_ms, 4),
                        "bf16_vs_fp32_tokens_rel_err": round(tok_rel, 5),
                        "bf16_vs_fp32_keypoint_set_agreement": round(kp_same, 4),
                        "bf16_vs_fp32_match_agreement": round(hit / max(tot, 1), 4), "frames": n, "pairs": n - 1,
                        "matches_fp32": int(tot)}
            del o32, tok32, tok16, toke, pipe_32
        vit_tf = n * vit_flop / (vit_ms * 1e-3) / 1e12
        vit_leg = {"value": round(n / dtv, 2), "unit": "frames/s", "ms_per_step": round(dtv * 1e3, 3),
                   "roofline": {"bound": "mfma", "kernel": "A0 + sslam_vit_forward (52 launches per 82-frame chunk: row-tile GEMMs, attention, fused MLP)",
                                "achieved": round(vit_tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(vit_tf / 2500.0, 4),
                                "launch_ms": round(vit_ms, 3), "flop_per_launch": int(n * vit_flop),
                                "note": "dense bf16 MFMA peak (spec); a pure bf16 MFMA loop on random data sustains ~1.3-1.5 PFLOP/s on this chip (DVFS)"},
                   "what": "images -> A0 -> HIP ViT-S/16 (A1, bf16 MFMA, random DINOv3-architecture weights) -> A2..A9 -> M1",
                   "with_upload": vit_up, "fp32_reference_numerics": fp32_leg, "all_bf16": all_bf16,
                   "vit_gflop_per_frame": round(vit_flop / 1e9, 2), "matches_per_pair": round(float(ov["match_count"].float().mean().item()), 1)}
        del ov

    # additional block (N = 1): single-frame latency, the shape every reference caller has (tools/bench_latency.py)
    latency = None
    if world == 1 and not args.no_vit:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_latency
        latency = bench_latency.measure(pipe_v, toks[:8], imgs[:8], vit_pipe=pipe_v, reps=50)
        del pipe_v

    # additional leg (N = 1): the frames start in HOST memory (pinned uint8, as a decoder would leave them) and are uploaded in
    # chunks on a side stream while the previous chunk is extracted and matched (sslam_amd.harness.run_frames); tokens-in
    # mode like `value`.  Beside it: the raw H2D rate of the same bytes, i.e. the PCIe bound on frames/s.
    upload_leg = None
    if world == 1 and not args.no_upload:
        from sslam_amd.harness import run_frames
        imgs_pin = imgs.cpu().pin_memory()
        probe = torch.empty_like(imgs)
        for _ in range(2):
            probe.copy_(imgs_pin, non_blocking=True)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(3):
            probe.copy_(imgs_pin, non_blocking=True)
        torch.cuda.synchronize()
        h2d_s = (time.perf_counter() - tp) / 3
        del probe
        for _ in range(max(1, args.warmup)):
            ou = run_frames(pipe, n, h, w, spacings=(cfg.spacing,), tokens=toks, pinned_source=imgs_pin, preprocess_too=True)
        torch.cuda.synchronize()
        tu = time.perf_counter()
        for _ in range(args.steps):
            ou = run_frames(pipe, n, h, w, spacings=(cfg.spacing,), tokens=toks, pinned_source=imgs_pin, preprocess_too=True)
        torch.cuda.synchronize()
        dtu = (time.perf_counter() - tu) / args.steps
        same = bool(torch.equal(ou[cfg.spacing]["matches"], out["matches"]) and torch.equal(ou[cfg.spacing]["match_count"], out["match_count"])
                    and torch.equal(ou["frames"]["descriptors"], out["descriptors"]))
        pcie_fps, resident_fps = n / h2d_s, n * args.steps / dt
        upload_leg = {"value": round(n / dtu, 2), "unit": "frames/s", "ms_per_step": round(dtu * 1e3, 3),
                      "what": "pinned host uint8 frames -> chunked H2D on a side stream overlapping extract + match of the previous chunk "
                              "(tokens resident and A0 included, as in `value`)",
                      "h2d_gb_s": round(imgs_pin.numel() / h2d_s / 1e9, 2), "pcie_bound_frames_s": round(pcie_fps, 1),
                      "frac_of_min_resident_pcie": round((n / dtu) / min(pcie_fps, resident_fps), 4),
                      "equal_to_resident_pass": same}
        del ou, imgs_pin

    # additional leg (N = 1): a TUM sequence DIRECTORY -> matches (the reference's whole use case: main -> process_spacing ->
    # extract(path) -> match, visualize_matches_sequence.py:272-357): sorted rgb/*.png -> PIL decode on a host thread pool ->
    # pinned double buffer -> H2D on a side stream -> extract once -> all spacings.  Decode-inclusive, so host-bound.
    dir_leg = None
    if world == 1 and not args.no_directory:
        import shutil
        import tempfile

        from sslam_amd.harness import run_directory
        nd = min(n, args.tum_frames)
        tmp = None
        root = args.tum_root
        if root is None or not os.path.isdir(root):
            tmp = tempfile.mkdtemp(prefix="sslam_tum_")
            root = os.path.join(tmp, "rgbd_dataset_synthetic")
            synth.write_tum_rgb_sequence(root, imgs[:nd].cpu().numpy())
            src = f"synthetic directory of the workload's first {nd} frames (no TUM data on this box)"
        else:
            src = f"{root} (first {nd} frames)"
        try:
            spac = (1, 5, 10, 15, 20)                       # visualize_matches_sequence.py:369
            # tokens: the third-party ViT's output is an input here, as in `value` (real data would need real weights)
            kw = dict(pipe=pipe, tokens_fn=(lambda a, b: toks[a:b]), max_frames=nd)
            od = run_directory(root, "", spac, **kw)
            torch.cuda.synchronize()
            td = time.perf_counter()
            od = run_directory(root, "", spac, **kw)
            torch.cuda.synchronize()
            dtd = time.perf_counter() - td
            nf = len(od["files"])
            same = None
            if tmp is not None:
                same = bool(torch.equal(od[1]["matches"], out["matches"][:nf - 1]) and torch.equal(od[1]["match_count"], out["match_count"][:nf - 1]))
            dir_leg = {"value": round(nf / dtd, 1), "unit": "frames/s", "frames": nf, "source": src, "spacings": list(spac),
                       "pairs_matched": int(sum(od[s_]["match_count"].shape[0] for s_ in spac if s_ in od)),
                       "what": "rgb/*.png -> PIL decode (thread pool) -> pinned double buffer -> H2D side stream -> A0..A9 once per frame -> "
                               "M1 for every spacing; PNG decode on the host cores is the bound",
                       "host_cpus": os.cpu_count(), "equal_to_resident_pass": same}
            del od
        finally:
            if tmp is not None:
                shutil.rmtree(tmp, ignore_errors=True)

    # additional leg (N = 1): BASELINE configs[1] "bf16 conv stack + fp32 matcher" - the same pass with the saliency CNN and
    # the descriptor MLP on bf16 MFMA.  Not index-exact, so it is never `value`: reported with its agreement rates against
    # the exact pass above on the same frames.
    bf16_leg = None
    if world == 1 and not args.no_bf16:
        import dataclasses
        pipe_b = SequencePipeline(dataclasses.replace(cfg, precision="bf16"), ssd, rsd, device=dev)
        for _ in range(max(1, args.warmup)):
            ob = pipe_b.run(imgs, toks, with_preprocess=True)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(args.steps):
            ob = pipe_b.run(imgs, toks, with_preprocess=True)
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - tb) / args.steps
        ie, ib = out["idx"].cpu().numpy(), ob["idx"].cpu().numpy()
        kp_agree = float(np.mean([np.intersect1d(a, b).size / float(K) for a, b in zip(ie, ib)]))
        me, mb = out["matches"].cpu().numpy(), ob["matches"].cpu().numpy()
        ce, cb = out["match_count"].cpu().numpy(), ob["match_count"].cpu().numpy()
        hit = tot = 0
        for p_ in range(n - 1):     # a match = (cell of keypoint 1, cell of keypoint 2): independent of keypoint order
            se = set(zip(ie[p_][me[p_, :ce[p_], 0]].tolist(), ie[p_ + 1][me[p_, :ce[p_], 1]].tolist()))
            sb = set(zip(ib[p_][mb[p_, :cb[p_], 0]].tolist(), ib[p_ + 1][mb[p_, :cb[p_], 1]].tolist()))
            hit += len(se & sb)
            tot += len(se)
        bf16_leg = {"value": round(n / dtb, 2), "unit": "frames/s", "ms_per_step": round(dtb * 1e3, 3), "dtype": "bf16 operands, f32 accumulate",
                    "what": "A0, A2, A3 (bf16 MFMA), A4/A5, A6+A7 (bf16 MFMA), A9, M1 (fp32): BASELINE configs[1]",
                    "keypoint_set_agreement_vs_exact": round(kp_agree, 4),
                    "match_agreement_vs_exact": round(hit / max(tot, 1), 4),
                    "matches_per_pair": round(float(cb.mean()), 1)}
        del pipe_b, ob

    # additional legs (N > 1), reported beside `value` so that a scaling curve shows WHAT bounds it (SURVEY 8e expects the host-side
    # feed and the rank-0 gather, not xGMI): (a) every rank feeds ITS block from pinned host memory (chunked H2D on a side stream
    # under the extraction, harness.run_frames) - all ranks at once, so the host's PCIe lanes and cores are shared as they would
    # be; a per-rank local pass (no halo, no gather); (b) the sharded step with the HIP ViT inside (images -> A0 -> A1 -> ... -> M1,
    # halo + gather included).  Same bracket as `value`: barrier + synchronize on both sides, MAX over ranks.
    multi_legs = None
    if world > 1 and not (args.no_upload and args.no_vit):
        multi_legs = {}

        def rank_max(x):
            tt_ = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            return float(tt_.item())

        nl = max(1, min(args.steps, 3))
        if not args.no_upload:
            from sslam_amd.harness import run_frames
            imgs_pin = imgs.cpu().pin_memory()
            kw_u = dict(spacings=(cfg.spacing,), tokens=toks, pinned_source=imgs_pin, preprocess_too=True)
            run_frames(pipe, n, h, w, **kw_u)
            fence()
            tu = time.perf_counter()
            for _ in range(nl):
                ou = run_frames(pipe, n, h, w, **kw_u)
            fence()
            dtu = rank_max((time.perf_counter() - tu) / nl)
            npl = n - cfg.spacing
            same_u = bool(torch.equal(ou[cfg.spacing]["matches"], out["matches"][:npl]) and
                          torch.equal(ou[cfg.spacing]["match_count"], out["match_count"][:npl]) and
                          torch.equal(ou["frames"]["descriptors"], out["descriptors"]))
            same_u = rank_max(0.0 if same_u else 1.0) == 0.0
            multi_legs["with_upload"] = {"value": round(n * world / dtu, 2), "unit": "frames/s", "ms_per_step": round(dtu * 1e3, 3),
                                         "frac_of_resident_value": round((n * world / dtu) / (n * world * args.steps / dt), 4),
                                         "what": "every rank: pinned host uint8 frames of ITS block -> chunked H2D on a side stream under A0..A9 + M1 "
                                                 "(tokens resident, as in `value`); all ranks concurrently; per-rank local pass (no halo, no gather)",
                                         "equal_to_resident_pass_on_every_rank": same_u}
            del ou, imgs_pin
        if not args.no_vit:
            from sslam_amd.vit import DinoV3ViT
            torch.manual_seed(0)                              # every rank builds the same random DINOv3-architecture weights
            pipe_v = SequencePipeline(cfg, ssd, rsd, device=dev, vit=DinoV3ViT().to(dev).eval())

            def extract_v(_t, im, out=None):
                return pipe_v.extract(pipe_v.tokens_from_images(im), im, out=out)

            runner_v = ShardedSequenceRunner(extract_v, pipe_v.match, spacing=cfg.spacing, alloc_fn=lambda rows: pipe_v.alloc_extract(rows, True),
                                             halo=args.halo)
            step_v = lambda: runner_v.run(toks, imgs, gather=args.gather, frames_per_rank=frames_per_rank)      # noqa: E731
            step_v()
            fence()
            tv_ = time.perf_counter()
            for _ in range(nl):
                ov_ = step_v()
            fence()
            dtv_ = rank_max((time.perf_counter() - tv_) / nl)
            multi_legs["with_vit"] = {"value": round(n * world / dtv_, 2), "unit": "frames/s", "ms_per_step": round(dtv_ * 1e3, 3),
                                      "what": "the sharded step with the HIP ViT inside: images -> A0 -> bf16 HIP ViT-S/16 (random DINOv3-architecture "
                                              "weights, the same on every rank) -> A2..A9 -> M1, halo exchange and gather to rank 0 included",
                                      "pairs_gathered_on_rank0": int(ov_["all_match_count"].shape[0]) if rank == 0 else None,
                                      "matches_per_pair": round(float(ov_["match_count"].float().mean().item()), 1)}
            del ov_, pipe_v, runner_v

    # ---- parity gate (every rank; outside the timed region) ------------------------------------------------------------
    # The metric says "match-index bit-exact vs CPU ref": the last timed step's outputs are compared with the CPU oracle, bit for
    # bit - keypoint indices, scores, descriptors, intensities of every checked frame, and for every checked pair the match count,
    # the match pairs and the quality (visualize_matches_sequence.py:106-197 with the CLI thresholds :381-388, the pair loop
    # :297-320).  N = 1: the WHOLE sequence (613 frames at G = 28: ~3 s of oracle time on 16 threads; 2 965 at G = 40: 28 s), or
    # --parity-frames frames in blocks from both ends and the middle.  N > 1: every rank checks ITS block the same way on its
    # share of the host cores; rank 0 also regenerates the frames on either side of every shard boundary from the seed, runs
    # the oracle on them and compares the boundary pairs' rows of the GATHERED result (the only pairs that exercise the halo),
    # and compares a digest of every rank's local match arrays with the digest of that rank's rows in the gathered arrays
    # (what arrived is what was computed).  Any mismatch on any rank: no value.
    from oracle import ora
    from oracle_check import blocks_for, check_pass, compare_block
    if rank == 0:
        ora.lib()                 # (re)builds oracle/_build/liboracle.so if its sources are newer: one rank only, the others load it after
    if world > 1:
        dist.barrier()
    ora.set_num_threads(max(1, ora.host_threads() // world))
    # default: the whole block (the oracle covers the largest workload, 2 965 frames at G = 40, in 28 s on 16 host threads)
    want = args.parity_frames if args.parity_frames >= 0 else n
    tpar = time.perf_counter()
    parity = check_pass(out, imgs, toks, ssd, rsd, size, K, cfg, blocks_for(n, want, cfg.spacing), frame0=lo)
    parity["checked"] = "keypoint indices, scores, descriptors, intensities, match counts, match pairs, match quality"
    parity["frames_total"], parity["pairs_total"] = n * world, n * world - cfg.spacing
    if world > 1:
        from oracle_check import check_boundaries, check_gathered_rows, digest_matches, merge_rank_reports
        mine = dict(parity, rank=rank, digest=digest_matches(out["matches"], out["quality"], out["match_count"]))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        if rank == 0:
            merged = merge_rank_reports(per_rank)
            gm = {"matches": out["all_matches"], "quality": out["all_quality"], "match_count": out["all_match_count"]}
            if args.test_corrupt_gathered:
                row_ = ranges[1][0] - cfg.spacing
                gm["matches"][row_, 0, 1] = (gm["matches"][row_, 0, 1] + 1) % K

            def regen(a_, b_):          # frames [a_, b_) of the one sequence, from the seed (on this rank's GPU, then to the host)
                bi, bt = synth_sequence(n * world, a_, b_, h, w, grid, dev, seed=1234)
                return bi.cpu().numpy(), bt.cpu().numpy()

            okb, bpairs, bmatches, whyb = check_boundaries(gm, [ranges[r_][0] for r_ in range(1, world)], regen, ssd, rsd, size, K, cfg)
            same_rows = check_gathered_rows(gm, out["pairs_per_rank"], [pr["digest"] for pr in per_rank])
            if not okb or not same_rows:
                merged["bit_exact"] = False
                merged["first_mismatch"] = merged["first_mismatch"] or whyb or "gathered rows differ from a rank's local match arrays"
            merged.update(boundary_pairs_checked=bpairs, boundary_matches_checked=bmatches, boundaries=world - 1,
                          gathered_rows_equal_every_ranks_local_result=same_rows, ranks_checked=world)
            merged["pairs_checked"] += bpairs
            merged["matches_checked"] += bmatches
            parity = merged
    parity["seconds"] = round(time.perf_counter() - tpar, 2)
    parity["oracle_threads"] = ora.num_threads()

    if rank == 0:
        # per step: a sharded step extracts in two launch groups (boundary frames first), so sum the launches of a stage
        stage_ms = {k: round(float(np.sum([a.elapsed_time(b) for a, b in v])) / args.steps, 4) for k, v in ev.items()}
        cells = grid * grid
        conv_flop = n * cells * pipe.selector.hidden * (9 * 384) * 2 + n * cells * pipe.selector.hidden * 2
        conv_s = stage_ms["A3_selector_saliency"] * 1e-3
        achieved = conv_flop / conv_s / 1e12
        # HBM bytes per launch of the dominant kernel come from a separate rocprofv3 --pmc run (FETCH_SIZE / WRITE_SIZE with
        # the gfx950 corrections of MI355X_MICROARCH.md, tools/pmc_summary.py); the committed summary is keyed by workload
        traffic = traffic_src = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))
        pmc = cands[-1] if cands else ""
        if os.path.exists(pmc) and args.workload == "fr1_desk_613" and not args.frames:
            try:
                k = sorted(((name, v) for name, v in json.load(open(pmc)).items() if name.startswith("selector_saliency")),
                           key=lambda nv: "halo" not in nv[0])[0][1]      # the launched form: the halo kernel at G = 28
                traffic = int(k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"])
                traffic_src = ("NOT measured in this run: from the committed rocprofv3 --pmc summary profiles/" + os.path.basename(pmc) +
                               " (FETCH_SIZE / WRITE_SIZE passes of the same command, gfx950 corrections applied)")
            except Exception:
                traffic = traffic_src = None
        ok = bool(parity["bit_exact"])
        rehearse = None
        if rehearsal:
            # the whole sequence in ONE process on this GPU: the gathered result of the sharded run must equal it pair for pair
            imgs_all, toks_all = synth_sequence(n * world, 0, n * world, h, w, grid, dev, seed=1234)
            one = pipe.run(imgs_all, toks_all)
            same = bool(torch.equal(out["all_match_count"], one["match_count"]) and torch.equal(out["all_matches"], one["matches"])
                        and torch.equal(out["all_quality"].view(torch.int32), one["quality"].view(torch.int32)))
            ok = ok and same
            rehearse = {"what": f"{world} ranks sharing {torch.cuda.device_count()} GPU(s), torch.distributed on gloo with device buffers staged "
                                "through host memory: a functional rehearsal of the N-rank path (rank-0 weight broadcast, halo, "
                                f"{args.gather} gather), NOT a measurement",
                        "sharded_equals_single_process": same, "pairs": int(one["match_count"].shape[0]),
                        "matches": int(one["match_count"].sum().item()), "pairs_per_rank": out["pairs_per_rank"],
                        "gather": args.gather, "records_per_rank": out.get("records_per_rank"), "rehearsal_frames_per_s": round(n * world * args.steps / dt, 1)}
            del imgs_all, toks_all, one
        mpeak = measured_mfma_peak() if world == 1 else None
        res = {
            "metric": METRIC, "value": round(n * world * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} frames/GPU of {w}x{h} RGB + ViT-S/16 token grids {grid}x{grid}, "
                                   f"{K} keypoints, extract once + {n - 1} consecutive-pair matches, all-fp32 exact mode",
                       "vit": "A1 (third-party timm ViT) not in the timed region: tokens are an input (SURVEY 8f-1)",
                       "parallelism": f"frame-sharded x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": "selector_saliency_halo_kernel (A3 conv3x3 implicit GEMM, fp32 MFMA; the stage form selector_saliency_kernel on grids whose halo image does not fit)",
                         "achieved": round(achieved, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": n * cells * 384 * 4 + n * cells * 4 + 9 * 384 * pipe.selector.hidden * 4,
                         "flop_per_launch": conv_flop, "launch_ms": stage_ms["A3_selector_saliency"],
                         # context: what the matrix pipe itself sustains on this box (power / clock limited), and the kernel against it
                         "peak_measured": mpeak,
                         "frac_of_measured_peak": (round(achieved / mpeak["f32_burst_tflops"], 4) if mpeak else None)},
            # SURVEY 8d path-level figures: authored-path FLOP (A3 + A7 + A6 + one M1 per frame) and compulsory bytes per frame
            "path_roofline": path_roofline(n * world * args.steps / dt, grid, K, pipe.selector.hidden, h, w),
            "stage_ms": stage_ms,
            "parity": dict(parity, bit_exact=ok, **index_parity_vs_torch(grid)),
            "n_ranks_seen": dist.get_world_size() if world > 1 else 1, "frame_ranges": ranges,
            "matches_per_pair": round(float(out["match_count"].float().mean().item()), 1),
        }
        if sustained is not None:
            res["sustained"] = sustained
        if world == 1 and not args.no_vit:
            res["with_vit"] = vit_leg
        if latency is not None:
            res["latency"] = latency
        if bf16_leg is not None:
            res["bf16_mode"] = bf16_leg
        if upload_leg is not None:
            res["with_upload"] = upload_leg
        if dir_leg is not None:
            res["tum_directory"] = dir_leg
        if world == 1 and not args.no_cpu_baseline:
            nb = min(n, 64)
            res["cpu_baseline"], cb_out = cpu_baseline(imgs[:nb].cpu().numpy(), toks[:nb].cpu().numpy(), ssd, rsd, size, K, cfg)
            # what was timed is what is checked: the timed run's own outputs against the GPU pass (frames 0..63, pairs 0..62)
            okc, _, pc, mc, whyc = compare_block(cb_out, {k_: out[k_][:nb].cpu().numpy() for k_ in ("idx", "scores", "descriptors", "intensity")},
                                                 {k_: out[k_][:nb - cfg.spacing].cpu().numpy() for k_ in ("matches", "quality", "match_count")}, 0, K)
            res["cpu_baseline"]["timed_outputs_equal_gpu_pass"] = {"bit_exact": okc, "frames": nb, "pairs": pc, "matches": mc, "first_mismatch": whyc}
            if not okc:
                ok = False
                res["parity"]["bit_exact"] = False
                res["parity"]["first_mismatch"] = res["parity"].get("first_mismatch") or f"cpu_baseline outputs: {whyc}"
        if rehearse is not None:
            res["rehearsal"], res["value"] = rehearse, None
            res["parallelism_backend"] = "gloo (rehearsal)"
        elif world > 1:
            res["parallelism_backend"] = "nccl (RCCL)"
        if world > 1:
            res["gather"], res["halo"] = args.gather, args.halo
            if multi_legs:
                res.update(multi_legs)
        if not ok and rehearse is None:
            # the metric says "match-index bit-exact vs CPU ref": a run that is not, reports no value and fails
            res["value_unverified"], res["value"] = res["value"], None
        json_out.write(json.dumps(res) + "\n")
        json_out.flush()
        if not ok:
            sys.exit(1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Phase timing inside the ViT row-tile GEMM workgroups.  Needs the probe build:
    make -C semantic-slam-master_amd/csrc clean all EXTRA=-DSSLAM_RT_PROBE && python tools/rt_probe.py [frames]
Prints, per GEMM of one transformer layer, wave 0's mean lifetime and shader-clock cycles per phase."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from sslam_amd import lib
if os.environ.get("SSLAM_BENCH_SO"):            # a probe build kept beside the product library (tools/build_variant.sh)
    lib.SO_PATH = os.path.abspath(os.environ["SSLAM_BENCH_SO"])
L = lib.lib()
if not hasattr(L, "sslam_probe_vit"):
    sys.exit("libsslam_hip.so was not built with -DSSLAM_RT_PROBE")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 41
T, rows = 789, n * 789
dev = "cuda"
x = torch.randn(rows, 384, device=dev)
stats = torch.stack([x[:, :192].sum(1), (x[:, :192] ** 2).sum(1), x[:, 192:].sum(1), (x[:, 192:] ** 2).sum(1)], 1).contiguous()
# drive single GEMMs through the full forward is simplest: run one forward, the probe keeps the LAST launch of each kind...
from sslam_amd.vit import DinoV3ViT
from sslam_amd.vit_hip import HipViT
torch.manual_seed(0)
hv = HipViT(DinoV3ViT().cuda().eval())
img = torch.randn(n, 3, 448, 448, device=dev)
names = ["lifetime", "prologue", "wait+barrier", "mfma loop", "epilogue", "gelu+exchange (fused MLP)"]
for stop, label in [(1, "QKV (ProLN, EpiQKV)"), (2, "o_proj (ProBf16, EpiResidual)"), (3, "MLP: fused kernel, or up (ProLN, EpiGelu) with SSLAM_VIT_NO_FUSED_MLP=1"), (4, "down (ProBf16 x4, EpiResidual) with SSLAM_VIT_NO_FUSED_MLP=1")]:
    L.sslam_test_set_knob(b"SSLAM_RT_STOP", stop, 0)
    for _ in range(2):
        hv.forward_features(img, chunk=n)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 2048, np.uint64)
    assert L.sslam_probe_vit(ctypes.c_void_p(buf.ctypes.data)) == 0
    t = buf.reshape(2048, 8).astype(np.float64)
    if stop == 3 and not os.environ.get("SSLAM_VIT_NO_FUSED_MLP"):
        t = t[: (rows + 127) // 128]          # the fused kernel runs one workgroup per row tile; later entries are stale
    t = t[t[:, 0] > 0]
    print(f"{label}: {len(t)} workgroups")
    life_us = t[:, 6] * 0.01
    start_us = (t[:, 7] - t[:, 7].min()) * 0.01
    span = (start_us + life_us).max()
    print(f"   shader clock {t[:, 0].sum() / t[:, 6].sum() * 0.1:5.3f} GHz (cycles per 100 MHz tick inside a wave); workgroup life {life_us.mean():6.1f} us; "
          f"launch span {span:7.1f} us; mean workgroups in flight {life_us.sum() / span:6.1f}")
    for i, nm in enumerate(names):
        print(f"   {nm:14s} {t[:, i].mean():10.0f} cycles ({100 * t[:, i].mean() / t[:, 0].mean():5.1f} %)   min {t[:, i].min():9.0f} max {t[:, i].max():9.0f}")

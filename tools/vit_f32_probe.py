#!/usr/bin/env python3
"""Phase timing inside the workgroups of the fp32 ViT's per-layer GEMM (gemm_f32_rows_kernel) and attention, from a probe build of the
library (make ... EXTRA=-DSSLAM_CLOCK_PROBE) passed as argv[1].  Runs each of the four GEMMs of a layer alone on 83 frames'
rows and prints wave 0's mean cycles in the prologue (first loads + first barrier), the k loop and the epilogue, and the span
from the first workgroup's start to the last one's end."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from sslam_amd import lib
lib.SO_PATH = os.path.abspath(sys.argv[1])
from sslam_amd.vit import DinoV3ViT
from sslam_amd.vit_hip import HipViTF32
L = lib.lib()
if not hasattr(L, "sslam_probe_gemm_f32"):
    sys.exit("not a probe build (-DSSLAM_CLOCK_PROBE)")
torch.manual_seed(0)
vit = DinoV3ViT(depth=12).cuda().eval()
hv = HipViTF32(vit)
x = torch.randn(83, 3, 448, 448, device="cuda")
import time
t0 = time.time()
while time.time() - t0 < float(os.environ.get("SSLAM_PROBE_WARM_S", "4")):      # let the clock settle where a sequence holds it
    hv.forward_features(x)
    torch.cuda.synchronize()
for name, K, ntn in (("QKV + RoPE", 384, 9), ("o_proj + residual", 384, 3), ("up + GELU", 384, 12), ("down + residual", 1536, 3)):
    assert L.sslam_probe_gemm_f32_select(K, ntn) == 0
    hv.forward_features(x)
    torch.cuda.synchronize()
    buf = np.zeros(4 * 8192, np.uint64)
    assert L.sslam_probe_gemm_f32(ctypes.c_void_p(buf.ctypes.data)) == 0
    t = buf.reshape(8192, 4).astype(np.float64)
    t = t[t[:, 1] > 0]
    nt = K // 32
    ghz = t[:, :3].sum(1).sum() / t[:, 3].sum() * 0.1          # cycles per 100 MHz tick
    print(f"{name:18s} {len(t):5d} workgroups: prologue {t[:, 0].mean():7.0f}  k loop {t[:, 1].mean():8.0f} ({t[:, 1].mean() / nt:6.0f} per k tile; "
          f"matrix time per SIMD and tile at three waves: 12288)  epilogue {t[:, 2].mean():7.0f}  sum {t[:, :3].sum(1).mean():8.0f}  shader clock {ghz:5.3f} GHz")
buf = np.zeros(4 * 8192, np.uint64)
assert L.sslam_probe_attn_f32(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(8192, 4).astype(np.float64)
t = t[t[:, 0] > 0]
n_kt = (789 + 31) // 32
start = (t[:, 3] - t[:, 3].min()) * 0.01            # us
life = t[:, 1] * 0.01
print(f"attention: kernel span {(start + life).max():7.1f} us; workgroup life mean {life.mean():6.1f} us, min {life.min():6.1f}, max {life.max():6.1f}; "
      f"starts per 50 us: {np.histogram(start, bins=np.arange(0, (start + life).max() + 50, 50))[0].tolist()}")
order = np.argsort(start)
print("   life by start order (deciles, us):", [round(float(life[order[i:i + len(t) // 10]].mean()), 1) for i in range(0, len(t) - len(t) // 10 + 1, len(t) // 10)])
print(f"attention          {len(t):5d} workgroups: life {t[:, 0].mean():8.0f} cycles ({t[:, 0].mean() / n_kt:6.0f} per key tile; matrix time per SIMD and tile at "
      f"three waves: 12288)  shader clock {t[:, 0].sum() / t[:, 1].sum() * 0.1:5.3f} GHz")

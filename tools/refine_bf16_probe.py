#!/usr/bin/env python3
"""Phase timing inside the workgroups of the row-resident bf16 descriptor MLP (refine_bf16_rows_kernel).  Needs a probe
build of the library (make ... EXTRA=-DSSLAM_CLOCK_PROBE), passed as argv[1]:

    python tools/refine_bf16_probe.py tools/microbench/_variants/libsslam_probe.so [x_in]

Prints, for waves 0, 5 and 11 of the first 4096 workgroups, the mean lifetime and the shader-clock cycles per phase."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
lib.SO_PATH = os.path.abspath(sys.argv[1])
from sslam_amd.pipeline import PackedRefiner
L = lib.lib()
if not hasattr(L, "sslam_probe_refine_bf16"):
    sys.exit("not a probe build (-DSSLAM_CLOCK_PROBE)")
F, K, G = 613, 500, 28
ref = PackedRefiner(synth.refiner_state(0), "cuda", bf16=True)
torch.manual_seed(0)
feat = torch.randn(F, G, G, 384, device="cuda")
kp = torch.rand(F, K, 2, device="cuda") * (G - 1)
x = torch.randn(F * K, 384, device="cuda")
use_x = len(sys.argv) > 2
for _ in range(3):
    lib.refine_bf16(x, ref.packed_bf16, ref.n_blocks) if use_x else lib.gather_refine_bf16(feat, kp, ref.packed_bf16, ref.n_blocks)
torch.cuda.synchronize()
buf = np.zeros(3 * 8 * 4096, np.uint64)
assert L.sslam_probe_refine_bf16(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(3, 4096, 8).astype(np.float64)[:, : (F * K + 95) // 96]
names = ["lifetime", "gather", "gemm loops (5)", "epilogues (5)", "out-proj gemm", "barriers"]
print("x_in entry" if use_x else "fused gather entry")
for w, tag in enumerate(("wave 0", "wave 5", "wave 11")):
    print(tag, " ".join(f"{n} {t[w, :, i].mean():8.0f}" for i, n in enumerate(names)))

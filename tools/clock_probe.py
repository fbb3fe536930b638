#!/usr/bin/env python3
"""Phase timing inside the descriptor-MLP workgroups (DESIGN_HISTORY.md section 9).  Needs the probe build of the kernel:

    make -C semantic-slam-master_amd/csrc clean all EXTRA=-DSSLAM_CLOCK_PROBE && python tools/clock_probe.py
    make -C semantic-slam-master_amd/csrc clean all        # back to the product build

Prints, for wave 0 of the first 4096 workgroups, the mean lifetime and the shader-clock cycles spent per phase."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
if os.environ.get("SSLAM_BENCH_SO"):            # a variant build of the library (the probe build)
    lib.SO_PATH = os.path.abspath(os.environ["SSLAM_BENCH_SO"])
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
L = lib.lib()
if not hasattr(L, "sslam_probe_refine"):
    sys.exit("libsslam_hip.so was not built with -DSSLAM_CLOCK_PROBE")
pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0))
toks = torch.randn(613, 789, 384, device="cuda") * 3 + 0.5
for _ in range(3):
    pipe.extract(toks)
torch.cuda.synchronize()
buf = np.zeros(8 * 4096, np.uint64)
assert L.sslam_probe_refine(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(4096, 8).astype(np.float64)
names = ["lifetime", "gather", "gemm loops (6)", "layernorm (4)", "tile stores (5)", "barriers"]
for i, n in enumerate(names):
    print(f"{n:16s} {t[:, i].mean():10.0f} cycles  ({100 * t[:, i].mean() / t[:, 0].mean():5.1f} %)")

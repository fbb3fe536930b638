"""Phase timing inside the MLP workgroups (needs a probe build of the library in place): see DESIGN.md section 9."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth
from sslam_amd import lib
from sslam_amd.pipeline import ExtractorConfig, SequencePipeline
pipe = SequencePipeline(ExtractorConfig(), synth.selector_state(0), synth.refiner_state(0))
toks = torch.randn(613, 789, 384, device="cuda") * 3 + 0.5
for _ in range(3):
    pipe.extract(toks)
torch.cuda.synchronize()
L = lib.lib()
buf = np.zeros(2 * 4096, np.uint64)
L.sslam_probe_refine(ctypes.c_void_p(buf.ctypes.data))
a, b = buf[0::2], buf[1::2]
print("gather", np.mean((a >> np.uint64(32)).astype(np.float64)), "store_tile x5", np.mean((a & np.uint64(0xffffffff)).astype(np.float64)), "layernorm x4", np.mean(b.astype(np.float64)))

"""Times the bf16-mode kernels on synthetic tokens (613 frames): python tools/bench_bf16.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semantic-slam-master_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import synth
from sslam_amd import lib
if os.environ.get("SSLAM_BENCH_SO"):            # a variant build of the library (experiments)
    lib.SO_PATH = os.path.abspath(os.environ["SSLAM_BENCH_SO"])

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

n, g = 613, 28
sd = synth.selector_state(0)
feat = torch.randn(n, g, g, 384, device="cuda")
w1p = torch.from_numpy(lib.pack_conv3x3_bf16(sd["conv.0.weight"])).cuda().view(torch.bfloat16)
b1 = torch.from_numpy(sd["conv.0.bias"]).cuda(); w2 = torch.from_numpy(sd["conv.2.weight"].reshape(-1).copy()).cuda(); b2 = torch.from_numpy(sd["conv.2.bias"]).cuda()
fb = lib.to_bf16(feat)
sal = torch.empty(n, g, g, device="cuda")
t_cvt = timeit(lambda: lib.to_bf16(feat, out=fb))
t_conv = timeit(lambda: lib.selector_saliency_bf16(fb, w1p, b1, w2, b2, 256, out=sal))
fl = 2.0 * n * g * g * 3456 * 256
print(f"f32->bf16 {t_cvt:.3f} ms   conv bf16 {t_conv:.3f} ms  ({fl / t_conv / 1e9:.1f} TFLOP/s)")

rsd = synth.refiner_state(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sslam_amd.pipeline import refiner_weight_list
ws, nb = refiner_weight_list(rsd)
pk = torch.from_numpy(lib.pack_refiner_bf16(ws, nb)).cuda()
pk32 = torch.from_numpy(lib.pack_refiner(ws, nb)).cuda()
K = 500
kp = (torch.rand(n, K, 2, device="cuda") * (g - 1)).contiguous()
desc = torch.empty(n, K, 128, device="cuda")
t_r = timeit(lambda: lib.gather_refine_bf16(feat, kp, pk, nb, out=desc))
t_r32 = timeit(lambda: lib.gather_refine(feat, kp, pk32, nb, out=desc))
fl = n * K * 1572864.0
print(f"gather+refine bf16 {t_r:.3f} ms ({fl / t_r / 1e9:.1f} TFLOP/s)   fp32 {t_r32:.3f} ms")

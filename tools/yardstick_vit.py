"""Library yardstick for the ViT-S/16 shapes (NOT used by the product): torch.matmul (hipBLASLt/rocBLAS) and SDPA in bf16."""
import torch
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 64 * 789
for (K, N, name) in [(384, 1152, "qkv"), (384, 384, "proj"), (384, 1536, "fc1"), (1536, 384, "fc2")]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: torch.matmul(x, w.t()))
    print(f"{name:5s} M={M} K={K} N={N}: {t*1e3:8.1f} us  {2.0*M*K*N/t/1e9:8.1f} TFLOP/s")
q = torch.randn(64, 6, 789, 64, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
print(f"sdpa 64x6x789x64: {t*1e3:8.1f} us  {4.0*64*6*789*789*64/t/1e9:8.1f} TFLOP/s")

// Does the 16x16x32 bf16 MFMA shape deliver more FLOP/s than 32x32x16 in the regime the ViT GEMMs run in - every A fragment
// read from LDS (1 KB per ds_read_b128, one per 32 cycles of matrix work), B fragments in registers, random operands, the
// chip holding its clock down under the load?  (MI355X_MICROARCH.md, DVFS item 7: +12-15 % in bare loops.)
// Same FLOP per LDS byte in both forms: a 1 KB A fragment feeds one 32x32x16 MFMA (32 768 FLOP, 32 cycles) or two 16x16x32
// MFMAs on two 16-token halves (2 x 16 384 FLOP, 2 x 16 cycles).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NF = 24;               // A fragments per sweep (24 KB of LDS)

template <int SHAPE, int THREADS>
__global__ __launch_bounds__(THREADS) void shape_kernel(const u32x4 *__restrict__ src, float *out, int iters) {
    __shared__ u32x4 lds[NF * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < NF * 64; i += THREADS) lds[i] = src[i];
    bf16x8 b[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) b[f] = __builtin_bit_cast(bf16x8, src[NF * 64 + f * 64 + lane]);
    __syncthreads();
    f32x16 acc32[6];
    f32x4 acc16[2][6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) acc32[i][e] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) acc16[0][i][e] = acc16[1][i][e] = 0.f;
    }
    for (int it = 0; it < iters; it++) {
        asm volatile("" ::: "memory");          // the fragments are re-read from LDS in every sweep (no hoisting into registers)
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, lds[f * 64 + lane]);
            if (SHAPE == 32) {
                acc32[f % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[f], acc32[f % 6], 0, 0, 0);
            } else {
                acc16[0][f % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[f], acc16[0][f % 6], 0, 0, 0);
                acc16[1][f % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[(f + 1) % NF], acc16[1][f % 6], 0, 0, 0);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int e = 0; e < 16; e++) s += acc32[i][e];
#pragma unroll
        for (int e = 0; e < 4; e++) s += acc16[0][i][e] + acc16[1][i][e];
    }
    out[(size_t)blockIdx.x * THREADS + tid] = s;
}

template <int SHAPE, int THREADS>
double run(const u32x4 *src, float *out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((shape_kernel<SHAPE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, src, out, iters / 8);
    hipDeviceSynchronize();
    double best = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((shape_kernel<SHAPE, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * (THREADS / 64) * iters * NF * 32768.0;
        const double tf = flop / (ms * 1e-3) / 1e12;
        if (rep >= 2) best += tf / 3;          // mean of the last three (the clock has settled)
    }
    return best;
}

int main() {
    std::vector<unsigned> h(2 * NF * 64 * 4);
    srand(1);
    for (auto &v : h) {   // random bf16 pairs in [-1, 1): random mantissas, small exponents
        const unsigned short lo = (unsigned short)(0x3f00 | (rand() & 0x80ff)), hi = (unsigned short)(0x3f00 | (rand() & 0x80ff));
        v = lo | ((unsigned)hi << 16);
    }
    u32x4 *src;
    float *out;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, 4096 * 512 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; pass++) {
        printf("one wave per SIMD  (256 thr x 256 blocks): 32x32x16 %7.1f TFLOP/s   16x16x32 %7.1f TFLOP/s\n",
               run<32, 256>(src, out, 256, 20000), run<16, 256>(src, out, 256, 20000));
        printf("two waves per SIMD (512 thr x 256 blocks): 32x32x16 %7.1f TFLOP/s   16x16x32 %7.1f TFLOP/s\n",
               run<32, 512>(src, out, 256, 10000), run<16, 512>(src, out, 256, 10000));
    }
    return 0;
}

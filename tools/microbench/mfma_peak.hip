// Matrix-pipe rate of v_mfma_f32_32x32x2_f32 (and the bf16 32x32x16 form) with no memory traffic: every wave of every CU
// issues independent accumulation chains.  This is the clock-limited ceiling the fp32 kernels of libsslam_hip.so can be
// compared with (the nominal 157.3 TFLOP/s assumes 2.4 GHz; under matrix load the box runs at about 2.27 GHz).
// build (done by __graft_entry__.build()): hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; ... -shared -fPIC -o libmfma_peak.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void f32_kernel(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int e = 0; e < 16; e++) acc[i][e] = (float)threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++)
        for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void bf16_kernel(float *out, int iters, float a) {
    f32x16 acc[NACC];
    bf16x8 x, y;
    for (int j = 0; j < 8; j++) { x[j] = (__bf16)a; y[j] = (__bf16)(a * 0.5f); }
    for (int i = 0; i < NACC; i++)
        for (int e = 0; e < 16; e++) acc[i][e] = (float)threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++)
        for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// C entry for bench.py (ctypes): TFLOP/s of one fp32 (kind 0) or bf16 (kind 1) run of about `iters` loop trips
extern "C" double mfma_peak_tflops(int kind, int iters) {
    static float *out = nullptr;
    const int blocks = 256 * 2;   // 2 waves per SIMD x 4 accumulators: 149 TFLOP/s fp32; see the sweep in main() - two odd
                                  // combinations (4 x 4 and 8 x 2) run 10-20 % slower for no reason we could establish
    if (!out && hipMalloc(&out, blocks * 256 * sizeof(float)) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    hipEventRecord(e0);
    if (kind == 0)
        hipLaunchKernelGGL(f32_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
    else
        hipLaunchKernelGGL(bf16_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-20f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    const double flop = (double)blocks * 4 * iters * 8 * 4 * (kind == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16);
    return flop / ms / 1e9;
}

int main() {
    float *out;
    const int blocks = 256 * 4;   // 4 workgroups of 4 waves per CU
    hipMalloc(&out, 8 * 256 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // pass 0 / 1: ~170 ms runs (clock settles at its power-limited value); pass 2 / 3: ~6.5 ms bursts separated by 20 ms of
    // idle time, the duty cycle of one kernel inside a larger pipeline
    for (int pass4 = 0; pass4 < 4; pass4++) {
        const int pass = pass4 & 1;
        const bool burst = pass4 >= 2;
        for (int rep = 0; rep < (burst ? 6 : 3); rep++) {
            const int iters = (pass == 0 ? 40000 : 80000) / (burst ? 26 : 1);
            if (burst) { hipDeviceSynchronize(); usleep(20000); }
            hipEventRecord(e0);
            if (pass == 0)
                hipLaunchKernelGGL(f32_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
            else
                hipLaunchKernelGGL(bf16_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-20f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = (double)blocks * 4 * iters * 8 * 4;
            const double flop = mfmas * (pass == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16);
            printf("%s %s %8.3f ms  %8.1f TFLOP/s  (%.0f MHz equivalent at %d flop/clk/CU)\n", pass == 0 ? "f32 32x32x2  " : "bf16 32x32x16", burst ? "burst    " : "sustained",
                   ms, flop / ms / 1e9, flop / ms / 1e3 / 256.0 / (pass == 0 ? 256.0 : 4096.0), pass == 0 ? 256 : 4096);
        }
    }
    // occupancy sweep (fp32, ~7 ms): 1, 2, 4 workgroups of 4 waves per CU and 1 / 2 / 4 independent accumulators per wave
    for (int k = 1; k <= 8; k *= 2)
        for (int nacc = 1; nacc <= 8; nacc *= 2) {
            hipDeviceSynchronize();
            usleep(20000);
            const int iters = 1540 * 16 / k / nacc;
            hipEventRecord(e0);
            if (nacc == 1) hipLaunchKernelGGL(f32_kernel<1>, dim3(256 * k), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
            if (nacc == 2) hipLaunchKernelGGL(f32_kernel<2>, dim3(256 * k), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
            if (nacc == 4) hipLaunchKernelGGL(f32_kernel<4>, dim3(256 * k), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
            if (nacc == 8) hipLaunchKernelGGL(f32_kernel<8>, dim3(256 * k), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)256 * k * 4 * iters * 8 * nacc * 2.0 * 32 * 32 * 2;
            printf("f32 sweep: %d waves/SIMD x %d accumulators  %8.3f ms  %8.1f TFLOP/s\n", k, nacc, ms, flop / ms / 1e9);
        }
    // the occupancies the library's fp32 kernels actually run at: 3 waves/SIMD x 3 accumulators (descriptor MLP),
    // 4 x 4 (saliency CNN, two workgroups per CU), 2 x 1 (matcher)
    for (int cfg = 0; cfg < 4; cfg++) {
        const int k = cfg == 0 ? 3 : (cfg == 1 ? 3 : (cfg == 2 ? 6 : 2)), nacc = cfg == 0 ? 3 : (cfg == 1 ? 6 : (cfg == 2 ? 3 : 1));
        hipDeviceSynchronize();
        usleep(20000);
        const int iters = 1540 * 16 / k / nacc;
        hipEventRecord(e0);
        if (nacc == 1) hipLaunchKernelGGL(f32_kernel<1>, dim3(256 * k), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
        if (nacc == 3) hipLaunchKernelGGL(f32_kernel<3>, dim3(256 * k), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
        if (nacc == 6) hipLaunchKernelGGL(f32_kernel<6>, dim3(256 * k), dim3(256), 0, 0, out, iters, 1.0f, 1e-30f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)256 * k * 4 * iters * 8 * nacc * 2.0 * 32 * 32 * 2;
        printf("f32 sweep: %d waves/SIMD x %d accumulators  %8.3f ms  %8.1f TFLOP/s\n", k, nacc, ms, flop / ms / 1e9);
    }
    return 0;
}

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// each wave copies 4 KB: 4 DMA instructions of 1 KB, per-lane source, LDS dest = uniform base + lane*16
__global__ __launch_bounds__(256) void probe(const u32x4 *src, u32x4 *out, int perm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int piece = wave * 4 + i;
        const int sl = perm ? (63 - lane) : lane;     // per-lane SOURCE address: reversed order when perm
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 64 + sl),
                                         (__attribute__((address_space(3))) void *)(smem + piece * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = reinterpret_cast<u32x4 *>(smem)[i];
}
int main() {
    u32x4 *h = (u32x4 *)malloc(16384), *o = (u32x4 *)malloc(16384);
    for (int i = 0; i < 1024; i++) h[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    u32x4 *d, *dout;
    hipMalloc(&d, 16384); hipMalloc(&dout, 16384);
    hipMemcpy(d, h, 16384, hipMemcpyHostToDevice);
    for (int perm = 0; perm < 2; perm++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 16384, 0, d, dout, perm);
        hipMemcpy(o, dout, 16384, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; i++) {
            int want = perm ? (i / 64) * 64 + 63 - (i % 64) : i;
            if ((int)o[i][0] != want || o[i][3] != 3u) bad++;
        }
        printf("perm %d: bad %d  (o[0]=%u o[1]=%u o[64]=%u)\n", perm, bad, o[0][0], o[1][0], o[64][0]);
    }
    return 0;
}

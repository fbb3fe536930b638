// How fast can every CU stream the SAME L2-resident weight buffer (the descriptor MLP's 1.57 MB, the ViT MLP's 2.36 MB), and by
// which path?  (a) LDS-DMA (global_load_lds_dwordx4, 1 KB pieces into an LDS ring, nobody reads them), (b) per-lane 16-byte
// loads into registers.  Waves keep `depth` pieces in flight each.  Prints aggregate TB/s and GB/s per CU: the ceiling of any
// kernel that re-streams its weights per 64-row / 128-token workgroup.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const u32x4 *__restrict__ src, int pieces_total, int sweeps, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *ring = smem + wave * DEPTH * 1024;
    for (int s = 0; s < sweeps; s++) {
        for (int p = wave, i = 0; p < pieces_total; p += WAVES, i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)p * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(ring + (i % DEPTH) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && sweeps < 0) sink[0] = *reinterpret_cast<unsigned *>(smem);
}

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void reg_kernel(const u32x4 *__restrict__ src, int pieces_total, int sweeps, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < sweeps; s++) {
        for (int p = wave; p < pieces_total; p += WAVES * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const int q = p + d * WAVES;
                v[d] = src[(size_t)(q < pieces_total ? q : p) * 64 + lane];
            }
#pragma unroll
            for (int d = 0; d < DEPTH; d++) acc ^= v[d];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc[0];
}

template <class K>
void timeit(const char *label, K launch, double bytes_per_launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-58s %8.3f ms  %6.2f TB/s  %6.1f GB/s per CU\n", label, ms, bytes_per_launch / ms / 1e9, bytes_per_launch / ms / 1e6 / 256);
}

int main() {
    for (int kb : {1536, 2304}) {
        const int pieces = kb;             // 1 KB pieces
        u32x4 *src;
        unsigned *sink;
        hipMalloc(&src, (size_t)kb * 1024);
        hipMalloc(&sink, 4096);
        hipMemset(src, 1, (size_t)kb * 1024);
        const int sweeps = 16, blocks = 256;
        const double bytes = (double)blocks * sweeps * kb * 1024.0;
        printf("buffer %d KB, every one of %d workgroups (one per CU) streams it %d times\n", kb, blocks, sweeps);
        timeit("LDS-DMA, 4 waves x 8 pieces in flight", [&] { hipLaunchKernelGGL((dma_kernel<4, 8>), dim3(blocks), dim3(256), 4 * 8 * 1024, 0, src, pieces, sweeps, sink); }, bytes);
        timeit("LDS-DMA, 4 waves x 16 pieces in flight", [&] { hipLaunchKernelGGL((dma_kernel<4, 16>), dim3(blocks), dim3(256), 4 * 16 * 1024, 0, src, pieces, sweeps, sink); }, bytes);
        timeit("LDS-DMA, 8 waves x 8 pieces in flight", [&] { hipLaunchKernelGGL((dma_kernel<8, 8>), dim3(blocks), dim3(512), 8 * 8 * 1024, 0, src, pieces, sweeps, sink); }, bytes);
        timeit("LDS-DMA, 8 waves x 16 pieces in flight", [&] { hipLaunchKernelGGL((dma_kernel<8, 16>), dim3(blocks), dim3(512), 8 * 16 * 1024, 0, src, pieces, sweeps, sink); }, bytes);
        timeit("registers, 4 waves x 8 loads in flight", [&] { hipLaunchKernelGGL((reg_kernel<4, 8>), dim3(blocks), dim3(256), 0, 0, src, pieces, sweeps, sink); }, bytes);
        timeit("registers, 8 waves x 8 loads in flight", [&] { hipLaunchKernelGGL((reg_kernel<8, 8>), dim3(blocks), dim3(512), 0, 0, src, pieces, sweeps, sink); }, bytes);
        timeit("registers, 8 waves x 16 loads in flight", [&] { hipLaunchKernelGGL((reg_kernel<8, 16>), dim3(blocks), dim3(512), 0, 0, src, pieces, sweeps, sink); }, bytes);
        hipFree(src);
        hipFree(sink);
    }
    return 0;
}

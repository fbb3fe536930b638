#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
// LDS tile [64 rows][64 cols] of 16-bit values value = row*64+col; each lane supplies the address of (row q, cols 4p..4p+3)
// of a 4x16 block; expect lane i of a 16-lane group to receive column i of the block's 4 rows.
__global__ void probe(short *out) {
    __shared__ __attribute__((aligned(16))) short tile[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) tile[i] = (short)i;
    __syncthreads();
    const int L = threadIdx.x, G = L >> 4, idx = L & 15, q4 = idx >> 2, p4 = idx & 3;
    const int row0 = 8 * G, col0 = 16 * (G & 1);       // a different block per group
    const short *addr = &tile[(row0 + q4) * 64 + col0 + 4 * p4];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)addr);
    for (int e = 0; e < 4; e++) out[L * 4 + e] = v[e];
}
int main() {
    short *d, h[256];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < 64; L++) {
        int G = L >> 4, idx = L & 15, row0 = 8 * G, col0 = 16 * (G & 1);
        for (int e = 0; e < 4; e++) {
            int want = (row0 + e) * 64 + col0 + idx;      // element e = row e of the block, column idx
            if (h[L * 4 + e] != want) bad++;
        }
    }
    printf("tr16_b64: bad %d; lane0 = %d %d %d %d, lane1 = %d %d %d %d, lane17 = %d %d %d %d\n", bad, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7],
           h[68], h[69], h[70], h[71]);
    return 0;
}

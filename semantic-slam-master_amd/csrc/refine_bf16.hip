// refine_bf16.hip - A6 + A7 in the bf16 THROUGHPUT mode (BASELINE.json configs[1]): bilinear
// gather + descriptor MLP (dino_backbone.py:114-152, descriptor_refiner.py:58-126) with bf16 GEMM operands on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 residual / LayerNorm statistics / L2 normalisation.
// NOT bit-exact against the fp32 reference by construction; the exact kernel is refine.hip.
//
// One workgroup (4 waves) owns 64 rows for the whole chain; wave w owns output columns [96w, 96w+96) of every layer
// (2 x 3 accumulator tiles), so the residual identity never leaves its registers.  The activation tile in LDS is
// bf16 (64 x 392, 50 KB -> two workgroups per CU); weights stream from L2 in fragment order (ring of 3 k-steps).
// LayerNorm is folded into the next GEMM so the tile can be written before the row statistics are known:
//     LN(x) W^T + b = rstd * (x (W*gamma)^T - mean * colsum(W*gamma)) + (b + W beta)
// x goes into the tile as bf16(x) straight from the accumulators; mean / rstd come from the fp32 values (row sums
// by a reduce-scatter butterfly over the 32 lanes of a half-wave, then across the 4 waves through LDS) and are
// applied in the epilogue of the next layer.  Two barriers per layer.
// Round 4: a row-resident form (96 rows x 12 waves, private weight streams straight from L2, transposed product with lane-local
// row statistics) was built, verified and probed - equal on the MLP (0.61 vs 0.62 ms), slower with the gather fused (0.80 vs
// 0.74 ms: one workgroup per CU leaves the HBM-bound gather exposed); it is in the history at commit e36950a, DESIGN_HISTORY.md.
#include "common.h"
#include "gather_taps.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RM = 64, NTHR = 256, HID = SSLAM_HID, KS = HID / 16;
constexpr int LDT = HID + 8;   // bf16 elements per tile row: 784 B = 49 x 16 B
constexpr long long LAYER_W_BYTES = (long long)HID * HID * 2;
constexpr long long LAYER_BYTES = LAYER_W_BYTES + 2 * HID * 4;        // weights + v0[384] + v1[384]
constexpr long long OUT_W_BYTES = (long long)SSLAM_D * HID * 2;
constexpr long long OUT_BYTES = OUT_W_BYTES + 2 * SSLAM_D * 4;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // one v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// acc[mi][t] += tile(64 x 384, bf16) . W^T for this wave's NT column tiles; W in fragment order [ks][n/32][64 lanes][8]
template <int NT>
__device__ __forceinline__ void gemm_bf16(const bf16 *tile, const unsigned char *__restrict__ w, int lane, int wn,
                                          f32x16 (&acc)[2][NT]) {
    constexpr int TILES = 4 * NT;   // N / 32
    const int r = lane & 31, h = lane >> 5;
    // buffer loads: descriptor over this layer's weights + 32-bit lane offset + SCALAR (k-step, tile) offset - no vector
    // address arithmetic in the k loop (every non-MFMA instruction costs matrix-pipe issue time, DESIGN_HISTORY.md section 9)
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(w), 0, KS * TILES * 1024, 0x00020000);
    const int loff = ((wn * NT) * 64 + lane) * 16;
    const bf16 *A = tile + r * LDT + 8 * h;
    bf16x8 q0[NT], q1[NT], q2[NT];
#define LOAD_B(dst, ks)                                                                            \
    _Pragma("unroll") for (int t = 0; t < NT; t++) dst[t] = __builtin_bit_cast(                    \
        bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, ((ks) * TILES + t) * 1024, 0));
#define STEP(cur, ks)                                                                              \
    {                                                                                              \
        const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(A + (ks) * 16);                        \
        const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(A + 32 * LDT + (ks) * 16);             \
        _Pragma("unroll") for (int t = 0; t < NT; t++) {                                           \
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, cur[t], acc[0][t], 0, 0, 0);   \
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, cur[t], acc[1][t], 0, 0, 0);   \
        }                                                                                          \
    }
    LOAD_B(q0, 0);
    LOAD_B(q1, 1);
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 3) {
        LOAD_B(q2, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
        STEP(q0, ks);
        __builtin_amdgcn_sched_barrier(0);
        LOAD_B(q0, min(ks + 3, KS - 1));      // unconditional (clamped): the compiler keeps an exact count of loads in flight
        __builtin_amdgcn_sched_barrier(0);
        STEP(q1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        LOAD_B(q1, min(ks + 4, KS - 1));
        __builtin_amdgcn_sched_barrier(0);
        STEP(q2, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef LOAD_B
#undef STEP
}

// v[32] per lane (index i) -> returns, in lane r of each half-wave, the sum over the half-wave's 32 lanes of v[r]
__device__ __forceinline__ float reduce_scatter32(float (&v)[32], int r) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        const bool up = (r & m) != 0;
#pragma unroll
        for (int i = 0; i < m; i++) {
            float lo = v[i], hi = v[i + m];
            asm volatile("" : "+v"(lo), "+v"(hi));     // keeps the selects from being folded into a dynamic vector index
            const float keep = up ? hi : lo, send = up ? lo : hi;
            v[i] = keep + __shfl_xor(send, m);
        }
    }
    return v[0];
}

// tile <- bf16(v) (this wave's 64 x 96 slab) and the wave's partial row sums / sums of squares -> part[wn][row]
__device__ __forceinline__ void store_slab(bf16 *tile, float *part_s, float *part_q, const f32x16 (&v)[2][3], int lane, int wn) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                tile[(mi * 32 + crow(e, h)) * LDT + wn * 96 + t * 32 + r] = (bf16)v[mi][t][e];
    const int row = (r >> 4) * 32 + crow(r & 15, h);
    float s[32];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int e = 0; e < 16; e++) s[mi * 16 + e] = (v[mi][0][e] + v[mi][1][e]) + v[mi][2][e];
    part_s[wn * RM + row] = reduce_scatter32(s, r);
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int e = 0; e < 16; e++)
            s[mi * 16 + e] = __builtin_fmaf(v[mi][2][e], v[mi][2][e], __builtin_fmaf(v[mi][1][e], v[mi][1][e], v[mi][0][e] * v[mi][0][e]));
    part_q[wn * RM + row] = reduce_scatter32(s, r);
}

__global__ __launch_bounds__(NTHR, 2) void refine_bf16_kernel(const float *__restrict__ feat, int G, const float *__restrict__ kp_xy,
                                                               int K, const float *__restrict__ x_in, long long rows,
                                                               const unsigned char *__restrict__ pk, int n_blocks,
                                                               float *__restrict__ desc) {
    __shared__ __attribute__((aligned(16))) bf16 tile[RM * LDT];
    __shared__ __attribute__((aligned(16))) float part_s[4 * RM], part_q[4 * RM], st_mean[RM], st_rstd[RM];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, r = lane & 31, h = lane >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8; give every XCD one contiguous range of row tiles so that the ~8
    // tiles gathering from one frame's feature map share that XCD's L2 instead of fetching the frame into all eight
    long long R0;
    {
        const int n_tiles = gridDim.x, b = blockIdx.x, q = n_tiles / 8, rem = n_tiles % 8, x = b % 8;
        R0 = (long long)((x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8) * RM;
    }

    // ---- phase 0: activation tile <- bf16(gathered features | x_in rows) ---------------------------------------
    {
        const int row = tid >> 2, part = tid & 3;
        long long R = R0 + row;
        if (R > rows - 1) R = rows - 1;
        bf16 *dst = tile + row * LDT + part * 96;
        if (feat) {
            const long long f = R / K;
            const Taps t = make_taps(feat + f * G * G * SSLAM_C, G, kp_xy[2 * R], kp_xy[2 * R + 1]);
#pragma unroll 4
            for (int j = 0; j < 12; j++) {
                const int c0 = part * 96 + 8 * j;
                const float4 lo = blend4(t, c0), hi = blend4(t, c0 + 4);
                u32x4 o;
                o[0] = pack_bf16(lo.x, lo.y); o[1] = pack_bf16(lo.z, lo.w); o[2] = pack_bf16(hi.x, hi.y); o[3] = pack_bf16(hi.z, hi.w);
                *reinterpret_cast<u32x4 *>(dst + 8 * j) = o;
            }
        } else {
            const float *src = x_in + R * SSLAM_C + part * 96;
#pragma unroll 4
            for (int j = 0; j < 12; j++) {
                const float4 lo = *reinterpret_cast<const float4 *>(src + 8 * j), hi = *reinterpret_cast<const float4 *>(src + 8 * j + 4);
                u32x4 o;
                o[0] = pack_bf16(lo.x, lo.y); o[1] = pack_bf16(lo.z, lo.w); o[2] = pack_bf16(hi.x, hi.y); o[3] = pack_bf16(hi.z, hi.w);
                *reinterpret_cast<u32x4 *>(dst + 8 * j) = o;
            }
        }
    }
    __syncthreads();

    f32x16 X[2][3], acc[2][3];
    // ---- input_proj + ReLU ---------------------------------------------------------------------------------------
    {
        const float *v0 = reinterpret_cast<const float *>(pk + LAYER_W_BYTES);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const float bv = v0[wn * 96 + t * 32 + r];
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[mi][t][e] = bv;
        }
        gemm_bf16<3>(tile, pk, lane, wn, acc);
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int e = 0; e < 16; e++) X[mi][t][e] = acc[mi][t][e] > 0.0f ? acc[mi][t][e] : 0.0f;
        __syncthreads();            // every wave has finished reading the tile
        store_slab(tile, part_s, part_q, X, lane, wn);
        __syncthreads();
    }

    // ---- residual blocks: 2 LN-folded GEMMs each -----------------------------------------------------------------
    for (int l = 0; l < 2 * n_blocks; l++) {
        const unsigned char *lw = pk + (long long)(1 + l) * LAYER_BYTES;
        const float *v0 = reinterpret_cast<const float *>(lw + LAYER_W_BYTES), *v1 = v0 + HID;
        if (tid < RM) {             // finalise the statistics of the tile just written (biased variance, eps 1e-5)
            const float s = ((part_s[tid] + part_s[RM + tid]) + part_s[2 * RM + tid]) + part_s[3 * RM + tid];
            const float q = ((part_q[tid] + part_q[RM + tid]) + part_q[2 * RM + tid]) + part_q[3 * RM + tid];
            const float mean = s / 384.0f;
            const float var = fmaxf(q / 384.0f - mean * mean, 0.0f);
            st_mean[tid] = mean;
            st_rstd[tid] = 1.0f / sqrtf(var + 1e-5f);
        }
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[mi][t][e] = 0.0f;
        gemm_bf16<3>(tile, lw, lane, wn, acc);
        __syncthreads();            // tile fully consumed; st_mean / st_rstd visible
        const bool second = l & 1;
        float c[3], cs[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            c[t] = v0[wn * 96 + t * 32 + r];
            cs[t] = v1[wn * 96 + t * 32 + r];
        }
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const f32x4 mean = *reinterpret_cast<const f32x4 *>(st_mean + mi * 32 + 8 * qd + 4 * h);
                const f32x4 rstd = *reinterpret_cast<const f32x4 *>(st_rstd + mi * 32 + 8 * qd + 4 * h);
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int e = 4 * qd + i;
                        float v = __builtin_fmaf(rstd[i], __builtin_fmaf(-mean[i], cs[t], acc[mi][t][e]), c[t]);
                        v = v + (second ? X[mi][t][e] : 0.0f);
                        v = v > 0.0f ? v : 0.0f;
                        X[mi][t][e] = second ? v : X[mi][t][e];
                        acc[mi][t][e] = v;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        store_slab(tile, part_s, part_q, acc, lane, wn);
        __syncthreads();
    }

    // ---- output_proj + L2 normalise ------------------------------------------------------------------------------
    {
        const unsigned char *lw = pk + (long long)(1 + 2 * n_blocks) * LAYER_BYTES;
        const float *v0 = reinterpret_cast<const float *>(lw + OUT_W_BYTES);
        f32x16 o[2][1];
        const float bv = v0[wn * 32 + r];
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) o[mi][0][e] = bv;
        gemm_bf16<1>(tile, lw, lane, wn, o);
        float s[32];
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) s[mi * 16 + e] = o[mi][0][e] * o[mi][0][e];
        const int row = (r >> 4) * 32 + crow(r & 15, h);
        part_s[wn * RM + row] = reduce_scatter32(s, r);
        __syncthreads();
        if (tid < RM) {
            const float ss = ((part_s[tid] + part_s[RM + tid]) + part_s[2 * RM + tid]) + part_s[3 * RM + tid];
            st_rstd[tid] = fmaxf(sqrtf(ss), 1e-12f);     // F.normalize denominator
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rw = mi * 32 + crow(e, h);
                if (R0 + rw < rows) desc[(R0 + rw) * SSLAM_D + wn * 32 + r] = o[mi][0][e] / st_rstd[rw];
            }
    }
}


unsigned short host_bf16r(float v) {
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
float host_bf16_value(unsigned short b) {
    const unsigned u = (unsigned)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// one GEMM layer: W (n_out, 384) [optionally scaled per k by gamma] -> fragment-ordered bf16, v0 = bias (+ W beta), v1 = colsum
void pack_layer(const float *W, const float *bias, const float *gamma, const float *beta, int n_out, unsigned char *dst) {
    unsigned short *wq = reinterpret_cast<unsigned short *>(dst);
    float *v0 = reinterpret_cast<float *>(dst + (long long)n_out * HID * 2), *v1 = v0 + n_out;
    for (int n = 0; n < n_out; n++) {
        double cs = 0.0, wb = 0.0;
        for (int k = 0; k < HID; k++) {
            const float wg = gamma ? W[(long long)n * HID + k] * gamma[k] : W[(long long)n * HID + k];
            const unsigned short b = host_bf16r(wg);
            const int ks = k / 16, hh = (k % 16) / 8, j = k % 8;
            wq[((((long long)ks * (n_out / 32) + n / 32) * 2 + hh) * 32 + n % 32) * 8 + j] = b;
            cs += (double)host_bf16_value(b);
            if (beta) wb += (double)W[(long long)n * HID + k] * (double)beta[k];
        }
        v0[n] = (float)((double)bias[n] + wb);
        v1[n] = gamma ? (float)cs : 0.0f;
    }
}

}  // namespace

extern "C" long long sslam_refiner_bf16_bytes(int n_blocks) {
    if (n_blocks < 0 || n_blocks > 8) return -1;
    return (1 + 2LL * n_blocks) * LAYER_BYTES + OUT_BYTES;
}

// w: the same 4 + 8*n_blocks host pointers as sslam_refiner_pack_host (state_dict order)
extern "C" int sslam_refiner_pack_bf16_host(const float *const *w, int n_blocks, void *out) {
    if (!w || !out || n_blocks < 0 || n_blocks > 8) return SSLAM_E_INVALID;
    unsigned char *o = (unsigned char *)out;
    pack_layer(w[0], w[1], nullptr, nullptr, HID, o);
    for (int b = 0; b < n_blocks; b++) {
        const float *const *p = w + 2 + 8 * b;   // norm1.w, norm1.b, fc1.w, fc1.b, norm2.w, norm2.b, fc2.w, fc2.b
        pack_layer(p[2], p[3], p[0], p[1], HID, o + (1 + 2LL * b) * LAYER_BYTES);
        pack_layer(p[6], p[7], p[4], p[5], HID, o + (2 + 2LL * b) * LAYER_BYTES);
    }
    const float *const *po = w + 2 + 8 * n_blocks;
    pack_layer(po[0], po[1], nullptr, nullptr, SSLAM_D, o + (1 + 2LL * n_blocks) * LAYER_BYTES);
    return SSLAM_OK;
}

static int launch_bf16(const float *feat, int G, const float *kp_xy, int K, const float *x_in, long long rows, const void *packed,
                       int n_blocks, float *desc, void *stream) {
    if (n_blocks < 0 || n_blocks > 8) return SSLAM_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((rows + RM - 1) / RM);
    hipLaunchKernelGGL(refine_bf16_kernel, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, feat, G, kp_xy, K, x_in, rows,
                       (const unsigned char *)packed, n_blocks, desc);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

extern "C" int sslam_refine_bf16(const float *x, long long rows, const void *packed_bf16, int n_blocks, float *desc, void *stream) {
    if (!x || !packed_bf16 || !desc || rows <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)x | (uintptr_t)packed_bf16) & 15) return SSLAM_E_INVALID;
    return launch_bf16(nullptr, 0, nullptr, 1, x, rows, packed_bf16, n_blocks, desc, stream);
}

extern "C" int sslam_gather_refine_bf16(const float *feat, int n_frames, int G, const float *kp_xy, int K, const void *packed_bf16,
                                        int n_blocks, float *desc, void *stream) {
    if (!feat || !kp_xy || !packed_bf16 || !desc || n_frames <= 0 || G <= 1 || K <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)feat | (uintptr_t)packed_bf16) & 15) return SSLAM_E_INVALID;
    return launch_bf16(feat, G, kp_xy, K, nullptr, (long long)n_frames * K, packed_bf16, n_blocks, desc, stream);
}

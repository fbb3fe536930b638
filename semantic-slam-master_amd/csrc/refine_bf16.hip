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
    // address arithmetic in the k loop (every non-MFMA instruction costs matrix-pipe issue time)
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


// ---------------------------------------------------------------------------------------------------------------------
// Row-resident form (round 4): 96 rows x 12 waves.  The column-slab kernel above streams the whole 1.57 MB of weights
// per 64 rows (64 B/clk per CU at the full matrix rate: more than an XCD's L2 delivers) and runs every phase at two waves
// per SIMD.  Here a workgroup owns 96 rows and wave w owns output columns [32 w, 32 w + 32) of every hidden layer for ALL
// 96 rows: its weight stream is private (one 1 KB fragment per k-step straight from L2 into a register ring: 43 B/clk per
// CU at the full rate, no LDS ring, nothing to synchronise), three waves share a SIMD, and the product is evaluated
// TRANSPOSED (weights = A operand, activation tile = B operand), so a lane owns one activation row: the LayerNorm fold, the
// residual and the row statistics are lane-local (one xor-32, then the 12 waves through LDS), the tile is written in 8-byte
// pieces, and the residual stream (3 x 16 fp32 registers) never leaves the lane.
// Optional phase timers (probe builds only, -DSSLAM_CLOCK_PROBE; read with tools/refine_bf16_probe.py): waves 0, 5 and 11 of
// the first 4096 workgroups record their lifetime and the shader-clock cycles spent in the gather, the GEMM loops, the
// epilogues and at barriers.  Compiled out of the product build.
#ifdef SSLAM_CLOCK_PROBE
__device__ unsigned long long g_probe_refbf[3 * 8 * 4096];
#define RB_PROBE_BEGIN() const unsigned long long pr_t0 = clock64(); unsigned long long pr_q = 0, pr_acc[6] = {0, 0, 0, 0, 0, 0}
#define RB_PROBE(slot, stmt) { pr_q = clock64(); stmt; pr_acc[slot] += clock64() - pr_q; }
#define RB_PROBE_END()                                                                                                   \
    if ((threadIdx.x == 0 || threadIdx.x == 320 || threadIdx.x == 704) && blockIdx.x < 4096) {                           \
        const int ws_ = threadIdx.x == 0 ? 0 : (threadIdx.x == 320 ? 1 : 2);                                              \
        unsigned long long *o_ = g_probe_refbf + (ws_ * 4096 + blockIdx.x) * 8;                                           \
        o_[0] = clock64() - pr_t0;                                                                                        \
        for (int i_ = 0; i_ < 6; i_++) o_[1 + i_] = pr_acc[i_];                                                           \
    }
#else
#define RB_PROBE_BEGIN()
#define RB_PROBE(slot, stmt) { stmt; }
#define RB_PROBE_END()
#endif
constexpr int RR = 96, RWAVES = 12, RTHR = 64 * RWAVES;
#ifndef REFBF_RING
#define REFBF_RING 4
#endif
constexpr int ROFF_PART = RR * LDT * 2;                               // bytes: tile | part_s | part_q | st_mean | st_rstd | consts
constexpr int ROFF_STAT = ROFF_PART + 2 * RWAVES * RR * 4;
constexpr int ROFF_CST = ROFF_STAT + 2 * RR * 4;

// acc[rb] += W[32 tile .. +32][:] . X[32 rb .. +32][:]^T over K = 384: one weight fragment (global, ring of 4) and NB
// activation fragments (LDS) per k-step; W in fragment order [ks][n/32][64 lanes][8]
template <int NB, int TILES>
__device__ __forceinline__ void gemm_rows(const bf16 *tile, const unsigned char *__restrict__ w, int lane, int wtile, int row0,
                                          f32x16 (&acc)[NB]) {
    const int r = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(w), 0, KS * TILES * 1024, 0x00020000);
    const int loff = (wtile * 64 + lane) * 16;
    const bf16 *B = tile + (row0 + r) * LDT + 8 * h;
    // ring of RING weight fragments (RING - 1 loads in flight: the L2 round trip under load is several k-steps long)
    constexpr int RING = REFBF_RING;
    bf16x8 q[RING];
#define LOADW(dst, ks) dst = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, (ks) * TILES * 1024, 0));
#define STEPR(cur, ks)                                                                                  \
    {                                                                                                   \
        bf16x8 bf[NB];                                                                                  \
        _Pragma("unroll") for (int rb = 0; rb < NB; rb++)                                               \
            bf[rb] = *reinterpret_cast<const bf16x8 *>(B + rb * 32 * LDT + (ks) * 16);                   \
        _Pragma("unroll") for (int rb = 0; rb < NB; rb++)                                               \
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, bf[rb], acc[rb], 0, 0, 0);           \
    }
#pragma unroll
    for (int i = 0; i < RING - 1; i++) LOADW(q[i], i);
    static_assert(KS % RING == 0, "the k loop is unrolled by the ring depth");
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += RING) {
#pragma unroll
        for (int i = 0; i < RING; i++) {
            // unconditional (clamped): the compiler keeps an exact count of the loads in flight
            LOADW(q[(i + RING - 1) % RING], min(ks + i + RING - 1, KS - 1));
            __builtin_amdgcn_sched_barrier(0);
            STEPR(q[i], ks + i);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef LOADW
#undef STEPR
}

__global__ __launch_bounds__(RTHR) void refine_bf16_rows_kernel(const float *__restrict__ feat, int G, const float *__restrict__ kp_xy,
                                                                 int K, const float *__restrict__ x_in, long long rows,
                                                                 const unsigned char *__restrict__ pk, int n_blocks,
                                                                 float *__restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    bf16 *tile = reinterpret_cast<bf16 *>(rsm);
    float *part_s = reinterpret_cast<float *>(rsm + ROFF_PART), *part_q = part_s + RWAVES * RR;
    float *st_mean = reinterpret_cast<float *>(rsm + ROFF_STAT), *st_rstd = st_mean + RR;
    float *cst = reinterpret_cast<float *>(rsm + ROFF_CST);             // [1 + 2 n_blocks][v0 | v1][384]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    long long R0;
    {   // XCD-aware order (as above): every XCD gets one contiguous range of row tiles
        const int n_tiles = gridDim.x, b = blockIdx.x, q = n_tiles / 8, rem = n_tiles % 8, x = b % 8;
        R0 = (long long)((x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8) * RR;
    }
    const int n_layers = 1 + 2 * n_blocks;
    RB_PROBE_BEGIN();
    // ---- the per-column constants of every hidden layer (bias / b + W beta, column sums) -> LDS ------------------------------
    for (int i = tid; i < n_layers * (2 * HID / 4); i += RTHR) {
        const int l = i / (2 * HID / 4), j = i % (2 * HID / 4);
        *reinterpret_cast<float4 *>(cst + l * 2 * HID + 4 * j) =
            *(reinterpret_cast<const float4 *>(pk + (long long)l * LAYER_BYTES + LAYER_W_BYTES) + j);
    }
    // ---- phase 0: activation tile <- bf16(gathered features | x_in rows): 8 threads per row, 48 channels each ----------------
    {
        const int row = tid >> 3, part = tid & 7;
        long long R = R0 + row;
        if (R > rows - 1) R = rows - 1;
        bf16 *dst = tile + row * LDT + part * 48;
        if (feat) {
            const long long f = R / K;
            const Taps t = make_taps(feat + f * G * G * SSLAM_C, G, kp_xy[2 * R], kp_xy[2 * R + 1]);
#pragma unroll 3
            for (int j = 0; j < 6; j++) {
                const int c0 = part * 48 + 8 * j;
                const float4 lo = blend4(t, c0), hi = blend4(t, c0 + 4);
                u32x4 o;
                o[0] = pack_bf16(lo.x, lo.y); o[1] = pack_bf16(lo.z, lo.w); o[2] = pack_bf16(hi.x, hi.y); o[3] = pack_bf16(hi.z, hi.w);
                *reinterpret_cast<u32x4 *>(dst + 8 * j) = o;
            }
        } else {
            const float *src = x_in + R * SSLAM_C + part * 48;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const float4 lo = *reinterpret_cast<const float4 *>(src + 8 * j), hi = *reinterpret_cast<const float4 *>(src + 8 * j + 4);
                u32x4 o;
                o[0] = pack_bf16(lo.x, lo.y); o[1] = pack_bf16(lo.z, lo.w); o[2] = pack_bf16(hi.x, hi.y); o[3] = pack_bf16(hi.z, hi.w);
                *reinterpret_cast<u32x4 *>(dst + 8 * j) = o;
            }
        }
    }
#ifdef SSLAM_CLOCK_PROBE
    pr_acc[0] = clock64() - pr_t0;       // gather (+ constants)
#endif
    RB_PROBE(4, __syncthreads();)

    // register e of acc[rb], lane (r, h): output column 32 wv + 8 (e >> 2) + 4 h + (e & 3) of activation row 32 rb + r
    f32x16 X[3], acc[3];
    const int ncol = 32 * wv + 4 * h;                 // + 8 q + i
    // epilogue of a hidden layer: fold / residual / ReLU on the lane's rows, row statistics, bf16 tile, fp32 residual
    auto epilogue = [&](int l, bool fold, bool second, bool keep) {
#ifdef REFBF_EXP_NOEPI
        return;       // experiment builds only (tools/refine_bf16_bench.py variants): GEMM phases alone, results invalid
#endif
        // the LDS addresses below are recomputed here from one opaque copy of the lane's row: left to itself hipcc hoists ~40
        // loop-invariant address registers out of the layer loop and spills them (168 registers per lane at three waves per SIMD)
        int rr = r;
        asm volatile("" : "+v"(rr));
        const float *c0 = cst + l * 2 * HID + ncol;
        bf16 *trow = tile + rr * LDT + ncol;
        float mean[3], rstd[3], sp[3], sq[3];
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            mean[rb] = fold ? st_mean[32 * rb + rr] : 0.0f;
            rstd[rb] = fold ? st_rstd[32 * rb + rr] : 1.0f;
            sp[rb] = sq[rb] = 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {               // four output columns at a time: bounded temporaries (168 registers per lane)
            const f32x4 cc = *reinterpret_cast<const f32x4 *>(c0 + 8 * q), cs = *reinterpret_cast<const f32x4 *>(c0 + HID + 8 * q);
#pragma unroll
            for (int rb = 0; rb < 3; rb++) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int e = 4 * q + i;
                    float t = fold ? __builtin_fmaf(rstd[rb], __builtin_fmaf(-mean[rb], cs[i], acc[rb][e]), cc[i]) : acc[rb][e];
                    if (second) t = t + X[rb][e];
                    t = t > 0.0f ? t : 0.0f;
                    if (keep) X[rb][e] = t;
                    v[i] = t;
                    sp[rb] += t;
                    sq[rb] = __builtin_fmaf(t, t, sq[rb]);
                }
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 o;
                o[0] = pack_bf16(v[0], v[1]);
                o[1] = pack_bf16(v[2], v[3]);
                *reinterpret_cast<u32x2 *>(trow + 32 * rb * LDT + 8 * q) = o;
            }
        }
        float *prow = part_s + wv * RR + rr;
#pragma unroll
        for (int rb = 0; rb < 3; rb++) {
            const float s2 = sp[rb] + __shfl_xor(sp[rb], 32), q2 = sq[rb] + __shfl_xor(sq[rb], 32);
            if (h == 0) {
                prow[32 * rb] = s2;
                prow[RWAVES * RR + 32 * rb] = q2;
            }
        }
    };

    // ---- input_proj + ReLU ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(cst + ncol + 8 * q);
#pragma unroll
        for (int rb = 0; rb < 3; rb++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[rb][4 * q + i] = bv[i];
    }
    RB_PROBE(1, (gemm_rows<3, 12>(tile, pk, lane, wv, 0, acc));)
    RB_PROBE(4, __syncthreads();)   // every wave has finished reading the tile
    RB_PROBE(2, epilogue(0, false, false, true);)
    RB_PROBE(4, __syncthreads();)

    // ---- residual blocks: 2 LN-folded GEMMs each -----------------------------------------------------------------------------
    for (int l = 0; l < 2 * n_blocks; l++) {
        if (tid < RR) {             // finalise the statistics of the tile just written (biased variance, eps 1e-5)
            int t2 = tid;
            asm volatile("" : "+v"(t2));          // one base register + immediate offsets, recomputed per layer (see epilogue)
            const float *ps = part_s + t2;
            float s = 0.0f, q = 0.0f;
#pragma unroll
            for (int w2 = 0; w2 < RWAVES; w2++) {
                s += ps[w2 * RR];
                q += ps[(RWAVES + w2) * RR];
            }
            const float mean = s / 384.0f;
            const float var = fmaxf(q / 384.0f - mean * mean, 0.0f);
            float *sm = st_mean + t2;
            sm[0] = mean;
            sm[RR] = 1.0f / sqrtf(var + 1e-5f);
        }
#pragma unroll
        for (int rb = 0; rb < 3; rb++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[rb][e] = 0.0f;
        RB_PROBE(1, (gemm_rows<3, 12>(tile, pk + (long long)(1 + l) * LAYER_BYTES, lane, wv, 0, acc));)
        RB_PROBE(4, __syncthreads();)   // tile fully consumed; st_mean / st_rstd visible
        if (l & 1)
            RB_PROBE(2, epilogue(1 + l, true, true, true);)
        else
            RB_PROBE(2, epilogue(1 + l, true, false, false);)
        RB_PROBE(4, __syncthreads();)
    }

    // ---- output_proj + L2 normalise: 3 row blocks x 4 column tiles = one 32 x 32 tile per wave ------------------------------------
    {
        const unsigned char *lw = pk + (long long)n_layers * LAYER_BYTES;
        const float *v0 = reinterpret_cast<const float *>(lw + OUT_W_BYTES);
        const int rb = wv >> 2, nt = wv & 3;
        f32x16 o[1];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(v0 + 32 * nt + 4 * h + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; i++) o[0][4 * q + i] = bv[i];
        }
        RB_PROBE(3, (gemm_rows<1, 4>(tile, lw, lane, nt, 32 * rb, o));)
        float ss = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; e++) ss = __builtin_fmaf(o[0][e], o[0][e], ss);
        ss += __shfl_xor(ss, 32);
        const int row = 32 * rb + r;
        if (h == 0) part_s[nt * RR + row] = ss;           // part_s was last read before the barrier that ended the last layer
        __syncthreads();
        const float tot = ((part_s[row] + part_s[RR + row]) + part_s[2 * RR + row]) + part_s[3 * RR + row];
        const float den = fmaxf(sqrtf(tot), 1e-12f);      // F.normalize denominator
        if (R0 + row < rows) {
            float *dst = desc + (R0 + row) * SSLAM_D + 32 * nt + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 t;
                t.x = o[0][4 * q] / den;
                t.y = o[0][4 * q + 1] / den;
                t.z = o[0][4 * q + 2] / den;
                t.w = o[0][4 * q + 3] / den;
                *reinterpret_cast<float4 *>(dst + 8 * q) = t;
            }
        }
    }
    RB_PROBE_END();
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

#ifdef SSLAM_CLOCK_PROBE
extern "C" int sslam_probe_refine_bf16(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe_refbf), sizeof(unsigned long long) * 3 * 8 * 4096) == hipSuccess ? 0 : -3;
}
#endif

static int launch_bf16(const float *feat, int G, const float *kp_xy, int K, const float *x_in, long long rows, const void *packed,
                       int n_blocks, float *desc, void *stream) {
    if (n_blocks < 0 || n_blocks > 8) return SSLAM_E_UNSUPPORTED;
    if (sslam_knob(KNOB_REFBF_FORM, 0) != 1) {      // test-only A/B knob: 1 = the round-1 column-slab kernel (64 rows, 4 waves)
        const unsigned grid = (unsigned)((rows + RR - 1) / RR);
        const int lds = ROFF_CST + (1 + 2 * n_blocks) * 2 * HID * 4;
        hipLaunchKernelGGL(refine_bf16_rows_kernel, dim3(grid), dim3(RTHR), lds, (hipStream_t)stream, feat, G, kp_xy, K, x_in, rows,
                           (const unsigned char *)packed, n_blocks, desc);
        SSLAM_CHECK_LAUNCH();
        return SSLAM_OK;
    }
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

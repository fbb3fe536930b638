// refine.hip - A6 + A7: bilinear feature gather and the descriptor MLP, fused per 64-row tile.
// Replaces DinoBackbone.extract_at_keypoints (reference semantic-slam/models/dino_backbone.py:114-152) and
// DescriptorRefiner.forward / ResidualBlock.forward (semantic-slam/models/descriptor_refiner.py:58-126).
//
// One workgroup (8 waves) owns 64 keypoint rows for the WHOLE chain: gather -> input_proj+ReLU -> n_blocks x
// {LN, fc1, ReLU, LN, fc2, +identity, ReLU} -> output_proj -> L2 normalise.  The 64x384 activation tile lives in LDS
// (KP8 order, 388-float rows) and is the MFMA A operand directly; the residual identity stays in registers in the
// MFMA C layout (each wave keeps the same 32x96 output sub-tile in every layer).  The pre-packed weights (3.17 MB,
// L2-resident, shared by all workgroups) are NOT staged through LDS: they are stored in MFMA-fragment order
// ([k/8][n][8 floats KP8]), so a wave's B fragment is one fully coalesced 1 KB global load per tile, prefetched two
// k-groups ahead into registers.  A layer's GEMM therefore has no barrier at all - waves drift freely and keep the
// matrix pipe busy; barriers only separate the layers (activation tile hand-over).
// All contractions are v_mfma_f32_32x32x2_f32 chains in increasing k from the bias: bit-identical to the oracle.
//
// Roofline: MFMA-bound, 1 572 864 FLOP per row (786.4 MFLOP per 500-keypoint frame) against 1.5 KB in / 0.5 KB out.
#include "common.h"
#include "gather_taps.h"

namespace {

#ifndef SSLAM_REFINE_WMR
#define SSLAM_REFINE_WMR 1
#endif
constexpr int WMR = SSLAM_REFINE_WMR;   // waves along M: 1 -> 32-row workgroups of 4 waves (3 co-resident per CU, so the
                                        // gather / LayerNorm phases of one overlap the GEMMs of the others); 2 -> 64 rows
constexpr int RM = 32 * WMR;        // rows per workgroup
constexpr int NTHR = 256 * WMR;
constexpr int HID = SSLAM_HID;      // 384
constexpr int LDH = HID + 4;        // activation row stride (floats)
constexpr int NKG = HID / 8;        // 48 k-groups of 8 per layer
constexpr int H_FLOATS = RM * LDH;
constexpr int SCRATCH_FLOATS = RM * 4;
constexpr int SMEM_FLOATS = H_FLOATS + SCRATCH_FLOATS;   // 100 352 B

struct RefArgs {
    const float *packed;
    sslam_refiner_layout_t lay;
};

// acc[t] (32 rows x 32 cols each, t-th N tile of this wave) = bias + H(64 x 384) . W^T, one fma chain per output in
// increasing k.  B fragments come straight from global memory (fragment-ordered packed weights), ring of 3 k-groups.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void gemm_lds(const float *H, __amdgpu_buffer_rsrc_t wrs, int w_off, const float *__restrict__ bias,
                                         int tid, f32x16 (&acc)[NT]) {
    constexpr int N = 128 * NT;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float bv = bias[wn * 32 * NT + t * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = bv;
    }
    // lane's B fragment of k-group g, tile t: 16 B at ((g*N + n)*8 + 4h) floats, n = wn*32*NT + t*32 + r.  Buffer loads:
    // address = descriptor base + 32-bit lane offset (constant) + SCALAR offset of (layer, k-group, tile), so the k loop
    // carries no vector address arithmetic at all - every non-MFMA instruction costs matrix-pipe issue time
    const int loff = ((wn * 32 * NT + r) * 2 + h) * 16;
    const float *A = H + (wm * 32 + r) * LDH + 4 * h;
    f32x4 b0[NT], b1[NT], b2[NT];
#define LOAD_B(dst, g)                                                               \
    _Pragma("unroll") for (int t = 0; t < NT; t++)                                   \
        dst[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, w_off + ((g) * N + t * 32) * 32, 0));
#define STEP(cur, g)                                                                  \
    {                                                                                 \
        const f32x4 a = an;                                                           \
        an = *reinterpret_cast<const f32x4 *>(A + 8 * (((g) + 1) < NKG ? (g) + 1 : (g))); /* next k-group's A */ \
        _Pragma("unroll") for (int st = 0; st < 4; st++)                              \
            _Pragma("unroll") for (int t = 0; t < NT; t++) acc[t] = mfma32(a[st], cur[t][st], acc[t]); \
    }
    LOAD_B(b0, 0);
    LOAD_B(b1, 1);
    f32x4 an = *reinterpret_cast<const f32x4 *>(A);
#pragma unroll 1
    for (int g = 0; g < NKG; g += 3) {
        // the refills are UNCONDITIONAL (clamped index, a redundant reload in the last trip): with a branch around them the
        // compiler cannot count the loads in flight and falls back to s_waitcnt vmcnt(0) before the third step of every
        // trip, which drains the whole prefetch ring
        // (sched_barrier: program order = issue order, or the scheduler sinks the loads next to their uses)
        LOAD_B(b2, g + 2);
        __builtin_amdgcn_sched_barrier(0);
        STEP(b0, g);
        __builtin_amdgcn_sched_barrier(0);
        LOAD_B(b0, min(g + 3, NKG - 1));
        __builtin_amdgcn_sched_barrier(0);
        STEP(b1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        LOAD_B(b1, min(g + 4, NKG - 1));
        __builtin_amdgcn_sched_barrier(0);
        STEP(b2, g + 2);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef LOAD_B
#undef STEP
}

// write this wave's C-layout tiles back into the activation tile (KP8 positions)
template <int NT>
__device__ __forceinline__ void store_tile(float *H, int tid, const f32x16 (&v)[NT]) {
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int col = kp8(wn * 32 * NT + t * 32 + r);
#pragma unroll
        for (int e = 0; e < 16; e++) H[(wm * 32 + crow(e, h)) * LDH + col] = v[t][e];
    }
}

// LayerNorm(384) in place on the 64 rows (wave w: rows 8w..8w+7; lane j < 48: elements 8j..8j+7); canonical order:
// 8 sequential adds per lane, 64-lane butterfly (lanes >= 48 hold 0), two passes (oracle layernorm384)
__device__ __forceinline__ void layernorm_rows(float *H, const float *__restrict__ gam, const float *__restrict__ bet,
                                               int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const bool act = lane < 48;
    const int j = act ? lane : 0;
    float gm[8], bt[8];
    {
        const float4 g0 = *reinterpret_cast<const float4 *>(gam + 8 * j), g1 = *reinterpret_cast<const float4 *>(gam + 8 * j + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(bet + 8 * j), b1 = *reinterpret_cast<const float4 *>(bet + 8 * j + 4);
        gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
        bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
    }
    // All 8 rows of the wave go through every butterfly step TOGETHER: the 12 cross-lane steps per row are LDS-path
    // round trips (ds_bpermute, several hundred cycles each while the other waves stream MFMA operands), and row by row
    // they added up to 96 serial round trips per call - a quarter of the workgroup's lifetime.  Per row the operations
    // and their order are unchanged (oracle layernorm384).
    constexpr int NR = RM / (NTHR / 64);      // rows per wave
    float x[NR][8], s[NR];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const float *p = H + (wave * NR + rr) * LDH + 8 * j;
        const float4 ev = *reinterpret_cast<const float4 *>(p), od = *reinterpret_cast<const float4 *>(p + 4);
        x[rr][0] = ev.x; x[rr][1] = od.x; x[rr][2] = ev.y; x[rr][3] = od.y;
        x[rr][4] = ev.z; x[rr][5] = od.z; x[rr][6] = ev.w; x[rr][7] = od.w;
        float t = x[rr][0];
#pragma unroll
        for (int i = 1; i < 8; i++) t = t + x[rr][i];
        s[rr] = act ? t : 0.0f;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int rr = 0; rr < NR; rr++) s[rr] = s[rr] + __shfl_xor(s[rr], m);
    float mean[NR];
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        mean[rr] = s[rr] / 384.0f;
        float s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float d = x[rr][i] - mean[rr];
            s2 = __builtin_fmaf(d, d, s2);
        }
        s[rr] = act ? s2 : 0.0f;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int rr = 0; rr < NR; rr++) s[rr] = s[rr] + __shfl_xor(s[rr], m);
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const float var = s[rr] / 384.0f;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; i++) y[i] = __builtin_fmaf((x[rr][i] - mean[rr]) * rstd, gm[i], bt[i]);
        if (act) {
            float *p = H + (wave * NR + rr) * LDH + 8 * j;
            *reinterpret_cast<float4 *>(p) = make_float4(y[0], y[2], y[4], y[6]);
            *reinterpret_cast<float4 *>(p + 4) = make_float4(y[1], y[3], y[5], y[7]);
        }
    }
}

// Optional phase timers (build with -DSSLAM_CLOCK_PROBE, read with tools/clock_probe.py): wave 0 of the first 4096 workgroups
// records its lifetime and the shader-clock cycles it spent in the gather, the GEMM loops, the LayerNorm calls, the tile
// stores and waiting at barriers.  Compiled out of the product build.
#ifdef SSLAM_CLOCK_PROBE
__device__ unsigned long long g_probe_refine[8 * 4096];
#define PROBE_BEGIN() const unsigned long long pr_t0 = clock64(); unsigned long long pr_q = 0, pr_acc[6] = {0, 0, 0, 0, 0, 0}
#define PROBE(slot, stmt) { pr_q = clock64(); stmt; pr_acc[slot] += clock64() - pr_q; }
#define PROBE_END()                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                            \
        g_probe_refine[8 * blockIdx.x] = clock64() - pr_t0;                                                 \
        for (int i_ = 0; i_ < 6; i_++) g_probe_refine[8 * blockIdx.x + 1 + i_] = pr_acc[i_];                \
    }
#else
#define PROBE_BEGIN()
#define PROBE(slot, stmt) { stmt; }
#define PROBE_END()
#endif
enum { PR_GATHER = 0, PR_GEMM = 1, PR_LN = 2, PR_STORE = 3, PR_BARRIER = 4 };

__global__ __launch_bounds__(NTHR, WMR == 1 ? 3 : 2) void gather_refine_kernel(const float *__restrict__ feat, int G,
                                                             const float *__restrict__ kp_xy, int K,
                                                             const float *__restrict__ x_in, long long rows,
                                                             RefArgs args, float *__restrict__ desc) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float *H = smem, *scratch = smem + H_FLOATS;
    const int tid = threadIdx.x;
    PROBE_BEGIN();
    // XCD-aware order: workgroup b runs on XCD b % 8; give every XCD one contiguous range of row tiles so that the ~8
    // tiles gathering from one frame's feature map share that XCD's L2 instead of fetching the frame into all eight
    long long R0;
    {
        const int n_tiles = gridDim.x, b = blockIdx.x, q = n_tiles / 8, rem = n_tiles % 8, x = b % 8;
        R0 = (long long)((x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8) * RM;
    }
    const float *pk = args.packed;
    const sslam_refiner_layout_t &L = args.lay;
    // buffer descriptor over the packed weights (raw buffer, 32-bit element format word as on gfx90a/gfx94x/gfx950)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pk), 0, (int)(L.total * 4), 0x00020000);

    // ---- phase 0: fill the activation tile (gathered features, or rows of x_in) --------------------------------
    {
        const int row = tid >> 3, part = tid & 7;
        long long R = R0 + row;
        if (R > rows - 1) R = rows - 1;
        float *dst = H + row * LDH;
        if (feat) {
            const long long f = R / K;
            const Taps t = make_taps(feat + f * G * G * SSLAM_C, G, kp_xy[2 * R], kp_xy[2 * R + 1]);
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int c0 = 8 * (part * 6 + j);
                float4 ev, od;
                kp8_split(blend4(t, c0), blend4(t, c0 + 4), ev, od);
                *reinterpret_cast<float4 *>(dst + c0) = ev;
                *reinterpret_cast<float4 *>(dst + c0 + 4) = od;
            }
        } else {
            const float *src = x_in + R * SSLAM_C;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int c0 = 8 * (part * 6 + j);
                float4 ev, od;
                kp8_split(*reinterpret_cast<const float4 *>(src + c0), *reinterpret_cast<const float4 *>(src + c0 + 4), ev, od);
                *reinterpret_cast<float4 *>(dst + c0) = ev;
                *reinterpret_cast<float4 *>(dst + c0 + 4) = od;
            }
        }
    }
#ifdef SSLAM_CLOCK_PROBE
    pr_acc[PR_GATHER] = clock64() - pr_t0;
#endif
    PROBE(PR_BARRIER, __syncthreads();)

    // ---- input_proj + ReLU (descriptor_refiner.py:76) -----------------------------------------------------------
    f32x16 X[3], acc[3];
    PROBE(PR_GEMM, gemm_lds<3>(H, wrs, (int)L.in_w * 4, pk + L.in_b, tid, acc);)
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) X[t][e] = acc[t][e] > 0.0f ? acc[t][e] : 0.0f;
    PROBE(PR_BARRIER, __syncthreads();)            // every wave has finished reading the tile
    PROBE(PR_STORE, store_tile<3>(H, tid, X);)
    PROBE(PR_BARRIER, __syncthreads();)

    // ---- residual blocks (descriptor_refiner.py:108-126) --------------------------------------------------------
    for (int b = 0; b < L.n_blocks; b++) {
        PROBE(PR_LN, layernorm_rows(H, pk + L.blk[b][0], pk + L.blk[b][1], tid);)
        PROBE(PR_BARRIER, __syncthreads();)
        PROBE(PR_GEMM, gemm_lds<3>(H, wrs, (int)L.blk[b][2] * 4, pk + L.blk[b][3], tid, acc);)
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[t][e] = acc[t][e] > 0.0f ? acc[t][e] : 0.0f;
        PROBE(PR_BARRIER, __syncthreads();)
        PROBE(PR_STORE, store_tile<3>(H, tid, acc);)
        PROBE(PR_BARRIER, __syncthreads();)
        PROBE(PR_LN, layernorm_rows(H, pk + L.blk[b][4], pk + L.blk[b][5], tid);)
        PROBE(PR_BARRIER, __syncthreads();)
        PROBE(PR_GEMM, gemm_lds<3>(H, wrs, (int)L.blk[b][6] * 4, pk + L.blk[b][7], tid, acc);)
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float v = acc[t][e] + X[t][e];
                X[t][e] = v > 0.0f ? v : 0.0f;
            }
        PROBE(PR_BARRIER, __syncthreads();)
        PROBE(PR_STORE, store_tile<3>(H, tid, X);)
        PROBE(PR_BARRIER, __syncthreads();)
    }

    // ---- output_proj + L2 normalise (:83-86; F.normalize eps 1e-12) --------------------------------------------
    f32x16 o[1];
    PROBE(PR_GEMM, gemm_lds<1>(H, wrs, (int)L.out_w * 4, pk + L.out_b, tid, o);)
    {
        const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
        float *part = scratch;  // [64 rows][4 waves]
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float t = bfly32(o[0][e] * o[0][e]);
            if (r == 0) part[(wm * 32 + crow(e, h)) * 4 + wn] = t;
        }
        PROBE(PR_BARRIER, __syncthreads();)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int row = wm * 32 + crow(e, h);
            const float4 t = *reinterpret_cast<const float4 *>(part + row * 4);
            const float ss = ((t.x + t.y) + t.z) + t.w;
            const float den = fmaxf(sqrtf(ss), 1e-12f);
            if (R0 + row < rows) desc[(R0 + row) * SSLAM_D + wn * 32 + r] = o[0][e] / den;
        }
    }
    PROBE_END();
}

__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ feat, int G, const float *__restrict__ kp_xy,
                                                      int K, long long rows, float *__restrict__ out) {
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;   // one float4 of one row
    const long long R = item / (SSLAM_C / 4);
    if (R >= rows) return;
    const int c0 = (int)(item % (SSLAM_C / 4)) * 4;
    const long long f = R / K;
    const Taps t = make_taps(feat + f * G * G * SSLAM_C, G, kp_xy[2 * R], kp_xy[2 * R + 1]);
    *reinterpret_cast<float4 *>(out + R * SSLAM_C + c0) = blend4(t, c0);
}

int launch_refine(const float *feat, int G, const float *kp_xy, int K, const float *x_in, long long rows,
                  const float *packed, int n_blocks, float *desc, void *stream) {
    RefArgs a;
    a.packed = packed;
    if (sslam_refiner_layout(n_blocks, &a.lay) != SSLAM_OK) return SSLAM_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((rows + RM - 1) / RM);
    hipLaunchKernelGGL(gather_refine_kernel, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, feat, G, kp_xy, K, x_in, rows,
                       a, desc);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

}  // namespace

extern "C" int sslam_refiner_layout(int n_blocks, sslam_refiner_layout_t *L) {
    if (!L || n_blocks < 0 || n_blocks > 8) return SSLAM_E_INVALID;
    long long off = 0;
    auto take = [&](long long n) { const long long o = off; off += n; return o; };
    L->n_blocks = n_blocks;
    L->in_w = take((long long)HID * SSLAM_C);
    L->in_b = take(HID);
    for (int b = 0; b < 8; b++)
        for (int k = 0; k < 8; k++) L->blk[b][k] = 0;
    for (int b = 0; b < n_blocks; b++) {
        L->blk[b][0] = take(HID); L->blk[b][1] = take(HID);
        L->blk[b][2] = take((long long)HID * HID); L->blk[b][3] = take(HID);
        L->blk[b][4] = take(HID); L->blk[b][5] = take(HID);
        L->blk[b][6] = take((long long)HID * HID); L->blk[b][7] = take(HID);
    }
    L->out_w = take((long long)SSLAM_D * HID);
    L->out_b = take(SSLAM_D);
    L->total = off;
    return SSLAM_OK;
}

// w (n_out, k_in) -> [k-group = k/8][n][8 floats in KP8 order]: the MFMA B-fragment order
extern "C" int sslam_pack_linear_host(const float *w, int n_out, int k_in, float *out) {
    if (!w || !out || n_out <= 0 || k_in <= 0 || (k_in % 8)) return SSLAM_E_INVALID;
    for (int n = 0; n < n_out; n++)
        for (int k = 0; k < k_in; k++)
            out[((long long)(k / 8) * n_out + n) * 8 + kp8(k % 8)] = w[(long long)n * k_in + k];
    return SSLAM_OK;
}

extern "C" int sslam_refiner_pack_host(const float *const *w, int n_blocks, float *out) {
    sslam_refiner_layout_t L;
    if (!w || !out || sslam_refiner_layout(n_blocks, &L) != SSLAM_OK) return SSLAM_E_INVALID;
    auto cp = [&](long long off, const float *src, int n) { for (int i = 0; i < n; i++) out[off + i] = src[i]; };
    sslam_pack_linear_host(w[0], HID, SSLAM_C, out + L.in_w);
    cp(L.in_b, w[1], HID);
    for (int b = 0; b < n_blocks; b++) {
        const float *const *p = w + 2 + 8 * b;
        cp(L.blk[b][0], p[0], HID); cp(L.blk[b][1], p[1], HID);
        sslam_pack_linear_host(p[2], HID, HID, out + L.blk[b][2]); cp(L.blk[b][3], p[3], HID);
        cp(L.blk[b][4], p[4], HID); cp(L.blk[b][5], p[5], HID);
        sslam_pack_linear_host(p[6], HID, HID, out + L.blk[b][6]); cp(L.blk[b][7], p[7], HID);
    }
    const float *const *po = w + 2 + 8 * n_blocks;
    sslam_pack_linear_host(po[0], SSLAM_D, HID, out + L.out_w);
    cp(L.out_b, po[1], SSLAM_D);
    return SSLAM_OK;
}

extern "C" int sslam_refine(const float *x, long long rows, const float *packed, int n_blocks, float *desc, void *stream) {
    if (!x || !packed || !desc || rows <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)x | (uintptr_t)packed) & 15) return SSLAM_E_INVALID;
    return launch_refine(nullptr, 0, nullptr, 1, x, rows, packed, n_blocks, desc, stream);
}

extern "C" int sslam_gather_refine(const float *feat, int n_frames, int G, const float *kp_xy, int K, const float *packed,
                                   int n_blocks, float *desc, void *stream) {
    if (!feat || !kp_xy || !packed || !desc || n_frames <= 0 || G <= 1 || K <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)feat | (uintptr_t)packed) & 15) return SSLAM_E_INVALID;
    return launch_refine(feat, G, kp_xy, K, nullptr, (long long)n_frames * K, packed, n_blocks, desc, stream);
}

extern "C" int sslam_gather(const float *feat, int n_frames, int G, const float *kp_xy, int K, float *out, void *stream) {
    if (!feat || !kp_xy || !out || n_frames <= 0 || G <= 1 || K <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)feat | (uintptr_t)out) & 15) return SSLAM_E_INVALID;
    const long long rows = (long long)n_frames * K, items = rows * (SSLAM_C / 4);
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, G, kp_xy,
                       K, rows, out);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

#ifdef SSLAM_CLOCK_PROBE
extern "C" int sslam_probe_refine(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe_refine), sizeof(unsigned long long) * 8 * 4096) == hipSuccess ? 0 : -3;
}
#endif

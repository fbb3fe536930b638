// selector.hip - A3: saliency CNN (conv3x3 384->hs + ReLU + conv1x1 hs->1 + sigmoid) as one fp32-MFMA implicit GEMM.
// Replaces KeypointSelector.forward (reference semantic-slam/models/keypoint_selector.py:45-67).
//
// GEMM view: M = n_frames*G*G cells, N = hs hidden channels, K = 3456 walked in 108 stages of 32 in the canonical
// order k = (chunk*9 + tap)*32 + c, chunk = 32-channel slice (0..11), tap = ky*3 + kx: the nine taps of one channel
// slice are consecutive, so a workgroup re-reads the same (BM + 2G + 2) x 128 B window nine times in a row - L1/L2
// hits instead of nine sweeps over the feature map (which measured 8.8x the algorithmic HBM bytes).
// Workgroup = 8 waves; every wave owns a 64 x (32*NI) sub-tile (2 x NI MFMA 32x32 tiles).  A rows are gathered from
// the NHWC feature map with the tap's spatial shift (zero rows outside the grid are multiplied through, exactly as
// the oracle does) and staged through LDS in the KP8 order (common.h) with a +4 float row pad: one conflict-free
// ds_read_b128 feeds four consecutive v_mfma_f32_32x32x2_f32 steps.  B (the pre-packed weights, MFMA-fragment order
// [stage][k-group][n][8]) is either staged through LDS too or, BDIRECT, loaded straight from L2 into registers
// (1 KB coalesced per wave-instruction).  Global->LDS staging is register double-buffered.
// Tiles are dealt to XCDs in contiguous ranges (blockIdx % 8 selects the range) so that neighbouring tiles, which
// share their halo rows, share an L2.
//
// Roofline: MFMA-bound (fp32 matrix peak 157.3 TFLOP/s).  Algorithmic work: cells * hs * 3456 * 2 FLOP
// (+ hs*2 for the 1x1), i.e. 1 387.7 MFLOP per 28x28 frame at hs = 256.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDT = BK + 4;  // padded LDS row (floats): 144 B = 9 x 16 B -> conflict-free b128 fragment reads
constexpr int NCHUNK = SSLAM_C / BK;
constexpr int NSTAGE = 9 * NCHUNK;

template <int WM, int WN, int NI, bool BDIRECT, int MI = 2>
__global__ __launch_bounds__(512, (BDIRECT && WM == 2) ? 4 : 1) void selector_saliency_kernel(const float *__restrict__ feat, int n_rows, int G,
                                                                 const float *__restrict__ w1p,
                                                                 const float *__restrict__ b1,
                                                                 const float *__restrict__ w2,
                                                                 const float *__restrict__ b2, float *__restrict__ sal,
                                                                 int n_tiles) {
    static_assert(WM * WN == 8, "8 waves");
    static_assert(MI == 2 || (MI == 1 && NI == 1 && BDIRECT), "latency form: one 32 x 32 tile per wave");
    constexpr int BM = 32 * MI * WM;        // MI = 1: the LATENCY form - 32-row workgroups, one MFMA tile per wave, so a stage is
                                            // 16 MFMAs per wave instead of 64 and a frame spreads over 4x as many CUs (25 workgroups
                                            // at G = 28): the 108-stage k chain is the same single fma chain per output (bit-exact)
    constexpr int HS = 32 * NI * WN;
    constexpr int NSLAB = HS / 64;
    constexpr int A_ITEMS = BM * 4 >= 512 ? BM * 4 / 512 : 1;   // 8-float items per thread per stage
    constexpr bool A_PART = BM * 4 < 512;           // fewer items than threads: only the first BM * 4 threads stage A
    constexpr int B_ITEMS = BDIRECT ? 1 : HS * 8 / 512;   // float4 items per thread per stage
    constexpr int STAGE_FLOATS = (BM + (BDIRECT ? 0 : HS)) * LDT;
    constexpr int RED_FLOATS = NSLAB * BM + (MI == 1 ? 4 * 32 * 64 : 0);   // + the odd column waves' products (latency form)
    constexpr int SMEM_FLOATS = 2 * STAGE_FLOATS > RED_FLOATS ? 2 * STAGE_FLOATS : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int cells = G * G;
    // XCD-aware tile mapping (bijective for any n_tiles): blocks b, b+8, b+16.. run on one XCD -> give them
    // consecutive tiles
    int tile;
    {
        const int b = blockIdx.x, q = n_tiles / 8, rem = n_tiles % 8, x = b % 8;
        tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8;
    }
    const long long m0 = (long long)tile * BM;

    // per-thread A rows (fixed for the whole K loop).  Loads are buffer loads: descriptor base + 32-bit lane offset of the
    // row's own cell + a SCALAR offset for (tap, channel chunk) - no per-stage vector address arithmetic (every non-MFMA
    // instruction costs matrix-pipe issue time, DESIGN_HISTORY.md section 9).  A tap outside the grid reads some other cell (or, outside the
    // buffer, zeros from the descriptor's range check); it is zeroed at STORE_STAGE by the precomputed 9-bit validity mask.
    const __amdgpu_buffer_rsrc_t frs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, (int)((unsigned)n_rows * (SSLAM_C * 4u)), 0x00020000);
    int a_row[A_ITEMS], a_kq[A_ITEMS], a_voff[A_ITEMS], a_mask[A_ITEMS];
#pragma unroll
    for (int i = 0; i < A_ITEMS; i++) {
        const int it = A_PART ? (tid < BM * 4 ? tid : 0) : tid + 512 * i;
        a_row[i] = it >> 2;
        a_kq[i] = it & 3;
        const long long m = m0 + a_row[i];
        const bool okr = m < n_rows;
        const long long mm = okr ? m : 0;
        const int cell = (int)(mm % cells), y = cell / G, x = cell % G;
        a_voff[i] = (int)((unsigned)mm * (SSLAM_C * 4u) + a_kq[i] * 32u);
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            mk |= (okr && yy >= 0 && yy < G && xx >= 0 && xx < G) ? (1 << t) : 0;
        }
        a_mask[i] = mk;
    }

    float4 ra_lo[A_ITEMS], ra_hi[A_ITEMS], rb[B_ITEMS];
    bool ra_ok[A_ITEMS];
    int ld_tap = 0, ld_dy = -1, ld_dx = -1, ld_chunk = 0;      // the stage LOAD_STAGE fetches next (stages are fetched in order)

#define LOAD_STAGE(S)                                                                                              \
    {                                                                                                              \
        const int s_ = (S);                                                                                        \
        const int chunk = ld_chunk, tap = ld_tap;       /* == s_ / 9, s_ % 9: carried incrementally (scalar ALU) */      \
        /* the tap shift can be negative and a buffer's scalar offset is unsigned: it goes into the lane offset (one  */ \
        /* add; a wrapped / too large result fails the range check and reads zeros), the chunk into the scalar part */ \
        const int toff_ = (ld_dy * G + ld_dx) * (SSLAM_C * 4), soff_ = chunk * (BK * 4);                             \
        _Pragma("unroll") for (int i = 0; i < A_ITEMS; i++) {                                                      \
            const int vt_ = a_voff[i] + toff_;                                                                     \
            ra_lo[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(frs, vt_, soff_, 0));             \
            ra_hi[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(frs, vt_, soff_ + 16, 0));        \
            ra_ok[i] = (a_mask[i] >> tap) & 1;                                                                     \
        }                                                                                                          \
        if (!BDIRECT) {                                                                                            \
            const float4 *wp = reinterpret_cast<const float4 *>(w1p + (long long)s_ * HS * BK);                    \
            _Pragma("unroll") for (int i = 0; i < B_ITEMS; i++) rb[i] = wp[tid + 512 * i];                         \
        }                                                                                                          \
        /* advance (tap, dy, dx, chunk) to the stage after s_ */                                                   \
        ld_tap++; ld_dx++;                                                                                         \
        if (ld_dx == 2) { ld_dx = -1; ld_dy++; }                                                                   \
        if (ld_tap == 9) { ld_tap = 0; ld_dy = -1; ld_chunk++; }                                                   \
    }
#define STORE_STAGE(BUF)                                                                                           \
    {                                                                                                              \
        float *As_ = smem + (BUF) * STAGE_FLOATS;                                                                  \
        float *Bs_ = As_ + BM * LDT;                                                                               \
        _Pragma("unroll") for (int i = 0; i < A_ITEMS; i++) if (!A_PART || tid < BM * 4) {                         \
            float4 ev, od;                                                                                         \
            const bool k_ = ra_ok[i];                                                                              \
            const float4 lo_ = make_float4(k_ ? ra_lo[i].x : 0.f, k_ ? ra_lo[i].y : 0.f, k_ ? ra_lo[i].z : 0.f, k_ ? ra_lo[i].w : 0.f); \
            const float4 hi_ = make_float4(k_ ? ra_hi[i].x : 0.f, k_ ? ra_hi[i].y : 0.f, k_ ? ra_hi[i].z : 0.f, k_ ? ra_hi[i].w : 0.f); \
            kp8_split(lo_, hi_, ev, od);                                                                           \
            float *d = As_ + a_row[i] * LDT + a_kq[i] * 8;                                                         \
            *reinterpret_cast<float4 *>(d) = ev;                                                                   \
            *reinterpret_cast<float4 *>(d + 4) = od;                                                               \
        }                                                                                                          \
        if (!BDIRECT) {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < B_ITEMS; i++) {                                                  \
                const int q = tid + 512 * i; /* float4 index in [g][n][2] */                                      \
                const int g_ = q / (HS * 2), n_ = (q >> 1) % HS;                                                   \
                *reinterpret_cast<float4 *>(Bs_ + n_ * LDT + 8 * g_ + 4 * (q & 1)) = rb[i];                        \
            }                                                                                                      \
        }                                                                                                          \
    }

    // accumulators start from the conv bias: the fma chain is b1[n] + sum_k a_k * w_k in canonical k order
    f32x16 acc[MI][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        const float bv = b1[wn * 32 * NI + ni * 32 + r];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ni][e] = bv;
    }
    // direct-B: lane's fragment of (stage s, k-group g, tile ni) = 16 B at (((s*4+g)*HS + n)*8 + 4h) floats
    const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(w1p) + ((wn * 32 * NI + r) * 2 + h);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w1p), 0, 9 * SSLAM_C * HS * 4, 0x00020000);
    const int b_voff = ((wn * 32 * NI + r) * 2 + h) * 16;

    // B fragments are one continuous stream over (stage, k-group) in the packed weights: slot g of the ring always holds
    // k-group g of the current stage and is refilled with k-group g of the NEXT stage right after its MFMAs, i.e. every
    // fragment is requested one full stage (>= 4096 matrix-pipe cycles) before it is needed.
    f32x4 bq[BK / 8][NI];
    if (BDIRECT) {
#pragma unroll
        for (int g = 0; g < BK / 8; g++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++) bq[g][ni] = bsrc[((long long)g * HS + ni * 32) * 2];
    }
    LOAD_STAGE(0);
    STORE_STAGE(0);
    __syncthreads();
    for (int s = 0; s < NSTAGE; s++) {
        if (s + 1 < NSTAGE) LOAD_STAGE(s + 1);
        const float *As = smem + (s & 1) * STAGE_FLOATS + (wm * 32 * MI + r) * LDT + 4 * h;
        const float *Bs = smem + (s & 1) * STAGE_FLOATS + BM * LDT + (wn * 32 * NI + r) * LDT + 4 * h;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < BK / 8; g++) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(As + 8 * g);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(As + (MI == 2 ? 32 : 0) * LDT + 8 * g);
            f32x4 b[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
                b[ni] = BDIRECT ? bq[g][ni] : *reinterpret_cast<const f32x4 *>(Bs + ni * 32 * LDT + 8 * g);
#pragma unroll
            for (int st = 0; st < 4; st++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    acc[0][ni] = mfma32(a0[st], b[ni][st], acc[0][ni]);
                    if (MI == 2) acc[MI - 1][ni] = mfma32(a1[st], b[ni][st], acc[MI - 1][ni]);
                }
            // slots are refilled in place with the k-groups of the NEXT stage once their MFMAs have been issued: the first
            // half of the ring in the middle of the stage (pinned there), the second half at its end
            if (BDIRECT && s + 1 < NSTAGE && (g == BK / 16 - 1 || g == BK / 8 - 1)) {
#pragma unroll
                for (int gg = g + 1 - BK / 16; gg <= g; gg++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        bq[gg][ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                            wrs, b_voff, (((s + 1) * (BK / 8) + gg) * HS + ni * 32) * 32, 0));
            }
            if (BDIRECT && g == BK / 16 - 1) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (s + 1 < NSTAGE) STORE_STAGE((s + 1) & 1);
        __syncthreads();
    }

    // epilogue: ReLU, 1x1 conv (canonical tree: in-lane pair per 64-column slab, 32-lane butterfly, slabs in
    // order), sigmoid
    float *red = smem;  // [NSLAB][BM]; every wave is past its last LDS read (barrier above)
    if (MI == 2) {
#pragma unroll
        for (int sl = 0; sl < NI / 2; sl++) {
            const int slab = wn * (NI / 2) + sl;
            const float w2a = w2[slab * 64 + r], w2b = w2[slab * 64 + 32 + r];
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const float h0 = acc[mi][2 * sl][e] > 0.0f ? acc[mi][2 * sl][e] : 0.0f;
                    const float h1 = acc[mi][(NI > 1 ? 2 * sl + 1 : 0)][e] > 0.0f ? acc[mi][(NI > 1 ? 2 * sl + 1 : 0)][e] : 0.0f;
                    const float q = h0 * w2a + h1 * w2b;
                    const float t = bfly32(q);
                    if (r == 0) red[slab * BM + wm * 64 + mi * 32 + crow(e, h)] = t;
                }
        }
    } else {
        // latency form: the two 32-column halves of a 64-column slab belong to waves wn = 2 slab and 2 slab + 1.  The odd wave
        // hands its products h1 * w2b over through LDS; the even wave forms q = h0 * w2a + h1 * w2b exactly as the in-lane
        // form does (two roundings of the products, one of the sum) and reduces it with the same butterfly.
        float *xch = smem + NSLAB * BM;           // [4 slabs][16 e][64 lanes]
        const int slab = wn >> 1;
        const float w2v = w2[wn * 32 + r];
        if (wn & 1) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float h1 = acc[0][0][e] > 0.0f ? acc[0][0][e] : 0.0f;
                xch[(slab * 16 + e) * 64 + lane] = h1 * w2v;
            }
        }
        __syncthreads();
        if (!(wn & 1)) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float h0 = acc[0][0][e] > 0.0f ? acc[0][0][e] : 0.0f;
                const float q = h0 * w2v + xch[(slab * 16 + e) * 64 + lane];
                const float t = bfly32(q);
                if (r == 0) red[slab * BM + crow(e, h)] = t;
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < BM; t += 512) {
        const long long m = m0 + t;
        if (m < n_rows) {
            float logit = b2[0];
#pragma unroll
            for (int s = 0; s < NSLAB; s++) logit = logit + red[s * BM + t];
            sal[m] = sslam_sigmoid(logit);
        }
    }
}
#undef LOAD_STAGE
#undef STORE_STAGE

// ---- halo form (hs = 256, throughput shape) ------------------------------------------------------------------------------
// Same arithmetic, same k order (stage = chunk * 9 + tap), same B stream as selector_saliency_kernel<2, 4, 2, true>; what
// changes is how the A rows reach LDS.  The stage form stages the 128-row tile once per (chunk, tap): 108 times, each with
// two loads, eight selects, two LDS stores per thread and a barrier.  Here a workgroup loads, once per 32-channel chunk, its
// 128 cells PLUS the halo the nine taps reach into an LDS image in ZERO-PADDED coordinates - a frame is (G + 2) rows of
// (G + 1) cells: one zero row above and below, one zero column on the right; padded index of (f, y, x) =
// f (G+2)(G+1) + (y+1)(G+1) + x - so a tap that leaves the grid lands on a zero row (same value the stage form multiplies
// through: +-0 products do not change a sum that starts from the bias... they add +0.0, which is exact) and the nine
// taps are nine row offsets: 12 staging passes and 12 barriers instead of 108, no validity masks.
constexpr int HIMG_ROWS = 256;                  // 512 threads x 2 items of 8 floats / 4 items per row
constexpr int HIMG_FLOATS = HIMG_ROWS * LDT;    // 36 864 B per buffer; two buffers -> two workgroups per CU (147 KB)

// One tile of the halo form.  SMALL = false: 128 cells, waves 2 (rows) x 4 (columns), 64 x 64 per wave.  SMALL = true: 32 cells,
// 8 column waves of 32 x 32 - a quarter of the work, used for the rows BEYOND the last whole round of big tiles (below).
template <bool SMALL>
__device__ __forceinline__ void halo_tile(float *hsmem, const float *__restrict__ feat, int n_rows, int G, const float *__restrict__ w1p,
                                          const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
                                          float *__restrict__ sal, const int m0, const int m_end) {
    // m_end: one past the last cell this tile computes (n_rows, or the end of the tile's frame in the per-frame tiling)
    constexpr int WN = SMALL ? 8 : 4, NI = SMALL ? 1 : 2, MI = SMALL ? 1 : 2, BM = 32 * MI * (8 / WN), HS = 256, NSLAB = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int cells = G * G, G1 = G + 1, P = (G + 2) * G1;
    auto padded = [&](int m) {
        const int f = m / cells, c = m - f * cells, y = c / G, x = c - y * G;
        return f * P + (y + 1) * G1 + x;
    };
    const int p_lo = padded(m0) - (G + 2);

    // loader: item i of this thread = 8 floats (k-group `kq` of the chunk) of image row (tid >> 2) + 128 i.  Rows outside the
    // grid get an offset beyond the descriptor (the range check ignores the scalar chunk offset): the loads return zeros.
    const __amdgpu_buffer_rsrc_t frs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, (int)((unsigned)n_rows * (SSLAM_C * 4u)), 0x00020000);
    const int kq = tid & 3;
    int a_voff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int pr = p_lo + (tid >> 2) + 128 * i;
        int off = -32;
        if (pr >= 0) {
            const int f = pr / P, q = pr - f * P, yy = q / G1, xx = q - yy * G1;
            const long long m = (long long)f * cells + (yy - 1) * G + xx;
            if (yy >= 1 && yy <= G && xx < G && m < n_rows) off = (int)((unsigned)m * (SSLAM_C * 4u) + kq * 32u);
        }
        a_voff[i] = off;
    }
    float4 ra_lo, ra_hi;
#define HL(I, C)                                                                                                   \
    {                                                                                                              \
        ra_lo = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(frs, a_voff[I], (C) * (BK * 4), 0));      \
        ra_hi = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(frs, a_voff[I], (C) * (BK * 4) + 16, 0)); \
    }
#define HS_(I, BUF)                                                                                                \
    {                                                                                                              \
        float4 ev, od;                                                                                             \
        kp8_split(ra_lo, ra_hi, ev, od);                                                                           \
        float *d = hsmem + (BUF) * HIMG_FLOATS + ((tid >> 2) + 128 * (I)) * LDT + kq * 8;                          \
        *reinterpret_cast<float4 *>(d) = ev;                                                                       \
        *reinterpret_cast<float4 *>(d + 4) = od;                                                                   \
    }

    // A fragment rows of this lane: image row of its cell (floats), + 4 h
    int a_base[MI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        const int m = min(m0 + wm * 32 * MI + mi * 32 + r, m_end - 1);
        a_base[mi] = (padded(m) - p_lo) * LDT + 4 * h;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        const float bv = b1[wn * 32 * NI + ni * 32 + r];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ni][e] = bv;
    }
    const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(w1p) + ((wn * 32 * NI + r) * 2 + h);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w1p), 0, 9 * SSLAM_C * HS * 4, 0x00020000);
    const int b_voff = ((wn * 32 * NI + r) * 2 + h) * 16;
    f32x4 bq[BK / 8][NI];
#pragma unroll
    for (int g = 0; g < BK / 8; g++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) bq[g][ni] = bsrc[((long long)g * HS + ni * 32) * 2];

    HL(0, 0);
    HS_(0, 0);
    HL(1, 0);
    HS_(1, 0);
    __syncthreads();
    int tap = 0, dy = -1, dx = -1, chunk = 0;
    for (int s = 0; s < NSTAGE; s++) {
        // next chunk's image: item 0 fetched during tap 0 and stored after tap 3, item 1 fetched then and stored after tap 8
        if (chunk + 1 < NCHUNK) {
            if (tap == 0) HL(0, chunk + 1);
            if (tap == 4) {
                HS_(0, (chunk + 1) & 1);
                HL(1, chunk + 1);
            }
        }
        const float *img = hsmem + (chunk & 1) * HIMG_FLOATS + (dy * G1 + dx) * LDT;
        const float *A0 = img + a_base[0], *A1 = img + a_base[MI - 1];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < BK / 8; g++) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(A0 + 8 * g);
            f32x4 a1 = a0;
            if (MI == 2) a1 = *reinterpret_cast<const f32x4 *>(A1 + 8 * g);
            f32x4 b[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ni++) b[ni] = bq[g][ni];
#pragma unroll
            for (int st = 0; st < 4; st++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    acc[0][ni] = mfma32(a0[st], b[ni][st], acc[0][ni]);
                    if (MI == 2) acc[MI - 1][ni] = mfma32(a1[st], b[ni][st], acc[MI - 1][ni]);
                }
            if (s + 1 < NSTAGE && (g == BK / 16 - 1 || g == BK / 8 - 1)) {
#pragma unroll
                for (int gg = g + 1 - BK / 16; gg <= g; gg++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        bq[gg][ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                            wrs, b_voff, (((s + 1) * (BK / 8) + gg) * HS + ni * 32) * 32, 0));
            }
            if (g == BK / 16 - 1) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        tap++; dx++;
        if (dx == 2) { dx = -1; dy++; }
        if (tap == 9) {
            if (chunk + 1 < NCHUNK) HS_(1, (chunk + 1) & 1);
            __syncthreads();
            tap = 0; dy = -1; chunk++;
        }
    }
#undef HL
#undef HS_
    // epilogue: identical to the stage form (ReLU, 1x1 conv tree, sigmoid)
    __syncthreads();
    float *red = hsmem;
    if (!SMALL) {
#pragma unroll
        for (int sl = 0; sl < NI / 2; sl++) {
            const int slab = wn * (NI / 2) + sl;
            const float w2a = w2[slab * 64 + r], w2b = w2[slab * 64 + 32 + r];
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const float h0 = acc[mi][2 * sl][e] > 0.0f ? acc[mi][2 * sl][e] : 0.0f;
                    const float h1 = acc[mi][NI > 1 ? 2 * sl + 1 : 0][e] > 0.0f ? acc[mi][NI > 1 ? 2 * sl + 1 : 0][e] : 0.0f;
                    const float q = h0 * w2a + h1 * w2b;
                    const float t = bfly32(q);
                    if (r == 0) red[slab * BM + wm * 64 + mi * 32 + crow(e, h)] = t;
                }
        }
    } else {
        // the two 32-column halves of a 64-column slab belong to waves wn = 2 slab and 2 slab + 1: the odd wave hands its
        // products over through LDS, the even wave forms q = h0 * w2a + h1 * w2b exactly as the in-lane form does
        float *xch = hsmem + NSLAB * BM;           // [4 slabs][16 e][64 lanes]
        const int slab = wn >> 1;
        const float w2v = w2[wn * 32 + r];
        if (wn & 1) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float h1 = acc[0][0][e] > 0.0f ? acc[0][0][e] : 0.0f;
                xch[(slab * 16 + e) * 64 + lane] = h1 * w2v;
            }
        }
        __syncthreads();
        if (!(wn & 1)) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float h0 = acc[0][0][e] > 0.0f ? acc[0][0][e] : 0.0f;
                const float q = h0 * w2v + xch[(slab * 16 + e) * 64 + lane];
                const float t = bfly32(q);
                if (r == 0) red[slab * BM + crow(e, h)] = t;
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < BM; t += 512) {
        const long long m = (long long)m0 + t;
        if (m < m_end) {
            float logit = b2[0];
#pragma unroll
            for (int sb = 0; sb < NSLAB; sb++) logit = logit + red[sb * BM + t];
            sal[m] = sslam_sigmoid(logit);
        }
    }
}

// Grid = n_big tiles of 128 cells (XCD-aware order) followed by the 32-cell tiles of the remaining rows.  A launch is
// 7.33 rounds of 512 workgroup slots at 613 frames: during whole rounds the matrix pipe is saturated (7.33 x 0.737 ms of MFMA issue
// = 5.40 of the 5.87 ms the all-big grid took), and the third of a round at the end ran one workgroup per CU at half rate.  With the
// big tiles cut at the last whole round, the rest is dispatched as quarter-size workgroups that fill every slot.
__global__ __launch_bounds__(512, 4) void selector_saliency_halo_kernel(const float *__restrict__ feat, int n_rows, int G,
                                                                         const float *__restrict__ w1p, const float *__restrict__ b1,
                                                                         const float *__restrict__ w2, const float *__restrict__ b2,
                                                                         float *__restrict__ sal, int n_big) {
    extern __shared__ __attribute__((aligned(16))) float hsmem[];      // 2 x HIMG_FLOATS
    const int b = blockIdx.x;
    if (b < n_big) {
        const int q = n_big / 8, rem = n_big % 8, x = b % 8;
        const int tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8;
        halo_tile<false>(hsmem, feat, n_rows, G, w1p, b1, w2, b2, sal, tile * 128, n_rows);
    } else {
        halo_tile<true>(hsmem, feat, n_rows, G, w1p, b1, w2, b2, sal, n_big * 128 + (b - n_big) * 32, n_rows);
    }
}

// PER-FRAME TILING of the same form, for the grids whose frame-crossing tiles need more image rows than the LDS image has
// (G = 40: 299, G = 60: 378 of 256): every frame is cut into its own tiles - tb = cells / 128 big ones, then 32-cell tiles for
// the rest (G = 40: 12 + 2, nothing wasted; G = 60: 28 + 1 with 16 of its 32 cells beyond the frame, 0.4 %) - so no tile sees
// a frame boundary and the image is the tile plus the halo (217 / 255 rows).  Same arithmetic and k order per cell: bit-identical
// to the stage form it replaces there.  Grid: the big tiles up to the last whole round first (XCD-aware ranges), then every
// remaining big tile as its four 32-cell quarters, then the frames' own small tiles.
__global__ __launch_bounds__(512, 4) void selector_saliency_halo_frames_kernel(const float *__restrict__ feat, int n_rows, int G,
                                                                                const float *__restrict__ w1p, const float *__restrict__ b1,
                                                                                const float *__restrict__ w2, const float *__restrict__ b2,
                                                                                float *__restrict__ sal, int n_big, int n_quarter, int tb, int ts) {
    extern __shared__ __attribute__((aligned(16))) float hsmem[];      // 2 x HIMG_FLOATS
    const int b = blockIdx.x, cells = G * G;
    if (b < n_big) {
        const int q = n_big / 8, rem = n_big % 8, x = b % 8;
        const int tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8;
        const int f = tile / tb, j = tile - f * tb;
        halo_tile<false>(hsmem, feat, n_rows, G, w1p, b1, w2, b2, sal, f * cells + j * 128, (f + 1) * cells);
    } else if (b < n_big + n_quarter) {
        const int u = b - n_big, tile = n_big + (u >> 2);
        const int f = tile / tb, j = tile - f * tb;
        halo_tile<true>(hsmem, feat, n_rows, G, w1p, b1, w2, b2, sal, f * cells + j * 128 + (u & 3) * 32, (f + 1) * cells);
    } else {
        const int u = b - n_big - n_quarter, f = u / ts, j = u - f * ts;
        halo_tile<true>(hsmem, feat, n_rows, G, w1p, b1, w2, b2, sal, f * cells + tb * 128 + j * 32, (f + 1) * cells);
    }
}

// ---- latency form 2 (hs = 256, few frames): halo image + the hidden channels split over TWO workgroups ------------------
// One frame is 25 tiles of 32 cells; the first latency form gives each tile one workgroup of 8 waves (two per SIMD: 2 x 16 MFMAs
// per stage and SIMD, 108 stages, a barrier each).  Here a tile has two workgroups of 4 waves - 128 hidden channels each, one
// wave per SIMD, 16 MFMAs per stage - on the halo image (12 barriers): the chain per output is the same single fma chain.  The
// 1x1 conv's canonical tree (in-lane pair per 64-column slab, 32-lane butterfly, slabs IN ORDER) crosses the two workgroups only
// at its last level: each writes its two slab totals per cell to `part` (cells x 4), `saliency_finish_kernel` adds them in order.
template <int NP>      // 8-float items per thread and chunk: the image has NP * 64 rows
__global__ __launch_bounds__(256) void selector_saliency_lat2_kernel(const float *__restrict__ feat, int n_rows, int G,
                                                                      const float *__restrict__ w1p, const float *__restrict__ b1,
                                                                      const float *__restrict__ w2, float *__restrict__ part) {
    constexpr int HS = 256, BM = 32;
    constexpr int IMG_FLOATS = NP * 64 * LDT;
    extern __shared__ __attribute__((aligned(16))) float hsmem[];      // 2 x IMG_FLOATS; reused by the epilogue
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, r = lane & 31, h = lane >> 5;
    const int half = blockIdx.x & 1, tile = blockIdx.x >> 1;
    const int col0 = half * 128 + wn * 32;                             // this wave's 32 hidden channels
    const int cells = G * G, G1 = G + 1, P = (G + 2) * G1;
    const int m0 = tile * BM;
    auto padded = [&](int m) {
        const int f = m / cells, c = m - f * cells, y = c / G, x = c - y * G;
        return f * P + (y + 1) * G1 + x;
    };
    const int p_lo = padded(m0) - (G + 2);
    const __amdgpu_buffer_rsrc_t frs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, (int)((unsigned)n_rows * (SSLAM_C * 4u)), 0x00020000);
    const int kq = tid & 3;
    int a_voff[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int pr = p_lo + (tid >> 2) + 64 * i;
        int off = -32;
        if (pr >= 0) {
            const int f = pr / P, q = pr - f * P, yy = q / G1, xx = q - yy * G1;
            const long long m = (long long)f * cells + (yy - 1) * G + xx;
            if (yy >= 1 && yy <= G && xx < G && m < n_rows) off = (int)((unsigned)m * (SSLAM_C * 4u) + kq * 32u);
        }
        a_voff[i] = off;
    }
    float4 ra_lo[NP], ra_hi[NP];
#define LL(C)                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NP; i++) {                                                                   \
        ra_lo[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(frs, a_voff[i], (C) * (BK * 4), 0));      \
        ra_hi[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(frs, a_voff[i], (C) * (BK * 4) + 16, 0)); \
    }
#define LS(BUF)                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NP; i++) {                                                                   \
        float4 ev, od;                                                                                                 \
        kp8_split(ra_lo[i], ra_hi[i], ev, od);                                                                         \
        float *d = hsmem + (BUF) * IMG_FLOATS + ((tid >> 2) + 64 * i) * LDT + kq * 8;                                  \
        *reinterpret_cast<float4 *>(d) = ev;                                                                           \
        *reinterpret_cast<float4 *>(d + 4) = od;                                                                       \
    }
    const int a_base = (padded(min(m0 + r, n_rows - 1)) - p_lo) * LDT + 4 * h;

    f32x16 acc;
    {
        const float bv = b1[col0 + r];
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = bv;
    }
    const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(w1p) + ((col0 + r) * 2 + h);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w1p), 0, 9 * SSLAM_C * HS * 4, 0x00020000);
    const int b_voff = ((col0 + r) * 2 + h) * 16;
    // ring of TWO stages (8 k-groups): with one wave per SIMD a fragment must be requested >= one L2 round trip ahead, and a
    // stage is only 1 024 matrix cycles here
    f32x4 bq[2][BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; g++) {
        bq[0][g] = bsrc[((long long)g * HS) * 2];
        bq[1][g] = bsrc[((long long)(BK / 8 + g) * HS) * 2];
    }
    LL(0);
    LS(0);
    __syncthreads();
    int tap = 0, dy = -1, dx = -1, chunk = 0;
#pragma unroll 2
    for (int s = 0; s < NSTAGE; s++) {
        if (tap == 0 && chunk + 1 < NCHUNK) LL(chunk + 1);
        const float *A = hsmem + (chunk & 1) * IMG_FLOATS + (dy * G1 + dx) * LDT + a_base;
        f32x4 a[BK / 8];
#pragma unroll
        for (int g = 0; g < BK / 8; g++) a[g] = *reinterpret_cast<const f32x4 *>(A + 8 * g);
#pragma unroll
        for (int g = 0; g < BK / 8; g++) {
#pragma unroll
            for (int st = 0; st < 4; st++) acc = mfma32(a[g][st], bq[s & 1][g][st], acc);
            const int sn = min(s + 2, NSTAGE - 1);                      // refill with the stage after next (clamped: a redundant reload)
            bq[s & 1][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, ((sn * (BK / 8) + g) * HS) * 32, 0));
        }
        tap++; dx++;
        if (dx == 2) { dx = -1; dy++; }
        if (tap == 9) {
            if (chunk + 1 < NCHUNK) LS((chunk + 1) & 1);
            __syncthreads();
            tap = 0; dy = -1; chunk++;
        }
    }
#undef LL
#undef LS
    // epilogue: waves (0, 1) are the 64-column slab 2 half, waves (2, 3) slab 2 half + 1; the odd wave hands its products over
    // through LDS, the even wave forms q = h0 * w2a + h1 * w2b exactly as the in-lane form does and reduces it (butterfly)
    __syncthreads();
    float *xch = hsmem;                        // [2 slabs][16 e][64 lanes]
    const int sl = wn >> 1;
    const float w2v = w2[col0 + r];
    if (wn & 1) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float h1 = acc[e] > 0.0f ? acc[e] : 0.0f;
            xch[(sl * 16 + e) * 64 + lane] = h1 * w2v;
        }
    }
    __syncthreads();
    if (!(wn & 1)) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float h0 = acc[e] > 0.0f ? acc[e] : 0.0f;
            const float q = h0 * w2v + xch[(sl * 16 + e) * 64 + lane];
            const float t = bfly32(q);
            const int m = m0 + crow(e, h);
            if (r == 0 && m < n_rows) part[(long long)m * 4 + half * 2 + sl] = t;
        }
    }
}

__global__ __launch_bounds__(256) void saliency_finish_kernel(const float *__restrict__ part, const float *__restrict__ b2, int n_rows,
                                                               float *__restrict__ sal) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_rows) return;
    const f32x4 p4 = *reinterpret_cast<const f32x4 *>(part + (long long)m * 4);
    float logit = b2[0];
#pragma unroll
    for (int sb = 0; sb < 4; sb++) logit = logit + p4[sb];
    sal[m] = sslam_sigmoid(logit);
}

// image rows a 32-cell tile needs (exact maximum over the tile positions of one period)
int halo_rows32(int G, long long n_rows) {
    thread_local int c_G = 0, c_val = 0;           // the scan below is ~3 000 iterations at 613 frames: remember the last answer
    thread_local long long c_rows = 0;
    if (G == c_G && n_rows == c_rows) return c_val;
    const int cells = G * G, G1 = G + 1, P = (G + 2) * G1;
    auto padded = [&](long long m) { const long long f = m / cells, c = m - f * cells, y = c / G, x = c - y * G; return f * P + (y + 1) * G1 + x; };
    long long worst = 0;
    const long long n_tiles = (n_rows + 31) / 32, scan = n_tiles < 4LL * cells ? n_tiles : 4LL * cells;
    for (long long t = 0; t < scan; t++) {
        const long long m0 = t * 32, m1 = m0 + 31 < n_rows - 1 ? m0 + 31 : n_rows - 1;
        const long long need = padded(m1) - padded(m0) + 2 * (G + 2) + 1;
        worst = need > worst ? need : worst;
    }
    c_G = G, c_rows = n_rows, c_val = (int)worst;
    return (int)worst;
}

// image rows the halo form needs (exact maximum over the tile positions of one period)
int halo_rows128(int G, long long n_rows) {
    thread_local int c_G = 0, c_val = 0;           // the scan below is ~3 000 iterations at 613 frames: remember the last answer
    thread_local long long c_rows = 0;
    if (G == c_G && n_rows == c_rows) return c_val;
    const int cells = G * G, G1 = G + 1, P = (G + 2) * G1;
    auto padded = [&](long long m) { const long long f = m / cells, c = m - f * cells, y = c / G, x = c - y * G; return f * P + (y + 1) * G1 + x; };
    long long worst = 0;
    const long long n_tiles = (n_rows + 127) / 128, scan = n_tiles < 4LL * cells ? n_tiles : 4LL * cells;
    for (long long t = 0; t < scan; t++) {
        const long long m0 = t * 128, m1 = m0 + 127 < n_rows - 1 ? m0 + 127 : n_rows - 1;
        const long long need = padded(m1) - padded(m0) + 2 * (G + 2) + 1;
        worst = need > worst ? need : worst;
    }
    c_G = G, c_rows = n_rows, c_val = (int)worst;
    return (int)worst;
}

// image rows the per-frame tiling needs: tiles start at multiples of 128 (big) or 32 (small) inside ONE frame
int halo_rows_frame(int G) {
    thread_local int c_G = 0, c_val = 0;
    if (G == c_G) return c_val;
    const int cells = G * G, G1 = G + 1;
    auto padded = [&](int c) { const int y = c / G, x = c - y * G; return (y + 1) * G1 + x; };
    int worst = 0;
    const int tb = cells / 128;
    for (int t = 0; t < tb; t++) {
        const int need = padded(t * 128 + 127) - padded(t * 128) + 2 * (G + 2) + 1;
        worst = need > worst ? need : worst;
    }
    for (int c0 = tb * 128; c0 < cells; c0 += 32) {
        const int c1 = c0 + 31 < cells - 1 ? c0 + 31 : cells - 1;
        const int need = padded(c1) - padded(c0) + 2 * (G + 2) + 1;
        worst = need > worst ? need : worst;
    }
    c_G = G, c_val = worst;
    return worst;
}

template <int WM, int WN, int NI, bool BD, int MI = 2>
void launch(const float *feat, long long rows, int G, const float *w1p, const float *b1, const float *w2, const float *b2,
            float *sal, hipStream_t st) {
    constexpr int BM = 32 * MI * WM;
    const int n_tiles = (int)((rows + BM - 1) / BM);
    hipLaunchKernelGGL((selector_saliency_kernel<WM, WN, NI, BD, MI>), dim3(n_tiles), dim3(512), 0, st, feat, (int)rows, G, w1p, b1,
                       w2, b2, sal, n_tiles);
}

}  // namespace

// rows up to which latency form 2 runs by default: one wave per SIMD stops paying once the 50 workgroups per frame fill the
// chip (6 frames at G = 28).  Measured (tools/conv_forms.py, G = 28): 1 / 4 / 8 / 16 frames: 0.070 / 0.077 / 0.129 / 0.247 ms
// against 0.123 / 0.123 / 0.129 / 0.224 ms for the 8-wave latency form
static const long long LAT2_DEFAULT_ROWS = 32 * 25 * 6;
// few frames (the drop-in scripts call frame by frame): 128-row tiles would occupy rows / 128 of the 256 CUs for 108 serial
// stages of 64 MFMAs; the latency form cuts a stage to 16 MFMAs per wave and uses 4x the workgroups (0.12 vs 0.47 ms for one
// frame; still ahead at 256 frames: 2.90 vs 3.02 ms).  Measured cross-over: ~400 frames at G = 28 (6.25 vs 6.10 ms at 613)
static const long long LAT_DEFAULT_ROWS = 128 * 1800;

// scratch of latency form 2: the two half-sums of the 1x1 conv per cell and workgroup half (16 bytes per cell)
extern "C" long long sslam_selector_saliency_workspace_bytes(int n_frames, int G) {
    if (n_frames <= 0 || G <= 0) return SSLAM_E_INVALID;
    const long long rows = (long long)n_frames * G * G;
    return rows <= sslam_knob(KNOB_CONV_LAT2_ROWS, LAT2_DEFAULT_ROWS) ? rows * 16 : 0;
}

extern "C" int sslam_selector_saliency_ws(const float *feat, int n_frames, int G, const float *w1_packed, const float *b1,
                                          const float *w2, const float *b2, int hs, float *sal, void *workspace,
                                          long long workspace_bytes, void *stream) {
    if (!feat || !w1_packed || !b1 || !w2 || !b2 || !sal || n_frames <= 0 || G <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)feat | (uintptr_t)w1_packed | (uintptr_t)workspace) & 15) return SSLAM_E_INVALID;
    const long long rows = (long long)n_frames * G * G;
    if (rows * (long long)(SSLAM_C * 4) > 0xffffffffLL) return SSLAM_E_UNSUPPORTED;   // one buffer descriptor spans the feature map
    hipStream_t st = (hipStream_t)stream;
    // test-only knobs (common.h), read once at load time; measured on MI355X (613 frames, G=28): variant 0: 7.16 ms, 1: 7.06, 2: 6.67 (default), 3: 7.97
    const int variant = (int)sslam_knob(KNOB_CONV_VARIANT, 2);
    const long long lat_rows = sslam_knob(KNOB_CONV_LATENCY_ROWS, LAT_DEFAULT_ROWS);   // 0 forces the throughput form
    const long long lat2_rows = sslam_knob(KNOB_CONV_LAT2_ROWS, LAT2_DEFAULT_ROWS);    // 0 disables latency form 2
    // latency form 2 needs 16 bytes of CALLER-OWNED scratch per cell; without it the 8-wave latency form runs (same bits)
    if (hs == 256 && rows <= lat2_rows && rows <= lat_rows && workspace && workspace_bytes >= rows * 16) {
        const int np = (halo_rows32(G, rows) + 63) / 64;
        if (np <= 6) {
            float *part = (float *)workspace;
            const int n_tiles = (int)((rows + 31) / 32);
#define LAT2(NP_)                                                                                                              \
    hipLaunchKernelGGL((selector_saliency_lat2_kernel<NP_>), dim3(2 * n_tiles), dim3(256), 2 * NP_ * 64 * LDT * sizeof(float), st, feat, \
                       (int)rows, G, w1_packed, b1, w2, part)
            switch (np) {
                case 6: LAT2(6); break;
                case 5: LAT2(5); break;
                case 4: LAT2(4); break;
                case 3: LAT2(3); break;
                default: LAT2(2); break;
            }
#undef LAT2
            SSLAM_CHECK_LAUNCH();
            hipLaunchKernelGGL(saliency_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, part, b2, (int)rows, sal);
            SSLAM_CHECK_LAUNCH();
            return SSLAM_OK;
        }
    }
    if (hs == 256 && rows <= lat_rows) {
        launch<1, 8, 1, true, 1>(feat, rows, G, w1_packed, b1, w2, b2, sal, st);
    } else if (hs == 256 && variant == 2 && !sslam_knob(KNOB_CONV_NO_HALO, 0) && halo_rows128(G, rows) <= HIMG_ROWS) {
        // big tiles up to the last whole round of 2 workgroups x 256 CUs, 32-cell tiles for the rest
        const int n_tiles = (int)((rows + 127) / 128);
        const int round = (int)sslam_knob(KNOB_CONV_TAIL, 512);       // round size in tiles (tests use a small one); 0: all big
        const int n_big = round > 0 && n_tiles > round ? n_tiles / round * round : n_tiles;
        const int n_small = n_big < n_tiles ? (int)((rows - (long long)n_big * 128 + 31) / 32) : 0;
        hipLaunchKernelGGL(selector_saliency_halo_kernel, dim3(n_big + n_small), dim3(512), 2 * HIMG_FLOATS * sizeof(float), st, feat, (int)rows, G,
                           w1_packed, b1, w2, b2, sal, n_big);
    } else if (hs == 256 && variant == 2 && !sslam_knob(KNOB_CONV_NO_HALO, 0) && G * G >= 128 && halo_rows_frame(G) <= HIMG_ROWS) {
        // the grids whose frame-crossing tiles do not fit the image: every frame tiled on its own
        const int cells = G * G, tb = cells / 128, ts = (cells - tb * 128 + 31) / 32;
        const long long nb = (long long)n_frames * tb;
        const int round = (int)sslam_knob(KNOB_CONV_TAIL, 512);
        const long long n_big = round > 0 && nb > round ? nb / round * round : nb;
        const long long n_quarter = (nb - n_big) * 4, n_small = (long long)n_frames * ts;
        hipLaunchKernelGGL(selector_saliency_halo_frames_kernel, dim3((unsigned)(n_big + n_quarter + n_small)), dim3(512),
                           2 * HIMG_FLOATS * sizeof(float), st, feat, (int)rows, G, w1_packed, b1, w2, b2, sal, (int)n_big, (int)n_quarter, tb, ts);
    } else if (hs == 256) {
        switch (variant) {
            case 0: launch<2, 4, 2, false>(feat, rows, G, w1_packed, b1, w2, b2, sal, st); break;
            case 1: launch<4, 2, 4, false>(feat, rows, G, w1_packed, b1, w2, b2, sal, st); break;
            case 3: launch<4, 2, 4, true>(feat, rows, G, w1_packed, b1, w2, b2, sal, st); break;
            default: launch<2, 4, 2, true>(feat, rows, G, w1_packed, b1, w2, b2, sal, st); break;
        }
    } else if (hs == 128) {
        if (variant >= 2)
            launch<4, 2, 2, true>(feat, rows, G, w1_packed, b1, w2, b2, sal, st);
        else
            launch<4, 2, 2, false>(feat, rows, G, w1_packed, b1, w2, b2, sal, st);
    } else {
        return SSLAM_E_UNSUPPORTED;
    }
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

// the form without a workspace: never allocates - few-frame calls take the 8-wave latency form instead of latency form 2
extern "C" int sslam_selector_saliency(const float *feat, int n_frames, int G, const float *w1_packed, const float *b1,
                                       const float *w2, const float *b2, int hs, float *sal, void *stream) {
    return sslam_selector_saliency_ws(feat, n_frames, G, w1_packed, b1, w2, b2, hs, sal, nullptr, 0, stream);
}

// w (hs, 384, 3, 3) -> [stage = chunk*9 + tap][k-group g (4)][n][8 floats in KP8 order] (MFMA B-fragment order)
extern "C" int sslam_pack_conv3x3_host(const float *w, int hs, float *out) {
    if (!w || !out || hs <= 0) return SSLAM_E_INVALID;
    for (int chunk = 0; chunk < NCHUNK; chunk++)
        for (int tap = 0; tap < 9; tap++)
            for (int n = 0; n < hs; n++)
                for (int k = 0; k < BK; k++) {
                    const int c = chunk * BK + k, stage = chunk * 9 + tap;
                    out[((((long long)stage * (BK / 8) + k / 8) * hs + n) * 8) + kp8(k % 8)] =
                        w[((long long)n * SSLAM_C + c) * 9 + tap];
                }
    return SSLAM_OK;
}

// selector_bf16.hip - A3 in the bf16 THROUGHPUT mode (BASELINE.json configs[1] "bf16 conv stack"):
// the saliency CNN of keypoint_selector.py:45-67 with bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation and
// the same fused fp32 epilogue (bias start, ReLU, 1x1 conv tree, canonical sigmoid) as the exact kernel (selector.hip).
// NOT index-exact against the fp32 CPU reference by construction (operands are rounded to 8 significant bits): the
// host layer reports the keypoint / match agreement with the exact mode next to the throughput.  Deterministic.
//
// Same implicit-GEMM structure as selector.hip: M = cells, N = hs, K = 3456 chunk-major in 27 stages of 128 channels
// (3 channel chunks x 9 taps); A rows come from a bf16 copy of the feature map (written by sslam_bn_tokens) through a
// register-double-buffered LDS stage (272-B rows), B fragments stream from L2 in fragment order, 4 k-steps ahead.
// With 16x the matrix rate the kernel is bound by the A-tile traffic (9 L1/L2 re-reads of 256 B per row and stage).
#include <algorithm>

#include "common.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BKB = 128;                 // channels per stage
constexpr int LDB = BKB + 8;             // LDS row (bf16 elements): 272 B = 17 x 16 B
constexpr int NSTB = 9 * (SSLAM_C / BKB);
constexpr int KSB = BKB / 16;            // MFMA k-steps per stage
constexpr int RING = 4;

// WM x WN waves; each wave owns MI x 2 accumulator tiles (32*MI rows x 64 columns).  MI = 4 halves the B-fragment
// traffic per MFMA (the L1 path of the direct B loads is what bounds the MI = 2 form).
template <int WM, int WN, int MI>
__global__ __launch_bounds__(64 * WM * WN) void selector_bf16_kernel(const bf16 *__restrict__ feat, int n_rows, int G,
                                                                     const bf16 *__restrict__ w1p, const float *__restrict__ b1,
                                                                     const float *__restrict__ w2, const float *__restrict__ b2,
                                                                     float *__restrict__ sal, int n_tiles) {
    constexpr int NTH = 64 * WM * WN, BM = 32 * MI * WM, HS = 64 * WN, NSLAB = HS / 64;
    constexpr int A_ITEMS = BM * (BKB / 8) / NTH;          // 16-B pieces per thread per stage
    constexpr int ROWSTEP = NTH / (BKB / 8);               // rows between a thread's consecutive pieces
    constexpr int STAGE = BM * LDB;
    constexpr int SMEM = 2 * STAGE * 2 > NSLAB * BM * 4 ? 2 * STAGE * 2 : NSLAB * BM * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM];
    bf16 *smem = reinterpret_cast<bf16 *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int cells = G * G;
    int tile;
    {
        const int b = blockIdx.x, q = n_tiles / 8, rem = n_tiles % 8, x = b % 8;
        tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8;
    }
    const long long m0 = (long long)tile * BM;

    // piece i of this thread: row = tid / 16 + ROWSTEP * i, 16-B column a_pc = tid % 16 (the same for every piece).
    // Buffer loads (see selector.hip): descriptor base + 32-bit lane offset of the row's own cell (+ the tap shift, one
    // add) + SCALAR chunk offset; taps outside the grid are zeroed at STORE_STAGE by a precomputed 9-bit validity mask.
    const int a_row0 = tid / (BKB / 8), a_pc = tid % (BKB / 8);
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(feat), 0, (int)((unsigned)n_rows * (SSLAM_C * 2u)), 0x00020000);
    int a_voff[A_ITEMS], a_mask[A_ITEMS];
#pragma unroll
    for (int i = 0; i < A_ITEMS; i++) {
        const long long m = m0 + a_row0 + ROWSTEP * i;
        const bool okr = m < n_rows;
        const long long mm = okr ? m : m0;
        const int cell = (int)(mm % cells), y = cell / G, x = cell % G;
        a_voff[i] = (int)((unsigned)mm * (SSLAM_C * 2u) + a_pc * 16u);
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            mk |= (okr && yy >= 0 && yy < G && xx >= 0 && xx < G) ? (1 << t) : 0;
        }
        a_mask[i] = mk;
    }
    u32x4 ra[A_ITEMS];
    unsigned ra_keep[A_ITEMS];
#define LOAD_STAGE(S)                                                                                        \
    {                                                                                                        \
        const int s_ = (S);                                                                                  \
        const int chunk = s_ / 9, tap = s_ - chunk * 9;                                                      \
        const int toff_ = ((tap / 3 - 1) * G + (tap % 3 - 1)) * (SSLAM_C * 2), soff_ = chunk * (BKB * 2);    \
        _Pragma("unroll") for (int i = 0; i < A_ITEMS; i++) {                                                \
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(frs, a_voff[i] + toff_, soff_, 0);                 \
            ra_keep[i] = ((a_mask[i] >> tap) & 1) ? 0xffffffffu : 0u;                                        \
        }                                                                                                    \
    }
#define STORE_STAGE(BUF)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < A_ITEMS; i++)                                                      \
        *reinterpret_cast<u32x4 *>(smem + (BUF) * STAGE + (a_row0 + ROWSTEP * i) * LDB + a_pc * 8) = ra[i] & ra_keep[i];

    f32x16 acc[MI][2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
        const float bv = b1[wn * 64 + ni * 32 + r];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ni][e] = bv;
    }
    // B fragment (global k-step = stage*8 + ks, N tile t): 16 B per lane at ((kstep * (HS/32) + t) * 64 + lane)
    const bf16x8 *bsrc = reinterpret_cast<const bf16x8 *>(w1p) + (wn * 2) * 64 + lane;
    constexpr int GSTR = (HS / 32) * 64;           // bf16x8 elements per global k-step
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(w1p), 0, 9 * SSLAM_C * HS * 2, 0x00020000);
    const int b_voff = ((wn * 2) * 64 + lane) * 16;
    bf16x8 bq[RING][2];
#pragma unroll
    for (int i = 0; i < RING; i++) {
        bq[i][0] = bsrc[(long long)i * GSTR];
        bq[i][1] = bsrc[(long long)i * GSTR + 64];
    }
    LOAD_STAGE(0);
    STORE_STAGE(0);
    __syncthreads();
    for (int s = 0; s < NSTB; s++) {
        if (s + 1 < NSTB) LOAD_STAGE(s + 1);
        const bf16 *As = smem + (s & 1) * STAGE + (wm * 32 * MI + r) * LDB + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KSB; ks++) {
            bf16x8 a[MI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) a[mi] = *reinterpret_cast<const bf16x8 *>(As + mi * 32 * LDB + ks * 16);
            const bf16x8 b0 = bq[ks % RING][0], b1v = bq[ks % RING][1];
            const int gn = min(s * KSB + ks + RING, NSTB * KSB - 1);      // unconditional refill (clamped at the tail)
            bq[ks % RING][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, gn * (GSTR * 16), 0));
            bq[ks % RING][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, gn * (GSTR * 16) + 1024, 0));
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b0, acc[mi][0], 0, 0, 0);
                acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b1v, acc[mi][1], 0, 0, 0);
            }
        }
        if (s + 1 < NSTB) STORE_STAGE((s + 1) & 1);
        __syncthreads();
    }
#undef LOAD_STAGE
#undef STORE_STAGE
    // epilogue identical to the exact kernel: ReLU, 1x1 conv tree, sigmoid (fp32)
    float *red = reinterpret_cast<float *>(smem_raw);
    {
        const float w2a = w2[wn * 64 + r], w2b = w2[wn * 64 + 32 + r];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float h0 = acc[mi][0][e] > 0.0f ? acc[mi][0][e] : 0.0f;
                const float h1 = acc[mi][1][e] > 0.0f ? acc[mi][1][e] : 0.0f;
                const float t = bfly32(h0 * w2a + h1 * w2b);
                if (r == 0) red[wn * BM + wm * 32 * MI + mi * 32 + crow(e, h)] = t;
            }
    }
    __syncthreads();
    for (int t = tid; t < BM; t += NTH) {
        const long long m = m0 + t;
        if (m < n_rows) {
            float logit = b2[0];
#pragma unroll
            for (int sl = 0; sl < NSLAB; sl++) logit = logit + red[sl * BM + t];
            sal[m] = sslam_sigmoid(logit);
        }
    }
}

// ---- halo form (hs = 256) ---------------------------------------------------------------------------------------------
// The stage-per-tap kernel above re-reads every 256-row A tile nine times from L1/L2 (once per tap): 3.3 GB of the 7 GB it
// moves into the CUs per 613 frames, at the ~8 TB/s that per-lane 16-byte loads sustain from L2.  Here a workgroup loads,
// once per 64-channel chunk, the rows of its 256 cells PLUS the halo the nine taps reach, and the taps become nine row
// offsets into that LDS image.  The image lives in PADDED coordinates: a frame is (G + 2) rows of (G + 1) cells - one zero
// row above and below, one zero column on the right - so that a tap that leaves the grid lands on a zero row by
// construction: no per-tap validity masks.  Padded index of cell (f, y, x): f * (G+2)(G+1) + (y+1)(G+1) + x.
//   - 8 waves = 2 (rows) x 4 (columns), wave tile 128 x 64 as above; 36 k-steps per chunk, 6 chunks: 6 barriers, not 27;
//   - image rows are 144 B (128 B of channels + 16 B pad: consecutive rows rotate by 4 banks mod 64 - 16 lanes cover all 64);
//   - B fragments stream from L2 exactly as above (same packed weights, same ring).
constexpr int HC = 64;                   // channels per chunk
constexpr int HROW = HC * 2 + 16;        // bytes per image row
constexpr int HKS = HC / 16;             // k-steps per (chunk, tap)
constexpr int HRING = 4;

// One tile: NP = 16-byte pieces per thread and chunk (the image has NP * 64 rows); MI = row tiles of 32 per wave: 4 -> 256-cell
// tiles, 2 -> 128-cell tiles (half the work; for the rows beyond the last whole round, see the kernel below)
template <int NP, int MI>
__device__ __forceinline__ void halo_bf16_tile(unsigned char *hsm, const bf16 *__restrict__ feat, int n_rows, int G,
                                               const bf16 *__restrict__ w1p, const float *__restrict__ b1, const float *__restrict__ w2,
                                               const float *__restrict__ b2, float *__restrict__ sal, const int m0) {
    constexpr int WN = 4, BM = 64 * MI, HS = 256, NSLAB = HS / 64, NTH = 512;
    constexpr int IMG = NP * 64 * HROW;                        // bytes per image buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int cells = G * G, G1 = G + 1, P = (G + 2) * G1;
    auto padded = [&](int m) {
        const int f = m / cells, c = m - f * cells, y = c / G, x = c - y * G;
        return f * P + (y + 1) * G1 + x;
    };
    const int p_lo = padded(m0) - (G + 2);                    // first padded row of the image (may be negative: zeros)

    // loader: piece i of this thread is 16 B (8 channels) `part` of image row tid / 8 + 64 i.  Source row decoded once;
    // rows outside the grid (padding, other side of the sequence ends) get an offset beyond the descriptor: the load returns 0.
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(feat), 0, (int)((unsigned)n_rows * (SSLAM_C * 2u)), 0x00020000);
    const int part = tid & 7;
    int a_voff[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int pr = p_lo + (tid >> 3) + 64 * i;
        int off = -16;                                        // 0xfffffff0: out of range
        if (pr >= 0) {
            const int f = pr / P, q = pr - f * P, yy = q / G1, xx = q - yy * G1;
            const long long m = (long long)f * cells + (yy - 1) * G + xx;
            if (yy >= 1 && yy <= G && xx < G && m < n_rows) off = (int)((unsigned)m * (SSLAM_C * 2u) + part * 16u);
        }
        a_voff[i] = off;
    }
    u32x4 ra[NP];
#define H_LOAD(C)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NP; i++) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(frs, a_voff[i], (C) * (HC * 2), 0);
#define H_STORE(BUF)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NP; i++)                                                           \
        *reinterpret_cast<u32x4 *>(hsm + (BUF) * IMG + ((tid >> 3) + 64 * i) * HROW + part * 16) = ra[i];

    // A fragment of lane (r, h), row tile mi: image row of its cell, + the tap's row offset, + 32 ks + 16 h bytes
    int a_off[MI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        const int m = min(m0 + wm * 32 * MI + mi * 32 + r, n_rows - 1);
        a_off[mi] = (padded(m) - p_lo) * HROW + 16 * h;
    }

    f32x16 acc[MI][2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
        const float bv = b1[wn * 64 + ni * 32 + r];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ni][e] = bv;
    }
    // B fragments: packed [stage = chunk128 * 9 + tap][ks8][n/32][64 lanes][8]; chunk c (64 channels), tap t, k-step k4 is
    // stage (c >> 1) * 9 + t, ks8 = (c & 1) * 4 + k4
    constexpr int GSTR = (HS / 32) * 64 * 16;          // bytes per global k-step
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(w1p), 0, 9 * SSLAM_C * HS * 2, 0x00020000);
    const int b_voff = ((wn * 2) * 64 + lane) * 16;
    auto chunk_base = [](int c) { return (((c >> 1) * 9) * 8 + (c & 1) * 4) * GSTR; };
    bf16x8 bq[HRING][2];
#pragma unroll
    for (int i = 0; i < HRING; i++) {                   // steps 0..3: chunk 0, tap 0, k4 = i
        bq[i][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, i * GSTR, 0));
        bq[i][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, i * GSTR + 1024, 0));
    }
    H_LOAD(0);
    H_STORE(0);
    __syncthreads();
    constexpr int NCH = SSLAM_C / HC;
    // Software pipeline, fixed by scheduling barriers (left alone, the scheduler sinks every load next to its use: the ring
    // collapses to one step and each MFMA group waits for an LDS round trip): per k-step
    //     [A fragments of step s + 1 : 4 ds_read_b128]  [8 MFMAs of step s]  [B refill of the slot just consumed <- step s + HRING]
#pragma unroll 1
    for (int c = 0; c < NCH; c++) {
        if (c + 1 < NCH) H_LOAD(c + 1);
        const unsigned char *img = hsm + (c & 1) * IMG;
        const int base_c = chunk_base(c), base_n = chunk_base(min(c + 1, NCH - 1));
        bf16x8 a[2][MI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++) a[0][mi] = *reinterpret_cast<const bf16x8 *>(img + a_off[mi] - (G1 + 1) * HROW);   // tap 0, k4 0
#pragma unroll
        for (int st = 0; st < 9 * HKS; st++) {
            const int cur = st & 1, slot = st % HRING;
            if (st + 1 < 9 * HKS) {
                const int tap = (st + 1) / HKS, k4 = (st + 1) % HKS;
                const int toff = ((tap / 3 - 1) * G1 + (tap % 3 - 1)) * HROW + k4 * 32;
#pragma unroll
                for (int mi = 0; mi < MI; mi++) a[cur ^ 1][mi] = *reinterpret_cast<const bf16x8 *>(img + a_off[mi] + toff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][mi], bq[slot][0], acc[mi][0], 0, 0, 0);
                acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][mi], bq[slot][1], acc[mi][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // refill: step + HRING, in this chunk or the next (clamped at the very end: a redundant reload)
                const int nxt = st + HRING;
                const int goff = nxt < 9 * HKS ? base_c + ((nxt / HKS) * 8 + nxt % HKS) * GSTR
                                               : base_n + (((nxt - 9 * HKS) / HKS) * 8 + (nxt - 9 * HKS) % HKS) * GSTR;
                bq[slot][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, goff, 0));
                bq[slot][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, b_voff, goff + 1024, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < NCH) H_STORE((c + 1) & 1);
        __syncthreads();
    }
#undef H_LOAD
#undef H_STORE
    // epilogue identical to the exact kernel: ReLU, 1x1 conv tree, sigmoid (fp32)
    float *red = reinterpret_cast<float *>(hsm);
    {
        const float w2a = w2[wn * 64 + r], w2b = w2[wn * 64 + 32 + r];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float h0 = acc[mi][0][e] > 0.0f ? acc[mi][0][e] : 0.0f;
                const float h1 = acc[mi][1][e] > 0.0f ? acc[mi][1][e] : 0.0f;
                const float t = bfly32(h0 * w2a + h1 * w2b);
                if (r == 0) red[wn * BM + wm * 32 * MI + mi * 32 + crow(e, h)] = t;
            }
    }
    __syncthreads();
    for (int t = tid; t < BM; t += NTH) {
        const long long m = (long long)m0 + t;
        if (m < n_rows) {
            float logit = b2[0];
#pragma unroll
            for (int sl = 0; sl < NSLAB; sl++) logit = logit + red[sl * BM + t];
            sal[m] = sslam_sigmoid(logit);
        }
    }
}

// Grid = n_big tiles of 256 cells (XCD-aware order), then 128-cell tiles for the remaining rows: with one workgroup per CU a
// launch of 7.33 rounds (613 frames) otherwise costs 8 (tools/conv_rounds.py shows the same steps for the exact kernel).
template <int NP>
__global__ __launch_bounds__(512) void selector_bf16_halo_kernel(const bf16 *__restrict__ feat, int n_rows, int G,
                                                                  const bf16 *__restrict__ w1p, const float *__restrict__ b1,
                                                                  const float *__restrict__ w2, const float *__restrict__ b2,
                                                                  float *__restrict__ sal, int n_big) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int b = blockIdx.x;
    if (b < n_big) {
        const int q = n_big / 8, rem = n_big % 8, x = b % 8;
        const int tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8;
        halo_bf16_tile<NP, 4>(hsm, feat, n_rows, G, w1p, b1, w2, b2, sal, tile * 256);
    } else {
        halo_bf16_tile<NP, 2>(hsm, feat, n_rows, G, w1p, b1, w2, b2, sal, n_big * 256 + (b - n_big) * 128);
    }
}

// image rows the halo kernel needs for a G x G grid (exact maximum over tile positions within one period of the pattern)
int halo_rows(int G, long long n_rows) {
    thread_local int c_G = 0, c_val = 0;           // the scan below is ~3 000 iterations at 613 frames: remember the last answer
    thread_local long long c_rows = 0;
    if (G == c_G && n_rows == c_rows) return c_val;
    const int cells = G * G, G1 = G + 1, P = (G + 2) * G1;
    auto padded = [&](long long m) { const long long f = m / cells, c = m - f * cells, y = c / G, x = c - y * G; return f * P + (y + 1) * G1 + x; };
    long long worst = 0;
    const long long n_tiles = (n_rows + 255) / 256, scan = std::min<long long>(n_tiles, 4LL * cells);   // the pattern repeats every lcm(256, cells) rows
    for (long long t = 0; t < scan; t++) {
        const long long m0 = t * 256, m1 = std::min<long long>(m0 + 255, n_rows - 1);
        worst = std::max(worst, padded(m1) - padded(m0) + 2 * (G + 2) + 1);
    }
    c_G = G, c_rows = n_rows, c_val = (int)worst;
    return (int)worst;
}

__device__ __forceinline__ unsigned short f2bf(float v) { return __builtin_bit_cast(unsigned short, (bf16)v); }

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ in, bf16 *__restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = *reinterpret_cast<const float4 *>(in + 8 * i), b = *reinterpret_cast<const float4 *>(in + 8 * i + 4);
    u32x4 o;
    o[0] = f2bf(a.x) | ((unsigned)f2bf(a.y) << 16);
    o[1] = f2bf(a.z) | ((unsigned)f2bf(a.w) << 16);
    o[2] = f2bf(b.x) | ((unsigned)f2bf(b.y) << 16);
    o[3] = f2bf(b.z) | ((unsigned)f2bf(b.w) << 16);
    *reinterpret_cast<u32x4 *>(out + 8 * i) = o;
}

unsigned short host_bf16(float v) {   // round-to-nearest-even, as v_cvt_pk_bf16_f32 (finite inputs)
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

// fp32 (n) -> bf16 (n), n % 8 == 0: the bf16 copy of the feature map consumed by sslam_selector_saliency_bf16
extern "C" int sslam_f32_to_bf16(const float *in, void *out_bf16, long long n, void *stream) {
    if (!in || !out_bf16 || n <= 0 || (n & 7) || (((uintptr_t)in | (uintptr_t)out_bf16) & 15)) return SSLAM_E_INVALID;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (bf16 *)out_bf16, n / 8);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

// w (hs, 384, 3, 3) fp32 -> bf16 in [stage = chunk*9 + tap][k-step (8)][n/32][half (2)][row (32)][8] order
extern "C" int sslam_pack_conv3x3_bf16_host(const float *w, int hs, void *out_bf16) {
    if (!w || !out_bf16 || hs <= 0 || hs % 32) return SSLAM_E_INVALID;
    unsigned short *out = (unsigned short *)out_bf16;
    for (int chunk = 0; chunk < SSLAM_C / BKB; chunk++)
        for (int tap = 0; tap < 9; tap++)
            for (int n = 0; n < hs; n++)
                for (int k = 0; k < BKB; k++) {
                    const int c = chunk * BKB + k, stage = chunk * 9 + tap, ks = k / 16, hh = (k % 16) / 8, j = k % 8;
                    const long long idx = (((((long long)stage * KSB + ks) * (hs / 32) + n / 32) * 2 + hh) * 32 + n % 32) * 8 + j;
                    out[idx] = host_bf16(w[((long long)n * SSLAM_C + c) * 9 + tap]);
                }
    return SSLAM_OK;
}

extern "C" int sslam_selector_saliency_bf16(const void *feat_bf16, int n_frames, int G, const void *w1_packed_bf16,
                                            const float *b1, const float *w2, const float *b2, int hs, float *sal, void *stream) {
    if (!feat_bf16 || !w1_packed_bf16 || !b1 || !w2 || !b2 || !sal || n_frames <= 0 || G <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)feat_bf16 | (uintptr_t)w1_packed_bf16) & 15) return SSLAM_E_INVALID;
    const long long rows = (long long)n_frames * G * G;
    if (rows * (long long)(SSLAM_C * 2) > 0xffffffffLL) return SSLAM_E_UNSUPPORTED;   // one buffer descriptor spans the bf16 feature map
    hipStream_t st = (hipStream_t)stream;
    if (hs == 256 && !sslam_knob(KNOB_CONVBF_NO_HALO, 0)) {
        const int np = std::max(5, (halo_rows(G, rows) + 63) / 64);      // instantiated for 5..10 x 64 image rows
        if (np <= 10 && rows < (1LL << 31) / 2) {
            const int n_tiles = (int)((rows + 255) / 256);
            const int round = (int)sslam_knob(KNOB_CONVBF_TAIL, 256);           // round size in tiles (one workgroup per CU); 0: all big
            const int n_big = round > 0 && n_tiles > round ? n_tiles / round * round : n_tiles;
            const int n_small = n_big < n_tiles ? (int)((rows - (long long)n_big * 256 + 127) / 128) : 0;
            const size_t lds = (size_t)2 * np * 64 * HROW;
#define HALO(NP_)                                                                                                            \
    hipLaunchKernelGGL((selector_bf16_halo_kernel<NP_>), dim3(n_big + n_small), dim3(512), lds, st, (const bf16 *)feat_bf16, (int)rows, G, \
                       (const bf16 *)w1_packed_bf16, b1, w2, b2, sal, n_big)
            switch (np) {
                case 10: HALO(10); break;
                case 9: HALO(9); break;
                case 8: HALO(8); break;
                case 7: HALO(7); break;
                case 6: HALO(6); break;
                default: HALO(5); break;
            }
#undef HALO
            SSLAM_CHECK_LAUNCH();
            return SSLAM_OK;
        }
    }
    if (hs == 256) {
        const int variant = (int)sslam_knob(KNOB_CONVBF_VARIANT, 2);   // measured: 0: 1.08 ms, 1: 1.28 ms, 2: 0.96 ms / 613 frames
        if (variant == 1) {
            const int n_tiles = (int)((rows + 127) / 128);
            hipLaunchKernelGGL((selector_bf16_kernel<1, 4, 4>), dim3(n_tiles), dim3(256), 0, st, (const bf16 *)feat_bf16, (int)rows, G,
                               (const bf16 *)w1_packed_bf16, b1, w2, b2, sal, n_tiles);
        } else if (variant == 2) {
            const int n_tiles = (int)((rows + 255) / 256);
            hipLaunchKernelGGL((selector_bf16_kernel<2, 4, 4>), dim3(n_tiles), dim3(512), 0, st, (const bf16 *)feat_bf16, (int)rows, G,
                               (const bf16 *)w1_packed_bf16, b1, w2, b2, sal, n_tiles);
        } else {
            const int n_tiles = (int)((rows + 127) / 128);
            hipLaunchKernelGGL((selector_bf16_kernel<2, 4, 2>), dim3(n_tiles), dim3(512), 0, st, (const bf16 *)feat_bf16, (int)rows, G,
                               (const bf16 *)w1_packed_bf16, b1, w2, b2, sal, n_tiles);
        }
    } else if (hs == 128) {
        const int n_tiles = (int)((rows + 255) / 256);
        hipLaunchKernelGGL((selector_bf16_kernel<4, 2, 2>), dim3(n_tiles), dim3(512), 0, st, (const bf16 *)feat_bf16, (int)rows, G,
                           (const bf16 *)w1_packed_bf16, b1, w2, b2, sal, n_tiles);
    } else {
        return SSLAM_E_UNSUPPORTED;
    }
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

// vit.hip - A1: DINOv3 ViT-S/16 forward, the stage the reference delegates to the third-party `timm`
// package (call site semantic-slam/models/dino_backbone.py:85).  Architecture restated from the public DINOv3
// definition (see sslam_amd/vit.py, the torch-op reference this is checked against).
//
// Numerics: bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; LayerNorm, softmax, RoPE, GELU, LayerScale
// and the residual stream stay fp32.  This stage is tolerance-checked (it replaces an fp32 third-party model whose
// pretrained weights are a remote fetch), unlike the bit-exact authored path that consumes its tokens.
//
// Per layer: LN1 -> [QKV GEMM + bias + RoPE + 1/sqrt(d) scale, scattered to (B, H, T, 64)] -> flash-style attention ->
// [o_proj GEMM + bias, LayerScale, residual add in place] -> LN2 -> [up GEMM + bias + exact GELU] ->
// [down GEMM + bias, LayerScale, residual add].  One "A-resident" bf16 GEMM (below) with the epilogue as a template
// functor; weights are pre-packed into MFMA B-fragment order.  Attention keeps the query on the lane (S^T = K.Q^T), so the
// running max / sum are lane-local and the probability tile is fed to the P.V MFMA straight from the accumulator
// registers (no LDS round trip for P); K is rotated and V transposed once per layer by kv_prep_kernel.
#include "common.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's uint4 struct did not)

namespace {

constexpr int VD = 384, VH = 6, VHD = 64, VMLP = 1536, VPATCH = 16, VPREFIX = 5, VLAYERS = 12;
constexpr float LOG2E = 1.44269504088896341f;

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------- A-resident GEMM (K in chunks of 384)
// C (M x N) = A (M x K, bf16 row-major) . W^T, W pre-packed in MFMA-fragment order [N/32][K/16][2][32][8] bf16
// (fragment element = lane: half h = lane>>5 holds k = 8h..8h+7 of row lane&31).
// Work item = (128-row tile, column group of 128*S); persistent workgroups of 4 waves walk the items.  The item's
// 128 x 384 A chunk lives in LDS (392-element rows); wave w owns the 32-column slices {s*4 + w} of the group, i.e.
// 128 rows x 32 columns per slice = 4 MFMA tiles per B fragment (the reuse that keeps the L1 path under its 64 B/clk).
// B fragments go global -> registers (1 KB coalesced per wave-instruction, 4-deep ring); the NEXT chunk's A rows are
// prefetched into registers one 16-B piece per k-step, so the main loop has no barrier and chunk hand-over costs two.
#ifndef SSLAM_ARES_MT
#define SSLAM_ARES_MT 2
#endif
#ifndef SSLAM_ARES_OCC
#define SSLAM_ARES_OCC 3
#endif
constexpr int ARES_OCC = SSLAM_ARES_OCC;   // workgroups per CU the register budget is sized for (MT == 2)
constexpr int MT = SSLAM_ARES_MT;   // M tiles per wave: 2 -> 64-row workgroups, 50 KB LDS, 3 co-resident per CU (latencies of one
                                    // overlap the MFMAs of the others); 4 -> 128 rows, 1 per CU (measured slower: nothing overlaps)
constexpr int RM128 = 32 * MT, KC = 384, ALD2 = KC + 8, KSTEPS = KC / 16, A_PIECES = RM128 * (KC / 8) / 256;   // pieces / thread
#ifndef SSLAM_BRING
#define SSLAM_BRING 8
#endif
constexpr int BRING = SSLAM_BRING;   // B fragments in flight per wave (1 KB each): Little's law needs >= 32 KB per CU

template <int S, class Epi>
__global__ __launch_bounds__(256, MT <= 3 ? ARES_OCC : 1) void gemm_ares_kernel(const bf16 *__restrict__ A, const bf16 *__restrict__ Wp, int M, int N,
                                                            int K, int n_items, int n_groups, Epi epi) {
    __shared__ __attribute__((aligned(16))) bf16 As[RM128 * ALD2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int nchunk = K / KC;
    const int kfr = K / 16;                                  // k-steps per full row of W
    // A staging map: piece p of thread t = row (t>>4) + 16*(p % RB), 16-B column (t&15) + 16*(p / RB), RB = rows/16:
    // 16 consecutive threads move 256 contiguous bytes of one row; no per-piece address registers are kept.
    const int srow = tid >> 4, scol = (tid & 15) * 8;
    u32x4 pre[A_PIECES];
#define A_SRC(rt_, kc_, p_)                                                                                \
    reinterpret_cast<const u32x4 *>(A + (long long)min((rt_) * RM128 + srow + 16 * ((p_) % RB), M - 1) * K + \
                                    (kc_) * KC + scol + 128 * ((p_) / RB))
#define A_DST(p_) reinterpret_cast<u32x4 *>(As + (srow + 16 * ((p_) % RB)) * ALD2 + scol + 128 * ((p_) / RB))
    constexpr int RB = RM128 / 16;

    // contiguous, balanced item range per workgroup: consecutive items share their row tile, so for K = 384 the A tile is
    // staged once per row tile and reused for all its column groups
    const int per = n_items / gridDim.x, extra = n_items % gridDim.x, b = blockIdx.x;
    const int item_lo = b * per + min(b, extra), item_hi = item_lo + per + (b < extra ? 1 : 0);
    if (item_lo >= item_hi) return;
    int item = item_lo;
#pragma unroll
    for (int p = 0; p < A_PIECES; p++) pre[p] = *A_SRC(item / n_groups, 0, p);
#pragma unroll
    for (int p = 0; p < A_PIECES; p++) *A_DST(p) = pre[p];
    __syncthreads();

    const bf16 *Ab = As + r * ALD2 + 8 * h;
    for (; item < item_hi; item++) {
        const int ng = item % n_groups, rt = item / n_groups;
        f32x16 acc[S][MT];
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[s][mt][e] = 0.0f;

        for (int kc = 0; kc < nchunk; kc++) {
            // what comes after this chunk (next K chunk of the item, or the first chunk of the next item)?
            const bool last_kc = kc + 1 == nchunk;
            const int nitem = last_kc ? item + 1 : item, nkc = last_kc ? 0 : kc + 1;
            // the next (item, chunk) needs a different A chunk unless K is one chunk and the row tile stays the same
            const int nrt = nitem / n_groups;
            const bool have_next = nitem < item_hi && !(nchunk == 1 && nrt == rt);
#pragma unroll
            for (int s = 0; s < S; s++) {
                const int slice = ng * (4 * S) + s * 4 + wave;                  // 32-column slice of N
                const bf16x8 *bsrc = reinterpret_cast<const bf16x8 *>(Wp) + ((long long)slice * kfr + kc * KSTEPS) * 64 + lane;
                bf16x8 bq[BRING];
#pragma unroll
                for (int i = 0; i < BRING; i++) bq[i] = bsrc[i * 64];
                bf16x8 an[MT], ac[MT];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) an[mt] = *reinterpret_cast<const bf16x8 *>(Ab + mt * 32 * ALD2);
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks++) {
                    // software pipeline, program order = issue order: A fragments of step ks+1 (LDS), the B fragment of
                    // step ks+4 and one piece of the next chunk (global) are requested BEFORE the four MFMAs of step ks
                    const bf16x8 bnow = bq[ks % BRING];
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) ac[mt] = an[mt];
                    if (ks + 1 < KSTEPS) {
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
                            an[mt] = *reinterpret_cast<const bf16x8 *>(Ab + mt * 32 * ALD2 + (ks + 1) * 16);
                    }
                    if (ks + BRING < KSTEPS) bq[ks % BRING] = bsrc[(ks + BRING) * 64];
                    if (s == 0 && have_next && ks < A_PIECES) pre[ks] = *A_SRC(nrt, nkc, ks);
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[s][mt] = mfma_bf16(ac[mt], bnow, acc[s][mt]);
                    __builtin_amdgcn_sched_barrier(0);      // keep this order; do not hoist later LDS reads up here
                }
            }
            if (have_next) {
                __syncthreads();                   // every wave is done reading this chunk
#pragma unroll
                for (int p = 0; p < A_PIECES; p++) *A_DST(p) = pre[p];
            }
            if (last_kc) {
                // epilogue of the item (registers only)
#pragma unroll
                for (int s = 0; s < S; s++) {
                    epi.slice(acc[s], rt * RM128, (ng * (4 * S) + s * 4 + wave) * 32, r, h, M);
                }
            }
            if (have_next) __syncthreads();
        }
    }
#undef A_SRC
#undef A_DST
}

// x[row, col] += ls[col] * (acc + bias[col])          (o_proj / down_proj: LayerScale + residual, fp32 stream)
struct EpiResidual {
    const float *bias, *ls;
    float *x;
    __device__ __forceinline__ void slice(f32x16 (&acc)[MT], int row0, int col0, int r, int h, int M) const {
        const int col = col0 + r;
        const float b = bias[col], l = ls[col];
        int loff = 4 * h * VD + col;                 // lane part of the address; kept opaque so that the 32 per-element
        asm volatile("" : "+v"(loff));               // addresses are formed at use instead of being hoisted into 64 VGPRs
        float *base = x + (long long)row0 * VD + loff;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int o = mt * 32 + (e & 3) + 8 * (e >> 2);
                if (row0 + o + 4 * h < M) {
                    float *p = base + o * VD;
                    *p = *p + l * (acc[mt][e] + b);
                }
            }
    }
};

// exact-erf GELU (torch's default) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// rounding of the result): 15 VALU operations, two of them transcendental, instead of libm's ~45 - the up_proj epilogue
// is VALU-bound (64 x 1536 activations per 64-row tile against 48 MFMAs per wave and item)
__device__ __forceinline__ float gelu_erf(float v) {
    const float ax = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = __builtin_fmaf(p, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __builtin_amdgcn_exp2f(ax * ax * -LOG2E);
    const float erf_abs = __builtin_fmaf(-p, e, 1.0f);
    return v * __builtin_fmaf(0.5f, __builtin_copysignf(erf_abs, v), 0.5f);
}

// out[row, col] = gelu(acc + bias[col]) as bf16       (up_proj; exact erf GELU = torch's default)
struct EpiGelu {
    const float *bias;
    bf16 *out;
    int ldo;
    __device__ __forceinline__ void slice(f32x16 (&acc)[MT], int row0, int col0, int r, int h, int M) const {
        const int col = col0 + r;
        const float b = bias[col];
        int loff = 4 * h * ldo + col;
        asm volatile("" : "+v"(loff));
        bf16 *base = out + (long long)row0 * ldo + loff;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int o = mt * 32 + (e & 3) + 8 * (e >> 2);
                if (row0 + o + 4 * h < M) {
                    const float v = acc[mt][e] + b;
                    base[(long long)o * ldo] = (bf16)gelu_erf(v);
                }
            }
    }
};

// patch embedding: row = frame*cells + patch -> x[frame*T + 5 + patch, col] = acc + bias[col]
struct EpiPatch {
    const float *bias;
    float *x;
    int cells, T;
    __device__ __forceinline__ void slice(f32x16 (&acc)[MT], int row0, int col0, int r, int h, int M) const {
        const int col = col0 + r;
        const float b = bias[col];
        const int f0 = row0 / cells, p0 = row0 - f0 * cells;
        int loff = 4 * h * VD + col;
        asm volatile("" : "+v"(loff));
        float *base = x + ((long long)f0 * T + VPREFIX + p0) * VD + loff;
        const int wrap = cells - p0 - 4 * h;           // rows at in-tile offset >= wrap belong to the next frame(s)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int o = mt * 32 + (e & 3) + 8 * (e >> 2);
                if (row0 + o + 4 * h < M) {
                    int skip = 0;                      // each frame boundary crossed skips that frame's 5 prefix rows
                    if (o >= wrap) skip = VPREFIX;
                    if (o >= wrap + cells) skip = 2 * VPREFIX;
                    base[(long long)(o + skip) * VD] = acc[mt][e] + b;
                }
            }
    }
};

// QKV: + bias, RoPE on the patch tokens of q and k (pairs (d, d+32) are the two N tiles of this wave), q *= 1/8,
// scatter to (B, H, T, 64) bf16.  The wave's 64 columns are exactly one head of one of q / k / v.
struct EpiQKV {
    const float *bias, *cosb, *sinb;
    bf16 *q, *k, *v;
    int T;
    // slice form: bias (+ 1/8 for q), scatter to (B, H, T, 64); RoPE is applied by the attention kernel on load
    __device__ __forceinline__ void slice(f32x16 (&acc)[MT], int row0, int col0, int r, int h, int M) const {
        const int which = col0 / VD, head = (col0 % VD) / VHD, d = (col0 % VHD) + r;
        bf16 *dst = which == 0 ? q : (which == 1 ? k : v);
        const float b = bias[col0 + r], sc = which == 0 ? 0.125f * LOG2E : 1.0f;   // q: 1/sqrt(64) and the exp2 domain
        const int f0 = row0 / T, t0 = row0 - f0 * T;      // one division per tile: a tile spans <= 2 frames (T > 128)
        int loff = 4 * h * VHD + d;
        asm volatile("" : "+v"(loff));
        bf16 *base = dst + (((long long)f0 * VH + head) * T + t0) * VHD + loff;
        const long long fstep = (long long)(VH - 1) * T * VHD;   // extra offset once the row wraps into frame f0 + 1
        const int wrap = T - t0 - 4 * h;                          // first in-tile row offset o that belongs to frame f0 + 1
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int o = mt * 32 + (e & 3) + 8 * (e >> 2);
                if (row0 + o + 4 * h < M) base[(long long)o * VHD + (o >= wrap ? fstep : 0)] = (bf16)((acc[mt][e] + b) * sc);
            }
    }
};

// --------------------------------------------------------------------------------------------- LayerNorm
// one wave per row of 384: lane l holds elements l*2 + 128*j .. (float2 x 3); fp32 two-pass statistics
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                       const float *__restrict__ b, float eps, long long rows,
                                                       void *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = x + row * VD;
    float2 v[3];
#pragma unroll
    for (int j = 0; j < 3; j++) v[j] = *reinterpret_cast<const float2 *>(p + 128 * j + 2 * lane);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) s += v[j].x + v[j].y;
    const float mean = bfly64(s) * (1.0f / VD);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float dx = v[j].x - mean, dy = v[j].y - mean;
        s2 += dx * dx + dy * dy;
    }
    const float rstd = rsqrtf(bfly64(s2) * (1.0f / VD) + eps);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int c = 128 * j + 2 * lane;
        const float y0 = (v[j].x - mean) * rstd * g[c] + b[c], y1 = (v[j].y - mean) * rstd * g[c + 1] + b[c + 1];
        if (OUT_BF16) {
            bf16 *o = reinterpret_cast<bf16 *>(out) + row * VD + c;
            o[0] = (bf16)y0;
            o[1] = (bf16)y1;
        } else {
            *reinterpret_cast<float2 *>(reinterpret_cast<float *>(out) + row * VD + c) = make_float2(y0, y1);
        }
    }
}

// ------------------------------------------------------------------------------------- patches / prefix rows
// (B, 3, S, S) fp32 -> (B*G*G, 768) bf16 with k = c*256 + ky*16 + kx   (nn.Conv2d(3, 384, 16, 16) as a GEMM)
__global__ __launch_bounds__(256) void im2patch_kernel(const float *__restrict__ img, int S, long long items, bf16 *__restrict__ out) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;   // one 8-pixel run
    if (it >= items) return;
    const int G = S / VPATCH;
    const int run = (int)(it % 96);                 // 768 / 8 runs per patch
    const long long patch = it / 96;
    const int cells = G * G;
    const long long f = patch / cells;
    const int pc = (int)(patch % cells), py = pc / G, px = pc % G;
    const int c = run / 32, ky = (run % 32) / 2, kx0 = (run & 1) * 8;
    const float *src = img + ((f * 3 + c) * S + (py * VPATCH + ky)) * (long long)S + px * VPATCH + kx0;
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    bf16x8 o;
    o[0] = (bf16)a.x; o[1] = (bf16)a.y; o[2] = (bf16)a.z; o[3] = (bf16)a.w;
    o[4] = (bf16)b.x; o[5] = (bf16)b.y; o[6] = (bf16)b.z; o[7] = (bf16)b.w;
    *reinterpret_cast<bf16x8 *>(out + patch * 768 + run * 8) = o;
}

__global__ __launch_bounds__(256) void prefix_rows_kernel(const float *__restrict__ prefix, int T, long long items, float *__restrict__ x) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;   // items = B * 5 * 384
    if (it >= items) return;
    const int c = (int)(it % VD), i = (int)((it / VD) % VPREFIX);
    const long long f = it / (VD * VPREFIX);
    x[(f * T + i) * VD + c] = prefix[i * VD + c];
}

// --------------------------------------------------------------------------------------------- attention
constexpr int AQ = 128, AKT = 64, KLD = 72, VLD = 76;

// RoPE (q' = q*cos + rotate_half(q)*sin on the patch tokens): the partner of element d is d +- 32.  The tables satisfy
// cos[d] == cos[d+32] (angles tiled twice), so one (cos, sin) octet serves both halves.
__device__ __forceinline__ void rope8(u32x4 &lo, u32x4 &hi, const float4 &c0, const float4 &c1, const float4 &s0, const float4 &s1) {
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    bf16x8 a = __builtin_bit_cast(bf16x8, lo), b = __builtin_bit_cast(bf16x8, hi);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float x0 = (float)a[j], x1 = (float)b[j];
        a[j] = (bf16)(x0 * cs[j] - x1 * sn[j]);
        b[j] = (bf16)(x1 * cs[j] + x0 * sn[j]);
    }
    lo = __builtin_bit_cast(u32x4, a);
    hi = __builtin_bit_cast(u32x4, b);
}

// K/V preparation, once per (frame, head, 64-token tile) instead of once per query tile inside attention:
// RoPE on K in place, and V written transposed + zero padded: vt (B, H, 64, Tp), Tp = 64 * ceil(T / 64).
__global__ __launch_bounds__(256) void kv_prep_kernel(bf16 *__restrict__ k, const bf16 *__restrict__ v, bf16 *__restrict__ vt,
                                                       const float *__restrict__ cosb, const float *__restrict__ sinb, int T, int Tp) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 66];      // [token][d], 66-element rows
    const int tid = threadIdx.x, head = blockIdx.y;
    const long long f = blockIdx.z;
    const long long bh = (f * VH + head) * (long long)T;
    const int t0 = blockIdx.x * 64;
    const int key = t0 + (tid >> 2), pr = tid & 3;
    if (key < T) {
        const long long off = (bh + key) * VHD + pr * 8;
        if (key >= VPREFIX) {
            u32x4 lo = *reinterpret_cast<const u32x4 *>(k + off), hi = *reinterpret_cast<const u32x4 *>(k + off + 32);
            const float *c = cosb + (long long)(key - VPREFIX) * VHD + pr * 8, *sn = sinb + (long long)(key - VPREFIX) * VHD + pr * 8;
            rope8(lo, hi, *reinterpret_cast<const float4 *>(c), *reinterpret_cast<const float4 *>(c + 4),
                  *reinterpret_cast<const float4 *>(sn), *reinterpret_cast<const float4 *>(sn + 4));
            *reinterpret_cast<u32x4 *>(k + off) = lo;
            *reinterpret_cast<u32x4 *>(k + off + 32) = hi;
        }
    }
    // V tile -> LDS (natural), then each thread gathers 16 tokens of one d and writes two 16-B pieces of vt[d][t0..]
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = tid + 256 * i, tk = c >> 3, dc = c & 7;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (t0 + tk < T) val = *reinterpret_cast<const u32x4 *>(v + (bh + t0 + tk) * VHD + dc * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) tile[tk * 66 + dc * 8 + j] = (unsigned short)(val[j >> 1] >> (16 * (j & 1)));
    }
    __syncthreads();
    {
        const int d = tid >> 2, part = tid & 3;
        u32x4 o[2];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const unsigned a = tile[(part * 16 + j) * 66 + d], b = tile[(part * 16 + j + 1) * 66 + d];
            o[j >> 3][(j >> 1) & 3] = a | (b << 16);
        }
        bf16 *dst = vt + ((f * VH + head) * VHD + d) * (long long)Tp + t0 + part * 16;
        *reinterpret_cast<u32x4 *>(dst) = o[0];
        *reinterpret_cast<u32x4 *>(dst + 8) = o[1];
    }
}

// Flash-style attention, one workgroup = 128 queries of one (frame, head), 4 waves x 32 queries, keys in tiles of 64.
// q arrives pre-scaled by log2(e)/sqrt(64) (QKV epilogue), k already rotated, vt already transposed.
__global__ __launch_bounds__(256) void attn_kernel(const bf16 *__restrict__ q, const bf16 *__restrict__ k,
                                                    const bf16 *__restrict__ vt, const float *__restrict__ cosb,
                                                    const float *__restrict__ sinb, bf16 *__restrict__ o, int T, int Tp) {
    __shared__ __attribute__((aligned(16))) bf16 Ks[2][AKT * KLD];
    __shared__ __attribute__((aligned(16))) bf16 Vt[2][VHD * VLD];
    __shared__ __attribute__((aligned(16))) float bc[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const long long f = blockIdx.z;
    const long long bh = (f * VH + head) * (long long)T;
    const bf16 *vth = vt + (f * VH + head) * (long long)VHD * Tp;
    const int i0 = blockIdx.x * AQ + wave * 32;
    const int qi = min(i0 + r, T - 1);

    bf16x8 qf[4];
    {
        u32x4 qraw[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qraw[ks] = *reinterpret_cast<const u32x4 *>(q + (bh + qi) * VHD + ks * 16 + 8 * h);
        if (qi >= VPREFIX) {   // element d of this lane's fragment ks is 16*ks + 8h + j; its partner d + 32 sits in fragment ks + 2
            const float *c = cosb + (long long)(qi - VPREFIX) * VHD + 8 * h, *sn = sinb + (long long)(qi - VPREFIX) * VHD + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
                rope8(qraw[ks], qraw[ks + 2], *reinterpret_cast<const float4 *>(c + 16 * ks), *reinterpret_cast<const float4 *>(c + 16 * ks + 4),
                      *reinterpret_cast<const float4 *>(sn + 16 * ks), *reinterpret_cast<const float4 *>(sn + 16 * ks + 4));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qf[ks] = __builtin_bit_cast(bf16x8, qraw[ks]);
    }

    // staging: 512 16-byte pieces of K (key rows) and 512 of V^T (d rows) per tile, two of each per thread
    u32x4 rk[2], rv[2];
#define A_LOAD(kt)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                  \
        const int c = tid + 256 * i, row = c >> 3, pc = c & 7;                                       \
        const int key = (kt) * AKT + row;                                                            \
        const unsigned msk = key < T ? 0xffffffffu : 0u;                                             \
        rk[i] = *reinterpret_cast<const u32x4 *>(k + (bh + min(key, T - 1)) * VHD + pc * 8) & msk;   \
        rv[i] = *reinterpret_cast<const u32x4 *>(vth + (long long)row * Tp + (kt) * AKT + pc * 8);   \
    }
#define A_STORE(buf)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                  \
        const int c = tid + 256 * i, row = c >> 3, pc = c & 7;                                       \
        *reinterpret_cast<u32x4 *>(&Ks[buf][row * KLD + pc * 8]) = rk[i];                            \
        unsigned long long *vd = reinterpret_cast<unsigned long long *>(&Vt[buf][row * VLD + pc * 8]); \
        vd[0] = (unsigned long long)rv[i][0] | ((unsigned long long)rv[i][1] << 32);                 \
        vd[1] = (unsigned long long)rv[i][2] | ((unsigned long long)rv[i][3] << 32);                 \
    }

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[dt][e] = 0.0f;
    float m = -INFINITY, l = 0.0f;

    const int ntile = (T + AKT - 1) / AKT;
    A_LOAD(0);
    A_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < ntile; kt++) {
        if (kt + 1 < ntile) A_LOAD(kt + 1);
        const bf16 *Kb = &Ks[kt & 1][0];
        const bf16 *Vb = &Vt[kt & 1][0];
        // S^T tiles: rows = keys, cols = queries (already in the log2 domain)
        f32x16 st[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int e = 0; e < 16; e++) st[j][e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(Kb + (j * 32 + r) * KLD + ks * 16 + 8 * h);
                st[j] = mfma_bf16(ka, qf[ks], st[j]);
            }
        }
        if (kt + 1 == ntile) {       // only the last tile has keys beyond T
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (kt * AKT + j * 32 + crow(e, h) >= T) st[j][e] = -INFINITY;
        }
        float mt = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int e = 1; e < 16; e++) mt = fmaxf(mt, fmaxf(st[0][e], st[1][e]));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt);
        float rs = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f(st[j][e] - mn);
                st[j][e] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32);
        if (__any(mn != m)) {
            // rescale O: alpha lives on the query's lane, O rows are queries -> broadcast through LDS (wave-local)
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
            if (h == 0) bc[wave][r] = alpha;
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS write has landed
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const float4 a4 = *reinterpret_cast<const float4 *>(&bc[wave][8 * g4 + 4 * h]);
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    oacc[dt][4 * g4 + 0] *= a4.x;
                    oacc[dt][4 * g4 + 1] *= a4.y;
                    oacc[dt][4 * g4 + 2] *= a4.z;
                    oacc[dt][4 * g4 + 3] *= a4.w;
                }
            }
            m = mn;
        }
        l += rs;
        // P.V: P^T tile registers 8*s2..8*s2+7 are the A fragment of k-step s2 (key order 16*s2 + 8*(j>>2) + 4h + (j&3))
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                bf16x8 pa;
#pragma unroll
                for (int e = 0; e < 8; e++) pa[e] = (bf16)st[j][8 * s2 + e];
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const bf16 *vp = Vb + (dt * 32 + r) * VLD + j * 32 + 16 * s2 + 4 * h;
                    const bf16x4 v0 = *reinterpret_cast<const bf16x4 *>(vp), v1 = *reinterpret_cast<const bf16x4 *>(vp + 8);
                    bf16x8 vb;
                    vb[0] = v0[0]; vb[1] = v0[1]; vb[2] = v0[2]; vb[3] = v0[3];
                    vb[4] = v1[0]; vb[5] = v1[1]; vb[6] = v1[2]; vb[7] = v1[3];
                    oacc[dt] = mfma_bf16(pa, vb, oacc[dt]);
                }
            }
        if (kt + 1 < ntile) A_STORE((kt + 1) & 1);
        __syncthreads();
    }
#undef A_LOAD
#undef A_STORE
    // normalise by the row sums (broadcast like alpha) and write (B, T, 384) bf16
    if (h == 0) bc[wave][r] = 1.0f / l;
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
        const float4 a4 = *reinterpret_cast<const float4 *>(&bc[wave][8 * g4 + 4 * h]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int row = i0 + 8 * g4 + 4 * h + u;
            if (row < T) {
#pragma unroll
                for (int dt = 0; dt < 2; dt++)
                    o[((long long)f * T + row) * VD + head * VHD + dt * 32 + r] = (bf16)(oacc[dt][4 * g4 + u] * av[u]);
            }
        }
    }
}

// A-resident persistent GEMM: one workgroup per CU walks (row tile, column group) items
template <int S, class Epi>
void launch_ares(const bf16 *A, const bf16 *Wp, long long M, int N, int K, Epi epi, hipStream_t st) {
    const int n_groups = N / (128 * S), n_tiles = (int)((M + RM128 - 1) / RM128), n_items = n_tiles * n_groups;
    const int slots = MT <= 3 ? 256 * ARES_OCC : 256;                  // persistent workgroups: 3 per CU at 64 rows, 1 at 128
    const int grid = n_tiles < slots ? n_tiles : slots;
    hipLaunchKernelGGL((gemm_ares_kernel<S, Epi>), dim3(grid), dim3(256), 0, st, A, Wp, (int)M, N, K, n_items, n_groups, epi);
}

size_t ws_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" long long sslam_vit_workspace_bytes(int n_frames, int size) {
    if (n_frames <= 0 || size <= 0 || size % VPATCH) return SSLAM_E_INVALID;
    const long long G = size / VPATCH, T = G * G + VPREFIX, rows = (long long)n_frames * T;
    size_t b = 0;
    b += ws_align(rows * VD * 4);                 // x   fp32 residual stream
    b += ws_align(rows * VD * 2);                 // y   bf16 LN output / attention output
    b += ws_align(rows * VD * 2 * 3);             // q, k, v bf16
    b += ws_align((long long)n_frames * VD * ((T + 63) / 64 * 64) * 2);   // v transposed + padded
    b += ws_align(rows * VMLP * 2);               // h   bf16 MLP hidden (also the patch matrix)
    return (long long)b;
}

extern "C" int sslam_vit_forward(const float *images_chw, int n_frames, int size, const sslam_vit_weights_t *w, void *workspace,
                                 long long workspace_bytes, float *tokens_out, void *stream) {
    if (!images_chw || !w || !workspace || !tokens_out || n_frames <= 0 || size <= 0 || size % VPATCH) return SSLAM_E_INVALID;
    if (workspace_bytes < sslam_vit_workspace_bytes(n_frames, size)) return SSLAM_E_INVALID;
    if (((uintptr_t)images_chw | (uintptr_t)workspace | (uintptr_t)tokens_out) & 15) return SSLAM_E_INVALID;
    const int G = size / VPATCH, cells = G * G, T = cells + VPREFIX;
    const long long rows = (long long)n_frames * T, prow = (long long)n_frames * cells;
    if (rows * VMLP > 0x7fffffffLL * 64) return SSLAM_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    char *p = (char *)workspace;
    float *x = (float *)p;            p += ws_align(rows * VD * 4);
    bf16 *y = (bf16 *)p;              p += ws_align(rows * VD * 2);
    bf16 *q = (bf16 *)p;              bf16 *k = q + rows * VD, *v = k + rows * VD;   p += ws_align(rows * VD * 2 * 3);
    const int Tp = (T + 63) / 64 * 64;
    bf16 *vt = (bf16 *)p;             p += ws_align((long long)n_frames * VD * Tp * 2);
    bf16 *hbuf = (bf16 *)p;

    // patch embedding + prefix tokens
    {
        const long long items = prow * 96;
        hipLaunchKernelGGL(im2patch_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, images_chw, size, items, hbuf);
        g_sslam_launches++;
        launch_ares<1>(hbuf, (const bf16 *)w->patch_w, prow, VD, 768, EpiPatch{w->patch_b, x, cells, T}, st);
        g_sslam_launches++;
        const long long pi = (long long)n_frames * VPREFIX * VD;
        hipLaunchKernelGGL(prefix_rows_kernel, dim3((unsigned)((pi + 255) / 256)), dim3(256), 0, st, w->prefix, T, pi, x);
        g_sslam_launches++;
    }
    const unsigned ln_grid = (unsigned)((rows + 3) / 4);
    for (int L = 0; L < VLAYERS; L++) {
        const sslam_vit_layer_t &ly = w->layer[L];
        hipLaunchKernelGGL(ln_rows_kernel<true>, dim3(ln_grid), dim3(256), 0, st, x, ly.ln1_g, ly.ln1_b, 1e-5f, rows, (void *)y);
        launch_ares<1>(y, (const bf16 *)ly.wqkv, rows, 3 * VD, VD, EpiQKV{ly.bqkv, w->rope_cos, w->rope_sin, q, k, v, T}, st);
        hipLaunchKernelGGL(kv_prep_kernel, dim3(Tp / 64, VH, n_frames), dim3(256), 0, st, k, v, vt, w->rope_cos, w->rope_sin, T, Tp);
        hipLaunchKernelGGL(attn_kernel, dim3((T + AQ - 1) / AQ, VH, n_frames), dim3(256), 0, st, q, k, vt, w->rope_cos, w->rope_sin, y, T, Tp);
        launch_ares<1>(y, (const bf16 *)ly.wo, rows, VD, VD, EpiResidual{ly.bo, ly.ls1, x}, st);
        hipLaunchKernelGGL(ln_rows_kernel<true>, dim3(ln_grid), dim3(256), 0, st, x, ly.ln2_g, ly.ln2_b, 1e-5f, rows, (void *)y);
        launch_ares<1>(y, (const bf16 *)ly.wup, rows, VMLP, VD, EpiGelu{ly.bup, hbuf, VMLP}, st);
        launch_ares<1>(hbuf, (const bf16 *)ly.wdown, rows, VD, VMLP, EpiResidual{ly.bdown, ly.ls2, x}, st);
        g_sslam_launches += 8;
    }
    hipLaunchKernelGGL(ln_rows_kernel<false>, dim3(ln_grid), dim3(256), 0, st, x, w->norm_g, w->norm_b, 1e-5f, rows, (void *)tokens_out);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

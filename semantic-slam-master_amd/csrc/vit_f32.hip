// vit_f32.hip - A1 with the REFERENCE'S numerics: the DINOv3 ViT-S/16 forward in fp32 (the reference's timm model is fp32,
// dino_backbone.py:85) on v_mfma_f32_32x32x2_f32 - every contraction an fp32 fma chain in increasing k, fp32 LayerNorm,
// softmax, GELU (erf) and residual stream.  Tolerance-level parity with the eager torch fp32 definition (sslam_amd/vit.py)
// at ~1e-5 relative (summation order differs); vit.hip is the bf16-operand throughput form of the same forward.
//
// The fp32 matrix pipe runs at 1/16 of the bf16 rate (64 cycles per 32x32x2 MFMA and SIMD), so every kernel here is bound
// by its MFMAs and nothing else needs to be clever: GEMM operands go global -> registers -> LDS (KP8 image, one
// ds_read_b128 per four MFMA steps) with one tile of look-ahead; LayerNorm is its own pass; the MLP's hidden activation
// makes a round trip through HBM (10 MB per frame and layer, ~3 % of the layer's matrix time).
//   gemm_f32_kernel<ALoad, Epi>   128 x 128 tile per workgroup, 4 waves x (64 x 64), K in steps of 32
//   attn_f32_kernel               flash-style: a wave owns 32 queries (registers), keys in tiles of 32 through LDS;
//                                 S^T = K Q^T and O^T = V^T P^T keep every per-query quantity on one lane pair, and the
//                                 accumulator registers of S^T ARE the B operand of the second product
//   ln_rows_f32_kernel            two-pass LayerNorm, one wave per row
#include "common.h"

namespace {

constexpr int FD = 384, FH = 6, FHD = 64, FMLP = 1536, FPATCH = 16, FPREFIX = 5, FLAYERS = 12;
constexpr int GBM = 128, GBN = 128, GBK = 32, GLDK = 36;     // LDS row stride 144 B: conflict-free ds_read_b128 / ds_write_b128

// ------------------------------------------------------------------------------------------------ A-operand views
struct ARows {                // a row-major fp32 matrix
    const float *p;
    int ld;
    __device__ __forceinline__ const float *at(long long row, int k) const { return p + row * ld + k; }
};
struct AIm2Patch {            // the (n * G * G, 768) patch matrix of a planar fp32 image, k = c * 256 + ky * 16 + kx, never materialised
    const float *img;
    int size, G;
    __device__ __forceinline__ const float *at(long long row, int k) const {
        const int cells = G * G;
        const long long f = row / cells;
        const int p = (int)(row - f * cells), py = p / G, px = p - py * G;
        const int c = k >> 8, ky = (k >> 4) & 15, kx = k & 15;
        return img + ((f * 3 + c) * size + py * FPATCH + ky) * size + px * FPATCH + kx;
    }
};

// ------------------------------------------------------------------------------------------------------ epilogues
// acc layout (A = activations, B = weights): register e of lane (r, h) is row crow(e, h), column r of a 32 x 32 tile
// Every epilogue first gathers what it needs (clamped rows: no predicate in front of a load, so the loads of all 64 outputs
// of a lane go out together), then computes, then stores under the row predicate: a load-wait-store chain per output
// costs a memory round trip each (64 of them measured as long as the whole k loop of a K = 384 GEMM).
struct EpiPatch {             // x[frame][5 + patch] = acc + bias
    const float *b;
    float *x;
    int cells, T;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[2][2], long long m0, int n0, int r, int h, long long M) const {
        const float b0 = b[n0 + r], b1 = b[n0 + 32 + r];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const unsigned row = (unsigned)(m0 + mt * 32 + crow(e, h));
                const unsigned f = row / (unsigned)cells;
                float *dst = x + ((long long)f * T + FPREFIX + (row - f * cells)) * FD + n0 + r;
                if (row < M) {
                    dst[0] = acc[mt][0][e] + b0;
                    dst[32] = acc[mt][1][e] + b1;
                }
            }
    }
    // the one-tile wave of the few-frame launch shape: acc[0][0] is rows m0 + crow(e, h), column n0 + r
    __device__ __forceinline__ void one_tile(const f32x16 &acc, long long m0, int n0, int r, int h, long long M) const {
        const float b0 = b[n0 + r];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const unsigned row = (unsigned)(m0 + crow(e, h));
            const unsigned f = row / (unsigned)cells;
            if (row < M) x[((long long)f * T + FPREFIX + (row - f * cells)) * FD + n0 + r] = acc[e] + b0;
        }
    }
};
// ---- epilogues of the per-layer GEMM (gemm_f32_rows_kernel evaluates the product TRANSPOSED: weights = A operand) ---------------
// acc layout there: register e of lane (r, h) in tile (ni, ri) is column n0 + 32 ni + crow(e, h) of row m0 + 32 ri + r - a lane owns
// a ROW and holds four consecutive columns per group of four registers, so everything it loads or stores is 16 bytes wide
// (dword stores cost ~6 x the time per byte of dwordx4 stores: the first form's 64 dword stores per lane made the epilogue of a
// 12-step GEMM as long as a third of its k loop).
struct TQKV {                 // + bias, RoPE on the patch tokens of q and k, scatter to (frame, head, token, 64)
    static constexpr bool TRANSPOSED = true;
    const float *b, *cosv, *sinv;
    float *q, *k, *v;
    int T;
    template <int RI, int NI>
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[2][2], long long m0, int n0, int r, int h, long long M) const {
        static_assert(NI == 2, "RoPE pairs (d, d + 32) live in the wave's two column tiles");
        const int which = n0 / FD, head = (n0 % FD) / FHD;            // a wave's 64 columns are exactly one head of q, k or v
        float *dst = which == 0 ? q : (which == 1 ? k : v);
        const bool rope = which < 2;
#pragma unroll
        for (int ri = 0; ri < RI; ri++) {
            const long long row = m0 + 32 * ri + r;
            if (row >= M) continue;
            const long long f = row / T;
            const int t = (int)(row - f * T);
            const bool rot = rope && t >= FPREFIX;                   // prefix tokens (and v) are not rotated
            const float *ct = cosv + (long long)(rot ? t - FPREFIX : 0) * FHD + 4 * h, *st = sinv + (long long)(rot ? t - FPREFIX : 0) * FHD + 4 * h;
            float *o = dst + ((f * FH + head) * T + t) * FHD + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                // d = 8 qd + 4 h + i in the first half of the head (tile ni = 0), d + 32 in the second (ni = 1): rotate_half pairs them
                const float4 b0 = *reinterpret_cast<const float4 *>(b + n0 + 8 * qd + 4 * h), b1 = *reinterpret_cast<const float4 *>(b + n0 + 32 + 8 * qd + 4 * h);
                // DINOv3's tables are [angles, angles] (the 32 angles tiled twice): columns d and d + 32 are equal, only 0..31 are read
                // (every 16-byte access of this epilogue touches 32 different lines - its cost is the NUMBER of such accesses)
                float4 c0 = make_float4(1.f, 1.f, 1.f, 1.f), s0 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rot) {
                    c0 = *reinterpret_cast<const float4 *>(ct + 8 * qd);
                    s0 = *reinterpret_cast<const float4 *>(st + 8 * qd);
                }
                const float4 c1 = c0, s1 = s0;
                const float v0x = acc[0][ri][4 * qd] + b0.x, v0y = acc[0][ri][4 * qd + 1] + b0.y, v0z = acc[0][ri][4 * qd + 2] + b0.z, v0w = acc[0][ri][4 * qd + 3] + b0.w;
                const float v1x = acc[1][ri][4 * qd] + b1.x, v1y = acc[1][ri][4 * qd + 1] + b1.y, v1z = acc[1][ri][4 * qd + 2] + b1.z, v1w = acc[1][ri][4 * qd + 3] + b1.w;
                // out[d] = x[d] cos[d] - x[d + 32] sin[d] (d < 32), x[d] cos[d] + x[d - 32] sin[d] (d >= 32)
                const float4 o0 = make_float4(v0x * c0.x - v1x * s0.x, v0y * c0.y - v1y * s0.y, v0z * c0.z - v1z * s0.z, v0w * c0.w - v1w * s0.w);
                const float4 o1 = make_float4(v1x * c1.x + v0x * s1.x, v1y * c1.y + v0y * s1.y, v1z * c1.z + v0z * s1.z, v1w * c1.w + v0w * s1.w);
                *reinterpret_cast<float4 *>(o + 8 * qd) = o0;
                *reinterpret_cast<float4 *>(o + 32 + 8 * qd) = o1;
            }
        }
    }
};
// The residual epilogue is a read-modify-write of x: with a row per lane every 16-byte access touches 32 different 128-byte lines
// (measured: 36 -> 69 k cycles for o_proj, 40 -> 141 k for the down projection), so these two GEMMs keep the product the other
// way round - register e of lane (r, h) in tile (mt, nt) is row 32 mt + crow(e, h), column 32 nt + r: dword accesses, but each
// instruction covers two whole 128-byte lines.
struct EpiResidual {          // x += ls * (acc + bias)
    static constexpr bool TRANSPOSED = false;
    const float *b, *ls;
    float *x;
    template <int RI, int NI>
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[2][2], long long m0, int n0, int r, int h, long long M) const {
        float bv[NI], lv[NI];
#pragma unroll
        for (int nt = 0; nt < NI; nt++) bv[nt] = b[n0 + 32 * nt + r], lv[nt] = ls[n0 + 32 * nt + r];
#pragma unroll
        for (int mt = 0; mt < RI; mt++) {            // 16 rows at a time: bounded temporaries
            float xv[NI][16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                long long row = m0 + mt * 32 + crow(e, h);
                if (row > M - 1) row = M - 1;
#pragma unroll
                for (int nt = 0; nt < NI; nt++) xv[nt][e] = x[row * FD + n0 + 32 * nt + r];
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const long long row = m0 + mt * 32 + crow(e, h);
                if (row < M) {
#pragma unroll
                    for (int nt = 0; nt < NI; nt++) x[row * FD + n0 + 32 * nt + r] = xv[nt][e] + lv[nt] * (acc[mt][nt][e] + bv[nt]);
                }
            }
        }
    }
};
// erf(x), branch-free: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute (fp32 erff: ~1e-7) - the library erff is two
// polynomial ranges behind a divergent branch, ~45 instructions and two exec-masked regions per value, 64 values per lane
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = __builtin_fmaf(p, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);      // exp(-x^2)
    const float y = __builtin_fmaf(-p * t, e, 1.0f);
    return __builtin_copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752f)); }
struct TGelu {                // hidden = gelu(acc + bias), the erf form (torch F.gelu default)
    static constexpr bool TRANSPOSED = true;
    const float *b;
    float *hid;
    template <int RI, int NI>
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[2][2], long long m0, int n0, int r, int h, long long M) const {
        static_assert(NI == 2, "two column tiles per wave");
#pragma unroll
        for (int ri = 0; ri < RI; ri++) {
            const long long row = m0 + 32 * ri + r;
            if (row >= M) continue;
            float *hr = hid + row * FMLP + n0 + 4 * h;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    const float4 bv = *reinterpret_cast<const float4 *>(b + n0 + 32 * ni + 8 * qd + 4 * h);
                    float4 o;
                    o.x = gelu_erf(acc[ni][ri][4 * qd] + bv.x);
                    o.y = gelu_erf(acc[ni][ri][4 * qd + 1] + bv.y);
                    o.z = gelu_erf(acc[ni][ri][4 * qd + 2] + bv.z);
                    o.w = gelu_erf(acc[ni][ri][4 * qd + 3] + bv.w);
                    *reinterpret_cast<float4 *>(hr + 32 * ni + 8 * qd) = o;
                }
        }
    }
};

// ----------------------------------------------------------------------------------------------------------- GEMM
// C (M x N) = A (M x K) . W^T, W an nn.Linear weight (N, K) row-major; N % 128 == 0, K % 32 == 0.  One fma chain per
// output in increasing k.  n tiles fastest in blockIdx: the workgroups that share an A tile run together (L2).
template <class ALoad, class Epi>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(ALoad al, const float *__restrict__ W, int K, long long M, int ntn, Epi epi) {
    __shared__ __attribute__((aligned(16))) float As[GBM * GLDK], Ws[GBN * GLDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)(blockIdx.x / ntn) * GBM;
    const int n0 = (blockIdx.x % ntn) * GBN;
    // staging: item = (row, group of 8 k); a thread moves rows tid / 4 and 64 + tid / 4, group tid % 4 of both operands
    const int srow = tid >> 2, sg = tid & 3;
    long long ar0 = m0 + srow, ar1 = m0 + 64 + srow;
    if (ar0 > M - 1) ar0 = M - 1;
    if (ar1 > M - 1) ar1 = M - 1;
    float4 pa[4], pw[4];
    auto fetch = [&](int k0) {
        const float *a0 = al.at(ar0, k0 + 8 * sg), *a1 = al.at(ar1, k0 + 8 * sg);
        const float *w0 = W + (long long)(n0 + srow) * K + k0 + 8 * sg, *w1 = W + (long long)(n0 + 64 + srow) * K + k0 + 8 * sg;
        pa[0] = *reinterpret_cast<const float4 *>(a0);
        pa[1] = *reinterpret_cast<const float4 *>(a0 + 4);
        pa[2] = *reinterpret_cast<const float4 *>(a1);
        pa[3] = *reinterpret_cast<const float4 *>(a1 + 4);
        pw[0] = *reinterpret_cast<const float4 *>(w0);
        pw[1] = *reinterpret_cast<const float4 *>(w0 + 4);
        pw[2] = *reinterpret_cast<const float4 *>(w1);
        pw[3] = *reinterpret_cast<const float4 *>(w1 + 4);
    };
    auto stash = [&]() {
        float4 ev, od;
        kp8_split(pa[0], pa[1], ev, od);
        *reinterpret_cast<float4 *>(As + srow * GLDK + 8 * sg) = ev;
        *reinterpret_cast<float4 *>(As + srow * GLDK + 8 * sg + 4) = od;
        kp8_split(pa[2], pa[3], ev, od);
        *reinterpret_cast<float4 *>(As + (64 + srow) * GLDK + 8 * sg) = ev;
        *reinterpret_cast<float4 *>(As + (64 + srow) * GLDK + 8 * sg + 4) = od;
        kp8_split(pw[0], pw[1], ev, od);
        *reinterpret_cast<float4 *>(Ws + srow * GLDK + 8 * sg) = ev;
        *reinterpret_cast<float4 *>(Ws + srow * GLDK + 8 * sg + 4) = od;
        kp8_split(pw[2], pw[3], ev, od);
        *reinterpret_cast<float4 *>(Ws + (64 + srow) * GLDK + 8 * sg) = ev;
        *reinterpret_cast<float4 *>(Ws + (64 + srow) * GLDK + 8 * sg + 4) = od;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[0][0][e] = acc[0][1][e] = acc[1][0][e] = acc[1][1][e] = 0.0f;
    fetch(0);
    stash();
    __syncthreads();
    const float *Ar = As + (wm * 64 + r) * GLDK + 4 * h, *Wr = Ws + (wn * 64 + r) * GLDK + 4 * h;
    for (int k0 = 0; k0 < K; k0 += GBK) {
        // unconditional (the last iteration re-reads its own tile): behind a branch the loaded registers become phi values that
        // hipcc copies right after the loads - a wait for the whole memory latency in front of the MFMAs
        fetch(min(k0 + GBK, K - GBK));             // in flight during the 64 MFMAs below
        __builtin_amdgcn_sched_barrier(0);         // ... which the scheduler would otherwise sink to their use behind the MFMAs
#pragma unroll
        for (int g = 0; g < 4; g++) {
            // KP8 image: the float4 at 8 g + 4 h holds k = 8 g + 2 s + h, s = 0..3 -> MFMA step 4 g + s
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Ar + 8 * g), a1 = *reinterpret_cast<const f32x4 *>(Ar + 32 * GLDK + 8 * g);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Wr + 8 * g), b1 = *reinterpret_cast<const f32x4 *>(Wr + 32 * GLDK + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                acc[0][0] = mfma32(a0[s], b0[s], acc[0][0]);
                acc[0][1] = mfma32(a0[s], b1[s], acc[0][1]);
                acc[1][0] = mfma32(a1[s], b0[s], acc[1][0]);
                acc[1][1] = mfma32(a1[s], b1[s], acc[1][1]);
            }
        }
        __syncthreads();                           // every wave is through with this tile
        // unconditional as well (the last one is never read): a use only behind `if (more)` lets LLVM sink the loads into that
        // block, i.e. behind the MFMAs
        stash();
        __syncthreads();
    }
    epi(acc, m0 + wm * 64, n0 + wn * 64, r, h, M);
}

// The same GEMM on 32-row workgroups (four waves side by side, one 32 x 32 tile each): the launch shape for a FEW frames.  One frame's
// patch embedding is 7 x 3 workgroups of the form above, each walking 24 k tiles of 64 MFMAs per wave behind two barriers - 60 us
// on 21 CUs whatever the batch; here it is 25 x 3 workgroups of 16 MFMAs per wave and k tile.  Same k order per output: identical bits.
template <class ALoad, class Epi>
__global__ __launch_bounds__(256, 3) void gemm_f32_small_kernel(ALoad al, const float *__restrict__ W, int K, long long M, int ntn, Epi epi) {
    __shared__ __attribute__((aligned(16))) float As[32 * GLDK], Ws[GBN * GLDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const long long m0 = (long long)(blockIdx.x / ntn) * 32;
    const int n0 = (blockIdx.x % ntn) * GBN;
    const int srow = tid >> 2, sg = tid & 3;          // A: rows 0..31 (threads 0..127); W: rows srow and 64 + srow
    long long ar0 = m0 + (srow & 31);
    if (ar0 > M - 1) ar0 = M - 1;
    float4 pa[2], pw[4];
    auto fetch = [&](int k0) {
        const float *a0 = al.at(ar0, k0 + 8 * sg);
        const float *w0 = W + (long long)(n0 + srow) * K + k0 + 8 * sg, *w1 = W + (long long)(n0 + 64 + srow) * K + k0 + 8 * sg;
        pa[0] = *reinterpret_cast<const float4 *>(a0);
        pa[1] = *reinterpret_cast<const float4 *>(a0 + 4);
        pw[0] = *reinterpret_cast<const float4 *>(w0);
        pw[1] = *reinterpret_cast<const float4 *>(w0 + 4);
        pw[2] = *reinterpret_cast<const float4 *>(w1);
        pw[3] = *reinterpret_cast<const float4 *>(w1 + 4);
    };
    auto stash = [&]() {
        float4 ev, od;
        if (tid < 128) {
            kp8_split(pa[0], pa[1], ev, od);
            *reinterpret_cast<float4 *>(As + srow * GLDK + 8 * sg) = ev;
            *reinterpret_cast<float4 *>(As + srow * GLDK + 8 * sg + 4) = od;
        }
        kp8_split(pw[0], pw[1], ev, od);
        *reinterpret_cast<float4 *>(Ws + srow * GLDK + 8 * sg) = ev;
        *reinterpret_cast<float4 *>(Ws + srow * GLDK + 8 * sg + 4) = od;
        kp8_split(pw[2], pw[3], ev, od);
        *reinterpret_cast<float4 *>(Ws + (64 + srow) * GLDK + 8 * sg) = ev;
        *reinterpret_cast<float4 *>(Ws + (64 + srow) * GLDK + 8 * sg + 4) = od;
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    fetch(0);
    stash();
    __syncthreads();
    const float *Ar = As + r * GLDK + 4 * h, *Wr = Ws + (wave * 32 + r) * GLDK + 4 * h;
    for (int k0 = 0; k0 < K; k0 += GBK) {
        fetch(min(k0 + GBK, K - GBK));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Ar + 8 * g), b0 = *reinterpret_cast<const f32x4 *>(Wr + 8 * g);
#pragma unroll
            for (int sx = 0; sx < 4; sx++) acc = mfma32(a0[sx], b0[sx], acc);
        }
        __syncthreads();
        stash();
        __syncthreads();
    }
    epi.one_tile(acc, m0, n0 + wave * 32, r, h, M);
}

template <class ALoad, class Epi>
int launch_gemm(ALoad al, const float *W, int K, long long M, int N, Epi epi, hipStream_t st) {
    const int ntn = N / GBN;
    const long long blocks = (M + GBM - 1) / GBM * ntn;
    if (blocks < 192) {          // a few frames: fewer 128-row workgroups than a quarter of the chip's slots - 32-row workgroups
        hipLaunchKernelGGL((gemm_f32_small_kernel<ALoad, Epi>), dim3((unsigned)((M + 31) / 32 * ntn)), dim3(256), 0, st, al, W, K, M, ntn, epi);
    } else {
        hipLaunchKernelGGL((gemm_f32_kernel<ALoad, Epi>), dim3((unsigned)blocks), dim3(256), 0, st, al, W, K, M, ntn, epi);
    }
    sslam_count_launches(1);
    return hipGetLastError() == hipSuccess ? SSLAM_OK : SSLAM_E_LAUNCH;
}


// ------------------------------------------------------------------------------------------ GEMM, per-layer form
// The four GEMMs of a layer (96 % of the GEMM work): no LDS, no barrier, every wave on its own.
//   * W is PRE-PACKED in MFMA fragment order (sslam_vit_f32_pack_linear_host): one 1 KB buffer load per (column tile, group of 8 k)
//     with a SCALAR offset, straight from L2;
//   * with the k order 8 g + 4 h + s (below) lane (r, h)'s four MFMA steps of a group are 16 contiguous bytes of ITS row in the
//     natural layout: the A fragment is one buffer load per (row tile, group) too - 32 rows x 32 B per instruction; the two waves
//     that share the rows hit in L1, the lines are used up over the four groups of a k tile;
//   * the product is evaluated transposed (weights = MFMA A operand), so a lane owns a ROW of the output and every epilogue access
//     is 16 bytes wide;
//   * ring of four groups (16 registers each): three groups = 3 072 matrix cycles of latency cover for the A rows (MALL / HBM).
// Per group and wave: 16 MFMAs beside 4 loads and nothing else.
// Three forms of this kernel were built and clock-probed (tools/vit_f32_probe.py): LDS-staged with two barriers per k tile (the
// generic kernel above), LDS-staged A + streamed B with a third of its instructions, and this one.  All three ran at the same
// rate.  Where the time goes (probe: shader cycles against the 100 MHz real-time counter - the chip holds 2.31-2.36 GHz in these
// kernels, so the nominal 2.4 GHz peak is 3 % away, not more): the k loop is within 6 % of its matrix time (13.0 k cycles per k
// tile against 12.3 k); with the stores of the QKV epilogue switched off the kernel takes 425 us for 380 us of matrix time, with
// them 511 (482 after the RoPE tables were halved) - and it makes no difference whether those stores go to HBM or to a 4 MB
// region that stays in L2, whether the first round of workgroups starts staggered over a workgroup's life, or what issue priority
// the epilogue has.  Its cost is the NUMBER of vector-memory instructions whose 64 lanes touch 32 different 128-byte lines (a lane
// owns a row): ~50 cycles of the CU's address path each, 56 per wave in the first form.  A persistent form (768 workgroups
// looping over the tiles) was slower (QKV 528, up 670): the hardware's dispatch desynchronises the workgroups, a static loop
// keeps all epilogues of a CU in step.  The slot turnover (end of one workgroup to the start of the next) is ~7 us, 10 % of a
// K = 384 workgroup's life.
#ifdef SSLAM_CLOCK_PROBE
// probe builds only (tools/vit_f32_probe.py): wave 0 of the first 8192 workgroups stamps start / loop entry / loop exit / end
__device__ unsigned long long g_probe_gemm_f32[4 * 8192];
__device__ int g_probe_gemm_sel[2];        // (K, column tiles) of the GEMM to stamp
#endif
// RI = row tiles of 32 per wave: 2 (workgroup = 128 x 128, the throughput form) or 1 (64 x 128: twice the workgroups, half the
// work each - the form for a few frames, where a launch of the big form leaves most CUs empty and its time is one workgroup's
// latency; same k order per output, so a frame's tokens do not depend on the form).
// NI = column tiles of 32 per wave: 2, or 1 (with RI = 1: the four waves side by side, workgroup = 32 x 128 - the residual GEMMs of a
// one- or two-frame launch, where even the 64-row form is 39 workgroups of which each wave runs 1 536 dependent-pair MFMAs).
template <class Epi, int RI, int NI = 2>
__global__ __launch_bounds__(256, 3) void gemm_f32_rows_kernel(const float *__restrict__ A, int lda, const float *__restrict__ Wp, int K,
                                                                long long M, int ntn, Epi epi) {
    static_assert(NI == 2 || (RI == 1 && !Epi::TRANSPOSED), "the one-tile wave exists for the residual epilogue only");
    constexpr int BM_ = NI == 1 ? 32 : 64 * RI;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform BY CONSTRUCTION: the B loads' scalar offsets depend on it
    const int wm = NI == 1 ? 0 : wave >> 1, wn = NI == 1 ? wave : wave & 1;
    const int cw = wn * (32 * NI);                                    // the wave's first column inside the workgroup's 128
#ifdef SSLAM_CLOCK_PROBE
    const unsigned long long pr_t0 = clock64(), pr_w0 = wall_clock64();
#endif
    // XCD-aware order: workgroup b runs on XCD b % 8, and the ntn workgroups that share an A tile (one row tile, all column tiles)
    // share an L2: row tile = x + 8 (j / ntn), column tile = j % ntn for b = 8 j + x
    const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const long long m0 = (long long)(bx + 8 * (bj / ntn)) * BM_;
    if (m0 >= M) return;
    const int n0 = (bj % ntn) * GBN;
    const int groups = K / 8;
    // k order inside a group of 8: MFMA step s multiplies k = 8 g + s (lanes h = 0) and 8 g + 4 + s (h = 1) - any pairing of the
    // two lane halves is a valid k order as long as both operands use it, and with this one a lane's four steps are 16 CONTIGUOUS
    // bytes of its row in the natural layout (the KP8 order of the exact kernels, k = 8 g + 2 s + h, pins the fma chain to
    // increasing k; nothing here needs that).
    // A fragments: rows clamped to M - 1 (the epilogue masks them); byte offset of lane (r, h) in group g: row * lda * 4 + 32 g + 16 h
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, 0xfffffffe, 0x00020000);
    long long ra = m0 + wm * (32 * RI) + r, rb2 = ra + 32;
    if (ra > M - 1) ra = M - 1;
    if (rb2 > M - 1) rb2 = M - 1;
    const unsigned ao0 = (unsigned)(ra * lda * 4 + 16 * h), ao1 = (unsigned)(rb2 * lda * 4 + 16 * h);
    // B fragments: packed W, fragment (column tile nt, group g) at ((nt * K / 8 + g) * 1024) bytes, lane * 16 inside it
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, 0x7fffffff, 0x00020000);
    const int nt0 = (n0 + cw) / 32;
    const int wo0 = nt0 * groups * 1024, wo1 = wo0 + groups * 1024, loff = lane * 16;
    struct Frag {
        f32x4 a0, a1, b0, b1;
    };
    Frag q0, q1, q2, q3;
    auto load = [&](Frag &d, int G) {
        d.a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, ao0, G * 32, 0));
        if (RI == 2) d.a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, ao1, G * 32, 0));
        d.b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, wo0 + G * 1024, 0));
        if (NI == 2) d.b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, wo1 + G * 1024, 0));
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[0][0][e] = acc[0][1][e] = acc[1][0][e] = acc[1][1][e] = 0.0f;
    load(q0, 0);
    load(q1, 1);
    load(q2, 2);
#ifdef SSLAM_CLOCK_PROBE
    const unsigned long long pr_t1 = clock64();
#endif
    // TRANSPOSED product: weights are the MFMA's A operand, activation rows its B operand - acc[ni][ri] register e of lane (r, h) is
    // column 32 ni + crow(e, h), row 32 ri + r: a lane owns a row, four consecutive columns per four registers (16-byte epilogues)
    // (Epi::TRANSPOSED = false, the residual epilogues: rows on the registers, columns on the lanes)
    auto mma = [&](const Frag &c) {
#pragma unroll
        for (int st = 0; st < 4; st++) {
            if (Epi::TRANSPOSED) {
                acc[0][0] = mfma32(c.b0[st], c.a0[st], acc[0][0]);
                if (RI == 2) acc[0][1] = mfma32(c.b0[st], c.a1[st], acc[0][1]);
                acc[1][0] = mfma32(c.b1[st], c.a0[st], acc[1][0]);
                if (RI == 2) acc[1][1] = mfma32(c.b1[st], c.a1[st], acc[1][1]);
            } else {
                acc[0][0] = mfma32(c.a0[st], c.b0[st], acc[0][0]);
                if (NI == 2) acc[0][1] = mfma32(c.a0[st], c.b1[st], acc[0][1]);
                if (RI == 2) acc[1][0] = mfma32(c.a1[st], c.b0[st], acc[1][0]);
                if (RI == 2) acc[1][1] = mfma32(c.a1[st], c.b1[st], acc[1][1]);
            }
        }
    };
#pragma unroll 1
    for (int G = 0; G < groups; G += 4) {            // unconditional (clamped) loads: the compiler keeps an exact count in flight
        load(q3, min(G + 3, groups - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q0);
        __builtin_amdgcn_sched_barrier(0);
        load(q0, min(G + 4, groups - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q1);
        __builtin_amdgcn_sched_barrier(0);
        load(q1, min(G + 5, groups - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q2);
        __builtin_amdgcn_sched_barrier(0);
        load(q2, min(G + 6, groups - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q3);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef SSLAM_CLOCK_PROBE
    const unsigned long long pr_t2 = clock64();
#endif
    epi.template operator()<RI, NI>(acc, m0 + wm * (32 * RI), n0 + cw, r, h, M);
#ifdef SSLAM_CLOCK_PROBE
    if (tid == 0 && blockIdx.x < 8192 && K == g_probe_gemm_sel[0] && ntn == g_probe_gemm_sel[1]) {
        unsigned long long *o_ = g_probe_gemm_f32 + 4 * blockIdx.x;
        o_[0] = pr_t1 - pr_t0;
        o_[1] = pr_t2 - pr_t1;
        o_[2] = clock64() - pr_t2;
        o_[3] = wall_clock64() - pr_w0;      // 100 MHz ticks over the workgroup's life: cycles / ticks = the shader clock
    }
#endif
}

// FEW-FRAME FORM of a residual GEMM with a long K (the down projection, K = 1536; part of the form a batch of <= 8 frames runs,
// include/sslam_hip.h SSLAM_ATTN_KEY_SPLIT).  One frame is 25 row tiles of 32: in the one-tile-wave form above that is 75 workgroups
// whose every wave walks 768 dependent MFMAs (21 us of matrix time on 75 of 256 CUs), and at four frames 312 workgroups on 256
// CUs - the CUs that got two take twice as long.  Here a workgroup is ONE 32 x 32 output tile and its four waves are the four K
// QUARTERS: wave q sums k in [q K / 4, (q + 1) K / 4) as one fma chain, the quarters meet in LDS and wave 0 adds them in the
// fixed order ((q0 + q1) + q2) + q3 - deterministic and independent of the batch; against the single chain the result differs in
// the last bits, like the key-split attention, and the form is used under the same rule.  Four times the workgroups, a quarter
// of the chain each: the launch fills the chip at one frame (300 workgroups) and balances at four (1 188 quarter-length units on
// 768 slots instead of 312 full-length ones on 256 CUs).
template <class Epi>
__global__ __launch_bounds__(256, 3) void gemm_f32_rows_ks4_kernel(const float *__restrict__ A, int lda, const float *__restrict__ Wp, int K,
                                                                    long long M, int ntn32, Epi epi) {
    static_assert(!Epi::TRANSPOSED, "residual epilogue only");
    __shared__ float part[3][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int kq = __builtin_amdgcn_readfirstlane(tid >> 6);         // this wave's K quarter
    const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const long long m0 = (long long)(bx + 8 * (bj / ntn32)) * 32;
    if (m0 >= M) return;                                             // the whole workgroup
    const int n0 = (bj % ntn32) * 32;
    const int groups = K / 8, g0 = kq * (groups / 4), g1 = g0 + groups / 4;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, 0xfffffffe, 0x00020000);
    long long ra = m0 + r;
    if (ra > M - 1) ra = M - 1;
    const unsigned ao0 = (unsigned)(ra * lda * 4 + 16 * h);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wp), 0, 0x7fffffff, 0x00020000);
    const int wo0 = (n0 / 32) * groups * 1024, loff = lane * 16;
    struct Frag {
        f32x4 a, b;
    };
    Frag q0, q1, q2, q3;
    auto load = [&](Frag &d, int G) {
        d.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, ao0, G * 32, 0));
        d.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, wo0 + G * 1024, 0));
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[0][0][e] = acc[0][1][e] = acc[1][0][e] = acc[1][1][e] = 0.0f;
    load(q0, g0);
    load(q1, g0 + 1);
    load(q2, g0 + 2);
    auto mma = [&](const Frag &c) {
#pragma unroll
        for (int st = 0; st < 4; st++) acc[0][0] = mfma32(c.a[st], c.b[st], acc[0][0]);
    };
#pragma unroll 1
    for (int G = g0; G < g1; G += 4) {
        load(q3, min(G + 3, g1 - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q0);
        __builtin_amdgcn_sched_barrier(0);
        load(q0, min(G + 4, g1 - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q1);
        __builtin_amdgcn_sched_barrier(0);
        load(q1, min(G + 5, g1 - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q2);
        __builtin_amdgcn_sched_barrier(0);
        load(q2, min(G + 6, g1 - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(q3);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kq > 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) part[kq - 1][e * 64 + lane] = acc[0][0][e];
    }
    __syncthreads();
    if (kq > 0) return;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[0][0][e] = ((acc[0][0][e] + part[0][e * 64 + lane]) + part[1][e * 64 + lane]) + part[2][e * 64 + lane];
    epi.template operator()<1, 1>(acc, m0, n0, r, h, M);
}

template <class Epi>
int launch_gemm_rows_ks4(const float *A, int lda, const float *Wp, int K, long long M, int N, Epi epi, hipStream_t st) {
    if (K % 128 || N % 32) return SSLAM_E_UNSUPPORTED;             // four quarters of whole 4-group rounds
    const int ntn32 = N / 32;
    const long long blocks = ((M + 31) / 32 + 7) / 8 * 8 * ntn32;
    hipLaunchKernelGGL((gemm_f32_rows_ks4_kernel<Epi>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, Wp, K, M, ntn32, epi);
    sslam_count_launches(1);
    return hipGetLastError() == hipSuccess ? SSLAM_OK : SSLAM_E_LAUNCH;
}

#ifdef SSLAM_CLOCK_PROBE
extern "C" int sslam_probe_gemm_f32_select(int K, int ntn) {
    const int v[2] = {K, ntn};
    unsigned long long *p = nullptr;
    if (hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_probe_gemm_f32)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 4 * 8192);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_probe_gemm_sel), v, sizeof(v)) == hipSuccess ? 0 : -3;
}
extern "C" int sslam_probe_gemm_f32(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe_gemm_f32), sizeof(unsigned long long) * 4 * 8192) == hipSuccess ? 0 : -3;
}
#endif

#ifndef GEMM_SMALL_BELOW
#define GEMM_SMALL_BELOW 768
#endif
#ifndef GEMM_TINY_BELOW
#define GEMM_TINY_BELOW 192
#endif
template <class Epi>
int launch_gemm_rows(const float *A, int lda, const float *Wp, int K, long long M, int N, Epi epi, hipStream_t st) {
    const int ntn = N / GBN;
    const long long blocks = ((M + GBM - 1) / GBM + 7) / 8 * 8 * ntn;        // row tiles padded to a multiple of 8 (XCD-aware order)
    if (blocks < GEMM_SMALL_BELOW) {             // fewer workgroups than the chip has slots (256 CUs x 3): the 64-row form
        const long long blocks64 = ((M + 63) / 64 + 7) / 8 * 8 * ntn;
        if constexpr (!Epi::TRANSPOSED) {
            if (blocks64 < GEMM_TINY_BELOW) {    // still a fraction of the CUs: 32-row workgroups of four one-tile waves
                const long long blocks32 = ((M + 31) / 32 + 7) / 8 * 8 * ntn;
                hipLaunchKernelGGL((gemm_f32_rows_kernel<Epi, 1, 1>), dim3((unsigned)blocks32), dim3(256), 0, st, A, lda, Wp, K, M, ntn, epi);
                sslam_count_launches(1);
                return hipGetLastError() == hipSuccess ? SSLAM_OK : SSLAM_E_LAUNCH;
            }
        }
        hipLaunchKernelGGL((gemm_f32_rows_kernel<Epi, 1>), dim3((unsigned)blocks64), dim3(256), 0, st, A, lda, Wp, K, M, ntn, epi);
    } else {
        hipLaunchKernelGGL((gemm_f32_rows_kernel<Epi, 2>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, Wp, K, M, ntn, epi);
    }
    sslam_count_launches(1);
    return hipGetLastError() == hipSuccess ? SSLAM_OK : SSLAM_E_LAUNCH;
}

// ------------------------------------------------------------------------------------------------------ LayerNorm
// two-pass, one wave per row of 384 (torch.nn.LayerNorm: biased variance, eps inside the square root)
__global__ __launch_bounds__(256) void ln_rows_f32_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ b,
                                                           float eps, long long rows, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = x + row * FD;
    float v[6];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float2 t = *reinterpret_cast<const float2 *>(p + 128 * j + 2 * lane);
        v[2 * j] = t.x;
        v[2 * j + 1] = t.y;
        s += t.x + t.y;
    }
    const float mean = bfly64(s) / (float)FD;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) q += (v[j] - mean) * (v[j] - mean);
    const float rstd = 1.0f / sqrtf(bfly64(q) / (float)FD + eps);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int c = 128 * j + 2 * lane;
        float2 o;
        o.x = (v[2 * j] - mean) * rstd * g[c] + b[c];
        o.y = (v[2 * j + 1] - mean) * rstd * g[c + 1] + b[c + 1];
        *reinterpret_cast<float2 *>(out + row * FD + c) = o;
    }
}

__global__ __launch_bounds__(64) void prefix_rows_f32_kernel(const float *__restrict__ prefix, int T, float *__restrict__ x) {
    const int lane = threadIdx.x, i = blockIdx.x % FPREFIX;
    const long long row = (long long)(blockIdx.x / FPREFIX) * T + i;
#pragma unroll
    for (int j = 0; j < 3; j++)
        *reinterpret_cast<float2 *>(x + row * FD + 128 * j + 2 * lane) = *reinterpret_cast<const float2 *>(prefix + i * FD + 128 * j + 2 * lane);
}

// ------------------------------------------------------------------------------------------------------ attention
// q, k, v (n * 6, T, 64) fp32 -> y (n, T, 384), softmax(q k^T / 8) v.  A wave owns 32 queries: its query rows live in 32
// registers as the B operand of S^T = K Q^T (lane (r, h) holds q[r][8 g + 4 h + s] for step 4 g + s), so accumulator register e
// of lane (r, h) is the score of key crow(e, h) against query r - running maximum and row sum are lane-local (+ one xor-32), and
// register e, exponentiated, is directly the B operand (keys crow(e, 0), crow(e, 1)) of step e of O^T = V^T P^T.
//
// The tile loop is software-pipelined INSIDE the wave (as the matcher's, match.hip): iteration kt multiplies S^T of tile kt + 1
// and, behind every group of 4 of those MFMAs (256 cycles of matrix work), issues one slice of tile kt's softmax (and the LDS
// reads of its V^T fragments); then O^T += V^T P^T of tile kt with the LDS stores of the prefetched tiles behind its MFMAs.
// Run one after the other (multiply, softmax, multiply), two waves sharing a SIMD's matrix pipe fall into step - they multiply
// together and then exponentiate together, and the pipe idles through every softmax: 0.77 of its cycles busy, 782 us per layer
// and 83 frames.  sched_barrier pins the interleave (left alone, hipcc issues the MFMAs back to back).
// K runs one tile ahead of V in LDS: iteration kt reads K(kt + 1) and V(kt), stores K(kt + 2) and V(kt + 1).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifdef SSLAM_CLOCK_PROBE
// probe builds only (tools/vit_f32_probe.py): wave 0 of the first 8192 workgroups stamps shader cycles and the 100 MHz real-time counter
__device__ unsigned long long g_probe_attn_f32[4 * 8192];
#endif
constexpr int AW = 4, AKT = 32, ALDK = 68;      // waves per workgroup (4 x 32 queries; one wave per SIMD and workgroup), keys per tile
// KEY-SPLIT FORM (SPLIT = true; launches of a few frames - the reference's own callers run the backbone at B = 1 and B = 4,
// visualize_matches_sequence.py:72-74, train.py:300-302): one frame is 6 heads x 7 workgroups = 42 workgroups walking 25 key tiles one
// after the other - 62 us per layer on 42 of 256 CUs.  Here a workgroup takes ONE of ASPLIT contiguous key ranges (ASPLIT a
// compile-time constant, the ranges a function of T only) and leaves its un-normalised partial (O, m, l) per query in the workspace;
// attn_f32_merge_kernel combines the partials in the fixed order split 0 .. ASPLIT - 1.  The result is deterministic and a
// frame's tokens still do not depend on what else is in the launch - but they differ from the one-pass form's in the last bits
// (another summation order of the same softmax), so a launch of <= ASPLIT_MAX_FRAMES frames and a larger one agree to ~1e-6
// relative, not bit for bit (A1's bar is 1e-4 against the eager evaluation).
constexpr int ASPLIT = 5, ASPLIT_MAX_FRAMES = 8;
template <bool SPLIT>
__global__ __launch_bounds__(64 * AW, 3) void attn_f32_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                            float *__restrict__ y, int T, int nbh, int subs, float *__restrict__ part_o,
                                                            float *__restrict__ part_ml) {
    __shared__ __attribute__((aligned(16))) float Ks[2][AKT * ALDK], Vs[2][AKT * FHD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8; the `subs` workgroups of one (frame, head) stay on one XCD (K / V in its L2)
    // When the last workgroup of a (frame, head) has idle waves (T = 789: 25 query tiles = 6 x 4 + 1), those LIGHT workgroups are
    // dealt after all full ones: a light workgroup takes as long as a full one while it shares its SIMDs with full ones, but loads
    // the matrix pipe a quarter as much - interleaved (every 7th) they hold a seventh of the slots at a quarter of their capacity;
    // at the end of the launch they run among themselves, one or two waves per SIMD, in half the time.
    int bh, sub, split = 0;
    if constexpr (SPLIT) {
        // a few hundred workgroups at most: the hardware deals them round-robin over the XCDs and their CUs, one per CU
        // the workgroups of a (frame, head)'s last query tile group with idle waves (T = 789: 25 tiles = 6 x 4 + 1) are dealt after all
        // full ones, as in the one-pass form: four frames are 720 full workgroups - one round of the 768 slots - and 120 light ones
        const int n_qt_ = (T + 31) / 32, heavy = (n_qt_ % AW) ? subs - 1 : subs;
        int b = blockIdx.x;
        if (b < nbh * heavy * ASPLIT) {
            split = b % ASPLIT, sub = (b / ASPLIT) % heavy, bh = b / (ASPLIT * heavy);
        } else {
            b -= nbh * heavy * ASPLIT;
            split = b % ASPLIT, sub = heavy, bh = b / ASPLIT;
        }
    } else {
        const int n_qt_ = (T + 31) / 32, heavy = (n_qt_ % AW) ? subs - 1 : subs, n_chunks = (nbh + 7) / 8;
        int b = blockIdx.x, chunk;
        if (b < n_chunks * 8 * heavy) {
            chunk = b / (8 * heavy);
            const int within = b % (8 * heavy);
            bh = chunk * 8 + within % 8, sub = within / 8;
        } else {
            b -= n_chunks * 8 * heavy;
            chunk = b / 8;
            bh = chunk * 8 + b % 8, sub = heavy;
        }
    }
    if (bh >= nbh) return;
#ifdef SSLAM_CLOCK_PROBE
    const unsigned long long pr_c0 = clock64(), pr_w0 = wall_clock64();
#endif
    const int qt = sub * AW + wave;
    const int n_qt = (T + 31) / 32;
    const bool wave_on = qt < n_qt;                               // a wave without queries still helps staging
    const int qi = min(qt * 32 + r, T - 1);
    const float *qp = q + ((long long)bh * T + qi) * FHD, *kp = k + (long long)bh * T * FHD, *vp = v + (long long)bh * T * FHD;
    constexpr float QS = 0.125f * 1.4426950408889634f;           // 1 / sqrt(64) and log2(e): the softmax runs in the exp2 domain
    float qreg[32];
#pragma unroll
    for (int g = 0; g < 8; g++) {
        const float4 t = *reinterpret_cast<const float4 *>(qp + 8 * g + 4 * h);
        qreg[4 * g] = t.x * QS;
        qreg[4 * g + 1] = t.y * QS;
        qreg[4 * g + 2] = t.z * QS;
        qreg[4 * g + 3] = t.w * QS;
    }
    // staging, 256 threads: K tile = 32 keys x 16 float4 and V tile likewise, two float4 of each per thread (keys tid / 16 and
    // 16 + tid / 16); rows beyond T re-read row T - 1 (masked in the last tile)
    float4 pk0, pk1, pv0, pv1;
    const int skey = tid >> 4, sq4 = tid & 15;
    auto fetch_k = [&](int key0) {
        pk0 = *reinterpret_cast<const float4 *>(kp + (long long)min(key0 + skey, T - 1) * FHD + 4 * sq4);
        pk1 = *reinterpret_cast<const float4 *>(kp + (long long)min(key0 + 16 + skey, T - 1) * FHD + 4 * sq4);
    };
    auto fetch_v = [&](int key0) {
        pv0 = *reinterpret_cast<const float4 *>(vp + (long long)min(key0 + skey, T - 1) * FHD + 4 * sq4);
        pv1 = *reinterpret_cast<const float4 *>(vp + (long long)min(key0 + 16 + skey, T - 1) * FHD + 4 * sq4);
    };
    auto stash_k = [&](int buf) {
        *reinterpret_cast<float4 *>(&Ks[buf][skey * ALDK + 4 * sq4]) = pk0;
        *reinterpret_cast<float4 *>(&Ks[buf][(16 + skey) * ALDK + 4 * sq4]) = pk1;
    };
    auto stash_v = [&](int buf) {
        *reinterpret_cast<float4 *>(&Vs[buf][skey * FHD + 4 * sq4]) = pv0;
        *reinterpret_cast<float4 *>(&Vs[buf][(16 + skey) * FHD + 4 * sq4]) = pv1;
    };
    f32x16 o[2], sc;          // sc: the scores of the tile whose softmax is due
#pragma unroll
    for (int e = 0; e < 16; e++) o[0][e] = o[1][e] = sc[e] = 0.0f;
    float m = -1.0e30f, l = 0.0f;
    // key tiles [kt0, n_kt) of this workgroup: all of them, or the split's range (ceil(tiles / ASPLIT) each; a range past the end is empty)
    const int n_kt_all = (T + AKT - 1) / AKT, kt_per = SPLIT ? (n_kt_all + ASPLIT - 1) / ASPLIT : n_kt_all;
    const int kt0 = split * kt_per, n_kt = min(n_kt_all, kt0 + kt_per);
    if constexpr (SPLIT) {
        if (kt0 >= n_kt) {              // nothing to do: an empty partial (weight exp2(-1e30 - M) = 0 in the merge)
            const int qe = (sub * AW + wave) * 32 + r;
            if (qe < T && h == 0) {
                float *po = part_o + (((long long)split * nbh + bh) * T + qe) * FHD;
                for (int d = 0; d < FHD; d += 4) *reinterpret_cast<float4 *>(po + d) = make_float4(0.f, 0.f, 0.f, 0.f);
                float *pm = part_ml + (((long long)split * nbh + bh) * T + qe) * 2;
                pm[0] = -1.0e30f, pm[1] = 0.0f;
            }
            return;
        }
    }
    // prologue: K(kt0), V(kt0), K(kt0 + 1) in LDS; S^T of tile kt0 (buffers by tile parity, as in the loop)
    fetch_k(kt0 * AKT);
    fetch_v(kt0 * AKT);
    stash_k(kt0 & 1);
    stash_v(kt0 & 1);
    fetch_k(min(kt0 + 1, n_kt - 1) * AKT);
    stash_k((kt0 + 1) & 1);
    __syncthreads();
    if (wave_on) {
        const float *A = &Ks[kt0 & 1][r * ALDK + 4 * h];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(A + 8 * g);
#pragma unroll
            for (int st = 0; st < 4; st++) sc = mfma32(a[st], qreg[4 * g + st], sc);
        }
    }
    __syncthreads();          // the first iteration overwrites Ks[kt0 & 1]
    float vf0[16], vf1[16];
#define ATT_SLICE_MAX()                                                                                               \
    {                                                                                                                 \
        mx = m;                                                                                                       \
        _Pragma("unroll") for (int e = 0; e < 16; e++) mx = fmaxf(mx, sc[e]);                                         \
    }
#define ATT_SLICE_ALPHA()                                                                                             \
    {                                                                                                                 \
        const u32x2 w_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);    \
        mx = fmaxf(__uint_as_float(w_[0]), __uint_as_float(w_[1]));      /* xor-32 exchange without the LDS round trip */ \
        alpha = __builtin_amdgcn_exp2f(m - mx);                                                                       \
        m = mx;                                                                                                       \
    }
#define ATT_SLICE_EXP(e0_)                                                                                            \
    _Pragma("unroll") for (int e = (e0_); e < (e0_) + 4; e++) {                                                       \
        sc[e] = __builtin_amdgcn_exp2f(sc[e] - mx);                                                                   \
        ps += sc[e];                                                                                                  \
    }
#define ATT_VREADS(dst_, off_, e0_)                                                                                   \
    _Pragma("unroll") for (int e = (e0_); e < (e0_) + 8; e++) dst_[e] = V[crow(e, h) * FHD + (off_) + r];
    for (int kt = kt0; kt + 1 < n_kt; kt++) {
        const int kb = (kt + 1) & 1, vb = kt & 1;
        fetch_k(min(kt + 2, n_kt - 1) * AKT);          // unconditional and pinned here: see gemm_f32_kernel
        fetch_v((kt + 1) * AKT);
        __builtin_amdgcn_sched_barrier(0);
        if (wave_on) {
            const float *A = &Ks[kb][r * ALDK + 4 * h], *V = &Vs[vb][0];
            f32x4 a[8];
#pragma unroll
            for (int g = 0; g < 8; g++) a[g] = *reinterpret_cast<const f32x4 *>(A + 8 * g);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sn;
#pragma unroll
            for (int e = 0; e < 16; e++) sn[e] = 0.0f;
            float mx, alpha, ps = 0.0f;
#define ATT_MMA_S(g_)                                                                                                 \
    _Pragma("unroll") for (int st = 0; st < 4; st++) sn = mfma32(a[g_][st], qreg[4 * (g_) + st], sn);                 \
    __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(0)
            ATT_SLICE_MAX()
            __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(1)
            ATT_SLICE_ALPHA()
            __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(2)
            ATT_SLICE_EXP(0)
            __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(3)
            ATT_SLICE_EXP(4)
            __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(4)
            ATT_SLICE_EXP(8)
            ATT_VREADS(vf0, 0, 0)          // the V^T fragments take the registers the K fragments leave
            __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(5)
            ATT_SLICE_EXP(12)
            l = l * alpha + ps;
            ATT_VREADS(vf1, 32, 0)
            __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(6)
#pragma unroll
            for (int e = 0; e < 16; e++) o[0][e] *= alpha;
            ATT_VREADS(vf0, 0, 8)
            __builtin_amdgcn_sched_barrier(0);
            ATT_MMA_S(7)
#pragma unroll
            for (int e = 0; e < 16; e++) o[1][e] *= alpha;
            ATT_VREADS(vf1, 32, 8)
            __builtin_amdgcn_sched_barrier(0);
            // O^T += V^T P^T: step e multiplies keys (crow(e, 0), crow(e, 1)); A operand = V[key(h)][32 dt + r]; two independent chains
#pragma unroll
            for (int e = 0; e < 8; e++) {
                o[0] = mfma32(vf0[e], sc[e], o[0]);
                o[1] = mfma32(vf1[e], sc[e], o[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            stash_k(vb);          // K(kt + 2) over K(kt), last read in iteration kt - 1; V(kt + 1) over V(kt - 1) likewise
            stash_v(kb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 8; e < 16; e++) {
                o[0] = mfma32(vf0[e], sc[e], o[0]);
                o[1] = mfma32(vf1[e], sc[e], o[1]);
            }
            sc = sn;
        } else {
            stash_k(vb);
            stash_v(kb);
        }
        __syncthreads();
    }
#ifdef SSLAM_CLOCK_PROBE
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long *o_ = g_probe_attn_f32 + 4 * blockIdx.x;
        o_[0] = clock64() - pr_c0;
        o_[1] = wall_clock64() - pr_w0;
        o_[2] = pr_c0;
        o_[3] = pr_w0;
    }
#endif
    if (!wave_on) return;
    {   // last tile: the only one that can hold keys beyond T
        const int kt = n_kt - 1;
        const float *V = &Vs[kt & 1][0];
        ATT_VREADS(vf0, 0, 0)
        ATT_VREADS(vf0, 0, 8)
        ATT_VREADS(vf1, 32, 0)
        ATT_VREADS(vf1, 32, 8)
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (kt * AKT + crow(e, h) >= T) sc[e] = -INFINITY;
        float mx, alpha, ps = 0.0f;
        ATT_SLICE_MAX()
        ATT_SLICE_ALPHA()
        ATT_SLICE_EXP(0)
        ATT_SLICE_EXP(4)
        ATT_SLICE_EXP(8)
        ATT_SLICE_EXP(12)
        l = l * alpha + ps;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            o[0][e] *= alpha;
            o[1][e] *= alpha;
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            o[0] = mfma32(vf0[e], sc[e], o[0]);
            o[1] = mfma32(vf1[e], sc[e], o[1]);
        }
    }
#undef ATT_SLICE_MAX
#undef ATT_SLICE_ALPHA
#undef ATT_SLICE_EXP
#undef ATT_VREADS
#undef ATT_MMA_S
    if (qt * 32 + r >= T) return;
    l += __shfl_xor(l, 32);
    if constexpr (SPLIT) {
        // the partial as it stands: O un-normalised, the running maximum (exp2 domain) and the row sum
        float *po = part_o + (((long long)split * nbh + bh) * T + qt * 32 + r) * FHD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(po + 32 * dt + 8 * g) = make_float4(o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]);
        if (h == 0) {
            float *pm = part_ml + (((long long)split * nbh + bh) * T + qt * 32 + r) * 2;
            pm[0] = m, pm[1] = l;
        }
        return;
    }
    const float inv = 1.0f / l;
    const int frame = bh / FH, head = bh % FH;
    float *dst = y + ((long long)frame * T + qt * 32 + r) * FD + head * FHD + 4 * h;
    // register e of o[dt] is O[query r][32 dt + crow(e, h)]: four consecutive d per group of four registers
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            float4 t;
            t.x = o[dt][4 * g] * inv;
            t.y = o[dt][4 * g + 1] * inv;
            t.z = o[dt][4 * g + 2] * inv;
            t.w = o[dt][4 * g + 3] * inv;
            *reinterpret_cast<float4 *>(dst + 32 * dt + 8 * g) = t;
        }
}
// partials of the key-split form -> y (n, T, 384): M = max m_s, O = sum_s 2^(m_s - M) O_s, l = sum_s 2^(m_s - M) l_s in the order
// s = 0 .. ASPLIT - 1, y = O * (1 / l).  One thread per (frame-head, query, 4 consecutive d).
__global__ __launch_bounds__(256) void attn_f32_merge_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                            float *__restrict__ y, int T, int nbh) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)nbh * T * 16;
    if (i >= total) return;
    const int d4 = (int)(i & 15);
    const long long row = i >> 4;                         // bh * T + t
    const int bh = (int)(row / T), t = (int)(row % T);
    float ms[ASPLIT], ls[ASPLIT], M = -1.0e30f;
#pragma unroll
    for (int sp = 0; sp < ASPLIT; sp++) {
        const float2 ml = *reinterpret_cast<const float2 *>(part_ml + ((long long)sp * nbh * T + row) * 2);
        ms[sp] = ml.x, ls[sp] = ml.y;
        M = fmaxf(M, ml.x);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float l = 0.0f;
#pragma unroll
    for (int sp = 0; sp < ASPLIT; sp++) {
        const float wgt = __builtin_amdgcn_exp2f(ms[sp] - M);
        const float4 o = *reinterpret_cast<const float4 *>(part_o + ((long long)sp * nbh * T + row) * FHD + 4 * d4);
        acc.x = fmaf(wgt, o.x, acc.x), acc.y = fmaf(wgt, o.y, acc.y), acc.z = fmaf(wgt, o.z, acc.z), acc.w = fmaf(wgt, o.w, acc.w);
        l = fmaf(wgt, ls[sp], l);
    }
    const float inv = 1.0f / l;
    const int frame = bh / FH, head = bh % FH;
    *reinterpret_cast<float4 *>(y + ((long long)frame * T + t) * FD + head * FHD + 4 * d4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}
#ifdef SSLAM_CLOCK_PROBE
}  // namespace
extern "C" int sslam_probe_attn_f32(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe_attn_f32), sizeof(unsigned long long) * 4 * 8192) == hipSuccess ? 0 : -3;
}
namespace {
#endif

inline size_t ws_align(long long b) { return ((size_t)b + 255) & ~(size_t)255; }

}  // namespace

// nn.Linear weight (n_out, k_in) fp32 -> the fragment order gemm_f32_rows_kernel streams: out[nt][g][h][r][s] =
// w[32 nt + r][8 g + 4 h + s] (one 1 KB fragment per 32 columns x 8 k: lane (r, h) reads its four MFMA steps as one float4)
extern "C" int sslam_vit_f32_pack_linear_host(const float *w, int n_out, int k_in, float *out) {
    if (!w || !out || n_out <= 0 || k_in <= 0 || n_out % GBN || k_in % GBK) return SSLAM_E_INVALID;
    const int groups = k_in / 8;
    for (int n = 0; n < n_out; n++)
        for (int k = 0; k < k_in; k++) {
            const int nt = n / 32, r = n % 32, g = k / 8, h = (k % 8) / 4, s2 = k % 4;
            out[((((long long)nt * groups + g) * 2 + h) * 32 + r) * 4 + s2] = w[(long long)n * k_in + k];
        }
    return SSLAM_OK;
}

extern "C" long long sslam_vit_f32_workspace_bytes(int n_frames, int size) {
    if (n_frames <= 0 || size <= 0 || size % FPATCH) return SSLAM_E_INVALID;
    const long long G = size / FPATCH, T = G * G + FPREFIX, rows = (long long)n_frames * T;
    long long b = (long long)(ws_align(rows * FD * 4) * 2 + ws_align(rows * FD * 4 * 3) + ws_align(rows * FMLP * 4));   // x, y, qkv, hidden
    if (n_frames <= ASPLIT_MAX_FRAMES)           // the key-split attention's partials: ASPLIT x (O, (m, l)) per (frame, head, query)
        b += (long long)(ws_align(ASPLIT * rows * FD * 4) + ws_align(ASPLIT * rows * FH * 2 * 4));
    return b;
}

extern "C" int sslam_vit_forward_f32(const float *images_chw, int n_frames, int size, const sslam_vit_weights_f32_t *w, void *workspace,
                                     long long workspace_bytes, float *tokens_out, void *stream) {
    const bool few = n_frames <= ASPLIT_MAX_FRAMES && sslam_knob(KNOB_VIT_F32_NO_KEY_SPLIT, 0) == 0;
    return sslam_vit_forward_f32_form(images_chw, n_frames, size, w, workspace, workspace_bytes, tokens_out,
                                      few ? SSLAM_ATTN_KEY_SPLIT : SSLAM_ATTN_ONE_PASS, stream);
}

extern "C" int sslam_vit_forward_f32_form(const float *images_chw, int n_frames, int size, const sslam_vit_weights_f32_t *w, void *workspace,
                                          long long workspace_bytes, float *tokens_out, int attention_form, void *stream) {
    static_assert(ASPLIT_MAX_FRAMES == SSLAM_ATTN_KEY_SPLIT_MAX_FRAMES, "header and kernel disagree");
    if (!images_chw || !w || !workspace || !tokens_out || n_frames <= 0 || size <= 0 || size % FPATCH) return SSLAM_E_INVALID;
    if (attention_form != SSLAM_ATTN_ONE_PASS && attention_form != SSLAM_ATTN_KEY_SPLIT) return SSLAM_E_INVALID;
    if (attention_form == SSLAM_ATTN_KEY_SPLIT && n_frames > ASPLIT_MAX_FRAMES) return SSLAM_E_INVALID;
    if (workspace_bytes < sslam_vit_f32_workspace_bytes(n_frames, size)) return SSLAM_E_INVALID;
    if (((uintptr_t)images_chw | (uintptr_t)workspace | (uintptr_t)tokens_out) & 15) return SSLAM_E_INVALID;
    const int G = size / FPATCH, cells = G * G, T = cells + FPREFIX;
    const long long rows = (long long)n_frames * T, prow = (long long)n_frames * cells;
    // the per-layer GEMM addresses its A operand through one buffer descriptor with 32-bit byte offsets: the widest matrix (the MLP
    // hidden activation, 6 KB per token) must stay below 4 GiB - 886 frames at 448 x 448; callers cut larger batches into launch groups
    if (rows * FMLP * 4 > 0xfffffff0LL) return SSLAM_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    char *p = (char *)workspace;
    float *x = (float *)p;    p += ws_align(rows * FD * 4);
    float *y = (float *)p;    p += ws_align(rows * FD * 4);
    float *q = (float *)p, *k = q + rows * FD, *v = k + rows * FD;    p += ws_align(rows * FD * 4 * 3);
    float *hid = (float *)p;  p += ws_align(rows * FMLP * 4);
    const bool split = attention_form == SSLAM_ATTN_KEY_SPLIT;
    float *part_o = (float *)p, *part_ml = (float *)(p + ws_align(ASPLIT * rows * FD * 4));
    int rc = launch_gemm(AIm2Patch{images_chw, size, G}, w->patch_w, 3 * FPATCH * FPATCH, prow, FD, EpiPatch{w->patch_b, x, cells, T}, st);
    if (rc != SSLAM_OK) return rc;
    hipLaunchKernelGGL(prefix_rows_f32_kernel, dim3(n_frames * FPREFIX), dim3(64), 0, st, w->prefix, T, x);
    sslam_count_launches(1);
    const unsigned ln_grid = (unsigned)((rows + 3) / 4);
    const int n_qt = (T + 31) / 32, subs = (n_qt + AW - 1) / AW, nbh = n_frames * FH;
    for (int L = 0; L < FLAYERS; L++) {
        const sslam_vit_layer_f32_t &ly = w->layer[L];
        hipLaunchKernelGGL(ln_rows_f32_kernel, dim3(ln_grid), dim3(256), 0, st, x, ly.ln1_g, ly.ln1_b, 1e-5f, rows, y);
        sslam_count_launches(1);
        if ((rc = launch_gemm_rows(y, FD, ly.wqkv, FD, rows, 3 * FD, TQKV{ly.bqkv, w->rope_cos, w->rope_sin, q, k, v, T}, st)) != SSLAM_OK) return rc;
        if (split) {
            hipLaunchKernelGGL(attn_f32_kernel<true>, dim3((unsigned)(nbh * subs * ASPLIT)), dim3(64 * AW), 0, st, q, k, v, y, T, nbh, subs, part_o, part_ml);
            hipLaunchKernelGGL(attn_f32_merge_kernel, dim3((unsigned)(((long long)nbh * T * 16 + 255) / 256)), dim3(256), 0, st, part_o, part_ml, y, T, nbh);
            sslam_count_launches(2);
        } else {
            hipLaunchKernelGGL(attn_f32_kernel<false>, dim3((unsigned)((nbh + 7) / 8 * 8 * subs)), dim3(64 * AW), 0, st, q, k, v, y, T, nbh, subs,
                               (float *)nullptr, (float *)nullptr);
            sslam_count_launches(1);
        }
        if ((rc = launch_gemm_rows(y, FD, ly.wo, FD, rows, FD, EpiResidual{ly.bo, ly.ls1, x}, st)) != SSLAM_OK) return rc;
        hipLaunchKernelGGL(ln_rows_f32_kernel, dim3(ln_grid), dim3(256), 0, st, x, ly.ln2_g, ly.ln2_b, 1e-5f, rows, y);
        sslam_count_launches(1);
        if ((rc = launch_gemm_rows(y, FD, ly.wup, FD, rows, FMLP, TGelu{ly.bup, hid}, st)) != SSLAM_OK) return rc;
        if (split) {       // few-frame form: the long-K residual GEMM in four K quarters per workgroup
            if ((rc = launch_gemm_rows_ks4(hid, FMLP, ly.wdown, FMLP, rows, FD, EpiResidual{ly.bdown, ly.ls2, x}, st)) != SSLAM_OK) return rc;
        } else if ((rc = launch_gemm_rows(hid, FMLP, ly.wdown, FMLP, rows, FD, EpiResidual{ly.bdown, ly.ls2, x}, st)) != SSLAM_OK) return rc;
    }
    hipLaunchKernelGGL(ln_rows_f32_kernel, dim3(ln_grid), dim3(256), 0, st, x, w->norm_g, w->norm_b, 1e-5f, rows, tokens_out);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

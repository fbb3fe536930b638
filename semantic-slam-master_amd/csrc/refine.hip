// refine.hip - A6 + A7: bilinear feature gather and the descriptor MLP, fused per 64-row tile.
// Replaces DinoBackbone.extract_at_keypoints (reference semantic-slam/models/dino_backbone.py:114-152) and
// DescriptorRefiner.forward / ResidualBlock.forward (semantic-slam/models/descriptor_refiner.py:58-126).
//
// One workgroup (8 waves) owns 64 keypoint rows for the WHOLE chain: gather -> input_proj+ReLU -> n_blocks x
// {LN, fc1, ReLU, LN, fc2, +identity, ReLU} -> output_proj -> L2 normalise.  The 64x384 activation tile lives in LDS
// (KP8 order, 388-float rows) and is the MFMA A operand directly; the residual identity stays in registers in the
// MFMA C layout (each wave keeps the same 32x96 output sub-tile in every layer); only the pre-packed weights
// stream in (16-deep K stages, register double-buffered, L2-resident: 3.17 MB shared by all workgroups).
// All contractions are v_mfma_f32_32x32x2_f32 chains in increasing k from the bias: bit-identical to the oracle.
//
// Roofline: MFMA-bound, 1 572 864 FLOP per row (786.4 MFLOP per 500-keypoint frame) against 1.5 KB in / 0.5 KB out.
#include "common.h"

namespace {

constexpr int RM = 64;              // rows per workgroup
constexpr int HID = SSLAM_HID;      // 384
constexpr int LDH = HID + 4;        // activation row stride (floats)
constexpr int WBK = 16;             // K depth of one weight stage
constexpr int LDW = WBK + 4;        // weight stage row stride (floats): 80 B = 5 x 16 B
constexpr int NCHUNK = HID / WBK;   // 24 stages per layer
constexpr int H_FLOATS = RM * LDH;
constexpr int W_STAGE = HID * LDW;
constexpr int SMEM_FLOATS = H_FLOATS + 2 * W_STAGE;   // 160 768 B

struct RefArgs {
    const float *packed;
    sslam_refiner_layout_t lay;
};

// acc[t] (32 rows x 32 cols each, t-th N tile of this wave) = bias + H(64 x 384) . W^T, K walked in 24 stages
template <int NT>
__device__ __forceinline__ void gemm_lds(const float *H, float *Wst, const float *__restrict__ wp,
                                         const float *__restrict__ bias, int tid, f32x16 (&acc)[NT]) {
    constexpr int N = 128 * NT;
    constexpr int ITEMS = N * 4 / 512;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float bv = bias[wn * 32 * NT + t * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = bv;
    }
    float4 rw[ITEMS];
    const float4 *wp4 = reinterpret_cast<const float4 *>(wp);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) rw[i] = wp4[tid + 512 * i];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int q = tid + 512 * i;
        *reinterpret_cast<float4 *>(Wst + (q >> 2) * LDW + (q & 3) * 4) = rw[i];
    }
    __syncthreads();
    const float *A = H + (wm * 32 + r) * LDH + 4 * h;
    for (int s = 0; s < NCHUNK; s++) {
        if (s + 1 < NCHUNK) {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) rw[i] = wp4[(long long)(s + 1) * N * 4 + tid + 512 * i];
        }
        const float *B = Wst + (s & 1) * W_STAGE + (wn * 32 * NT + r) * LDW + 4 * h;
#pragma unroll
        for (int g = 0; g < WBK / 8; g++) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(A + s * WBK + 8 * g);
            f32x4 b[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) b[t] = *reinterpret_cast<const f32x4 *>(B + t * 32 * LDW + 8 * g);
#pragma unroll
            for (int st = 0; st < 4; st++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = mfma32(a[st], b[t][st], acc[t]);
        }
        if (s + 1 < NCHUNK) {
            float *W1 = Wst + ((s + 1) & 1) * W_STAGE;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const int q = tid + 512 * i;
                *reinterpret_cast<float4 *>(W1 + (q >> 2) * LDW + (q & 3) * 4) = rw[i];
            }
        }
        __syncthreads();
    }
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
#pragma unroll 2
    for (int rr = 0; rr < 8; rr++) {
        float *p = H + (wave * 8 + rr) * LDH + 8 * j;
        const float4 ev = *reinterpret_cast<const float4 *>(p), od = *reinterpret_cast<const float4 *>(p + 4);
        float x[8] = {ev.x, od.x, ev.y, od.y, ev.z, od.z, ev.w, od.w};
        float s = x[0];
#pragma unroll
        for (int i = 1; i < 8; i++) s = s + x[i];
        const float mean = bfly64(act ? s : 0.0f) / 384.0f;
        float s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float d = x[i] - mean;
            s2 = __builtin_fmaf(d, d, s2);
        }
        const float var = bfly64(act ? s2 : 0.0f) / 384.0f;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; i++) y[i] = __builtin_fmaf((x[i] - mean) * rstd, gm[i], bt[i]);
        if (act) {
            *reinterpret_cast<float4 *>(p) = make_float4(y[0], y[2], y[4], y[6]);
            *reinterpret_cast<float4 *>(p + 4) = make_float4(y[1], y[3], y[5], y[7]);
        }
    }
}

struct Taps {
    const float *src[4];
    float wt[4];
};

// grid_sample(bilinear, align_corners=True, zeros) tap set for one keypoint (oracle ora_gather)
__device__ __forceinline__ Taps make_taps(const float *feat_frame, int G, float x, float y) {
    const float gm1 = (float)(G - 1), half = gm1 / 2.0f;
    const float xn = 2.0f * x / gm1 - 1.0f, yn = 2.0f * y / gm1 - 1.0f;
    const float ix = (xn + 1.0f) * half, iy = (yn + 1.0f) * half;
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float w = ix - x0, e = 1.0f - w, n = iy - y0, s = 1.0f - n;
    Taps t;
    t.wt[0] = s * e; t.wt[1] = s * w; t.wt[2] = n * e; t.wt[3] = n * w;
    const int xi = (int)x0, yi = (int)y0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xs = xi + (k & 1), ys = yi + (k >> 1);
        t.src[k] = (xs < 0 || xs >= G || ys < 0 || ys >= G) ? nullptr : feat_frame + ((long long)ys * G + xs) * SSLAM_C;
    }
    return t;
}

__device__ __forceinline__ float4 ld4(const float *p, int off) {
    return p ? *reinterpret_cast<const float4 *>(p + off) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float4 blend4(const Taps &t, int off) {
    const float4 a = ld4(t.src[0], off), b = ld4(t.src[1], off), c = ld4(t.src[2], off), d = ld4(t.src[3], off);
    float4 o;
    o.x = ((a.x * t.wt[0] + b.x * t.wt[1]) + c.x * t.wt[2]) + d.x * t.wt[3];
    o.y = ((a.y * t.wt[0] + b.y * t.wt[1]) + c.y * t.wt[2]) + d.y * t.wt[3];
    o.z = ((a.z * t.wt[0] + b.z * t.wt[1]) + c.z * t.wt[2]) + d.z * t.wt[3];
    o.w = ((a.w * t.wt[0] + b.w * t.wt[1]) + c.w * t.wt[2]) + d.w * t.wt[3];
    return o;
}

__global__ __launch_bounds__(512) void gather_refine_kernel(const float *__restrict__ feat, int G,
                                                             const float *__restrict__ kp_xy, int K,
                                                             const float *__restrict__ x_in, long long rows,
                                                             RefArgs args, float *__restrict__ desc) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float *H = smem, *Wst = smem + H_FLOATS;
    const int tid = threadIdx.x;
    const long long R0 = (long long)blockIdx.x * RM;
    const float *pk = args.packed;
    const sslam_refiner_layout_t &L = args.lay;

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
    // (gemm_lds begins with a barrier after its own first weight stage is staged: H is visible by then)

    // ---- input_proj + ReLU (descriptor_refiner.py:76) -----------------------------------------------------------
    f32x16 X[3], acc[3];
    gemm_lds<3>(H, Wst, pk + L.in_w, pk + L.in_b, tid, acc);
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) X[t][e] = acc[t][e] > 0.0f ? acc[t][e] : 0.0f;
    store_tile<3>(H, tid, X);
    __syncthreads();

    // ---- residual blocks (descriptor_refiner.py:108-126) --------------------------------------------------------
    for (int b = 0; b < L.n_blocks; b++) {
        layernorm_rows(H, pk + L.blk[b][0], pk + L.blk[b][1], tid);
        gemm_lds<3>(H, Wst, pk + L.blk[b][2], pk + L.blk[b][3], tid, acc);
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[t][e] = acc[t][e] > 0.0f ? acc[t][e] : 0.0f;
        store_tile<3>(H, tid, acc);
        __syncthreads();
        layernorm_rows(H, pk + L.blk[b][4], pk + L.blk[b][5], tid);
        gemm_lds<3>(H, Wst, pk + L.blk[b][6], pk + L.blk[b][7], tid, acc);
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float v = acc[t][e] + X[t][e];
                X[t][e] = v > 0.0f ? v : 0.0f;
            }
        store_tile<3>(H, tid, X);
        __syncthreads();
    }

    // ---- output_proj + L2 normalise (:83-86; F.normalize eps 1e-12) --------------------------------------------
    f32x16 o[1];
    gemm_lds<1>(H, Wst, pk + L.out_w, pk + L.out_b, tid, o);
    {
        const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
        float *part = Wst;  // [64 rows][4 waves]; weight stages are idle (barrier at the end of gemm_lds)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float t = bfly32(o[0][e] * o[0][e]);
            if (r == 0) part[(wm * 32 + crow(e, h)) * 4 + wn] = t;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int row = wm * 32 + crow(e, h);
            const float4 t = *reinterpret_cast<const float4 *>(part + row * 4);
            const float ss = ((t.x + t.y) + t.z) + t.w;
            const float den = fmaxf(sqrtf(ss), 1e-12f);
            if (R0 + row < rows) desc[(R0 + row) * SSLAM_D + wn * 32 + r] = o[0][e] / den;
        }
    }
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
    hipLaunchKernelGGL(gather_refine_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, feat, G, kp_xy, K, x_in, rows,
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

// w (n_out, k_in) -> [chunk = k/16][n][16 floats in KP8 order]
extern "C" int sslam_pack_linear_host(const float *w, int n_out, int k_in, float *out) {
    if (!w || !out || n_out <= 0 || k_in <= 0 || (k_in % WBK)) return SSLAM_E_INVALID;
    for (int n = 0; n < n_out; n++)
        for (int k = 0; k < k_in; k++)
            out[((long long)(k / WBK) * n_out + n) * WBK + kp8(k % WBK)] = w[(long long)n * k_in + k];
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

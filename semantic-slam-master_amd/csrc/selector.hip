// selector.hip - A3: saliency CNN (conv3x3 384->hs + ReLU + conv1x1 hs->1 + sigmoid) as one fp32-MFMA implicit GEMM.
// Replaces KeypointSelector.forward (reference semantic-slam/models/keypoint_selector.py:45-67).
//
// GEMM view: M = n_frames*G*G cells, N = hs hidden channels, K = 9 taps * 384 channels (k = tap*384 + c).
// Workgroup = 8 waves, tile BM x hs with every wave owning a 64x64 sub-tile (2x2 MFMA 32x32 tiles):
//   hs = 256: waves 2(M) x 4(N), BM = 128;   hs = 128: waves 4(M) x 2(N), BM = 256.
// K is walked in 108 stages of 32 (tap-major), A rows are gathered from the NHWC feature map with the tap's
// spatial shift (zero rows outside the grid are multiplied through, exactly as the oracle does), B comes from the
// pre-packed weight image.  Both operands are staged through LDS in the KP8 order (common.h) with a +4 float row
// pad: one conflict-free ds_read_b128 then feeds four consecutive v_mfma_f32_32x32x2_f32 steps.
// Global->LDS staging is register double-buffered: the loads of stage s+1 are in flight while stage s computes.
//
// Roofline: MFMA-bound (fp32 matrix peak 157.3 TFLOP/s).  Algorithmic work: cells * hs * 3456 * 2 FLOP
// (+ hs*2 for the 1x1), i.e. 1 387.7 MFLOP per 28x28 frame at hs = 256.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDT = BK + 4;  // padded LDS row (floats): 144 B = 9 x 16 B -> conflict-free b128 fragment reads
constexpr int NSTAGE = 9 * (SSLAM_C / BK);

template <int WN>  // waves along N; hs = 64 * WN
__global__ __launch_bounds__(512) void selector_saliency_kernel(const float *__restrict__ feat, int n_rows, int G,
                                                                 const float *__restrict__ w1p,
                                                                 const float *__restrict__ b1,
                                                                 const float *__restrict__ w2,
                                                                 const float *__restrict__ b2, float *__restrict__ sal) {
    constexpr int WM = 8 / WN;
    constexpr int BM = 64 * WM;
    constexpr int HS = 64 * WN;
    constexpr int A_ITEMS = BM * 4 / 512;    // 8-float items per thread per stage
    constexpr int B_ITEMS = HS * 8 / 512;    // float4 items per thread per stage
    constexpr int STAGE_FLOATS = (BM + HS) * LDT;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int cells = G * G;
    const long long m0 = (long long)blockIdx.x * BM;

    // per-thread A rows (fixed for the whole K loop)
    int a_row[A_ITEMS], a_kq[A_ITEMS], a_y[A_ITEMS], a_x[A_ITEMS];
    const float *a_base[A_ITEMS];
    bool a_ok[A_ITEMS];
#pragma unroll
    for (int i = 0; i < A_ITEMS; i++) {
        const int it = tid + 512 * i;
        a_row[i] = it >> 2;
        a_kq[i] = it & 3;
        const long long m = m0 + a_row[i];
        a_ok[i] = m < n_rows;
        const long long mm = a_ok[i] ? m : 0;
        const int f = (int)(mm / cells), cell = (int)(mm % cells);
        a_y[i] = cell / G;
        a_x[i] = cell % G;
        a_base[i] = feat + (long long)f * cells * SSLAM_C + a_kq[i] * 8;
    }

    float4 ra_lo[A_ITEMS], ra_hi[A_ITEMS], rb[B_ITEMS];

#define LOAD_STAGE(S)                                                                                              \
    {                                                                                                              \
        const int s_ = (S);                                                                                        \
        const int tap = s_ / (SSLAM_C / BK), chunk = s_ % (SSLAM_C / BK);                                          \
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;                                                              \
        _Pragma("unroll") for (int i = 0; i < A_ITEMS; i++) {                                                      \
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;                                                          \
            const bool ok = a_ok[i] && yy >= 0 && yy < G && xx >= 0 && xx < G;                                     \
            const float4 *p = reinterpret_cast<const float4 *>(                                                    \
                a_base[i] + (ok ? ((long long)yy * G + xx) * SSLAM_C + chunk * BK : 0));                           \
            float4 v0 = p[0], v1 = p[1];                                                                           \
            ra_lo[i] = make_float4(ok ? v0.x : 0.f, ok ? v0.y : 0.f, ok ? v0.z : 0.f, ok ? v0.w : 0.f);            \
            ra_hi[i] = make_float4(ok ? v1.x : 0.f, ok ? v1.y : 0.f, ok ? v1.z : 0.f, ok ? v1.w : 0.f);            \
        }                                                                                                          \
        const float4 *wp = reinterpret_cast<const float4 *>(w1p + (long long)s_ * HS * BK);                        \
        _Pragma("unroll") for (int i = 0; i < B_ITEMS; i++) rb[i] = wp[tid + 512 * i];                             \
    }
#define STORE_STAGE(BUF)                                                                                           \
    {                                                                                                              \
        float *As_ = smem + (BUF) * STAGE_FLOATS;                                                                  \
        float *Bs_ = As_ + BM * LDT;                                                                               \
        _Pragma("unroll") for (int i = 0; i < A_ITEMS; i++) {                                                      \
            float4 ev, od;                                                                                         \
            kp8_split(ra_lo[i], ra_hi[i], ev, od);                                                                 \
            float *d = As_ + a_row[i] * LDT + a_kq[i] * 8;                                                         \
            *reinterpret_cast<float4 *>(d) = ev;                                                                   \
            *reinterpret_cast<float4 *>(d + 4) = od;                                                               \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < B_ITEMS; i++) {                                                      \
            const int q = tid + 512 * i;                                                                           \
            *reinterpret_cast<float4 *>(Bs_ + (q >> 3) * LDT + (q & 7) * 4) = rb[i];                               \
        }                                                                                                          \
    }

    // accumulators start from the conv bias: the fma chain is b1[n] + sum_k a_k * w_k in increasing k
    f32x16 acc[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
        const float bv = b1[wn * 64 + ni * 32 + r];
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ni][e] = bv;
    }

    LOAD_STAGE(0);
    STORE_STAGE(0);
    __syncthreads();
    for (int s = 0; s < NSTAGE; s++) {
        if (s + 1 < NSTAGE) LOAD_STAGE(s + 1);
        const float *As = smem + (s & 1) * STAGE_FLOATS + (wm * 64 + r) * LDT + 4 * h;
        const float *Bs = smem + (s & 1) * STAGE_FLOATS + BM * LDT + (wn * 64 + r) * LDT + 4 * h;
#pragma unroll
        for (int g = 0; g < BK / 8; g++) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(As + 8 * g);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(As + 32 * LDT + 8 * g);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bs + 8 * g);
            const f32x4 b1v = *reinterpret_cast<const f32x4 *>(Bs + 32 * LDT + 8 * g);
#pragma unroll
            for (int st = 0; st < 4; st++) {
                acc[0][0] = mfma32(a0[st], b0[st], acc[0][0]);
                acc[0][1] = mfma32(a0[st], b1v[st], acc[0][1]);
                acc[1][0] = mfma32(a1[st], b0[st], acc[1][0]);
                acc[1][1] = mfma32(a1[st], b1v[st], acc[1][1]);
            }
        }
        if (s + 1 < NSTAGE) STORE_STAGE((s + 1) & 1);
        __syncthreads();
    }

    // epilogue: ReLU, 1x1 conv (canonical tree: in-lane pair, 32-lane butterfly, slabs in order), sigmoid
    float *red = smem;  // [WN][BM]; every wave is past its last LDS read (barrier above)
    const float w2a = w2[wn * 64 + r], w2b = w2[wn * 64 + 32 + r];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float h0 = acc[mi][0][e] > 0.0f ? acc[mi][0][e] : 0.0f;
            const float h1 = acc[mi][1][e] > 0.0f ? acc[mi][1][e] : 0.0f;
            const float q = h0 * w2a + h1 * w2b;
            const float t = bfly32(q);
            if (r == 0) red[wn * BM + wm * 64 + mi * 32 + crow(e, h)] = t;
        }
    __syncthreads();
    for (int t = tid; t < BM; t += 512) {
        const long long m = m0 + t;
        if (m < n_rows) {
            float logit = b2[0];
#pragma unroll
            for (int s = 0; s < WN; s++) logit = logit + red[s * BM + t];
            sal[m] = sslam_sigmoid(logit);
        }
    }
}

#undef LOAD_STAGE
#undef STORE_STAGE
}  // namespace

extern "C" int sslam_selector_saliency(const float *feat, int n_frames, int G, const float *w1_packed, const float *b1,
                                       const float *w2, const float *b2, int hs, float *sal, void *stream) {
    if (!feat || !w1_packed || !b1 || !w2 || !b2 || !sal || n_frames <= 0 || G <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)feat | (uintptr_t)w1_packed) & 15) return SSLAM_E_INVALID;
    const long long rows = (long long)n_frames * G * G;
    if (rows > 0x7fffffffLL) return SSLAM_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hs == 256) {
        const unsigned grid = (unsigned)((rows + 127) / 128);
        hipLaunchKernelGGL(selector_saliency_kernel<4>, dim3(grid), dim3(512), 0, st, feat, (int)rows, G, w1_packed, b1,
                           w2, b2, sal);
    } else if (hs == 128) {
        const unsigned grid = (unsigned)((rows + 255) / 256);
        hipLaunchKernelGGL(selector_saliency_kernel<2>, dim3(grid), dim3(512), 0, st, feat, (int)rows, G, w1_packed, b1,
                           w2, b2, sal);
    } else {
        return SSLAM_E_UNSUPPORTED;
    }
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

// w (hs, 384, 3, 3) -> [stage = tap*12 + chunk][n][32 floats in KP8 order]
extern "C" int sslam_pack_conv3x3_host(const float *w, int hs, float *out) {
    if (!w || !out || hs <= 0) return SSLAM_E_INVALID;
    for (int tap = 0; tap < 9; tap++)
        for (int chunk = 0; chunk < SSLAM_C / BK; chunk++)
            for (int n = 0; n < hs; n++)
                for (int k = 0; k < BK; k++) {
                    const int c = chunk * BK + k;
                    out[(((long long)(tap * (SSLAM_C / BK) + chunk) * hs + n) * BK) + kp8(k)] =
                        w[((long long)n * SSLAM_C + c) * 9 + tap];
                }
    return SSLAM_OK;
}

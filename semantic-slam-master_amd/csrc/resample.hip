// resample.hip - A0 (image preprocessing) and A9 (per-keypoint intensity): Pillow's antialiased resampling in its
// 22-bit fixed-point integer arithmetic, bit-exact.
// Replaces transforms.Compose([Resize, ToTensor, Normalize]) applied at reference
// semantic-slam/visualize_matches_sequence.py:59-67,72 and the intensity lookup at :87-95.
// The arithmetic is third-party (Pillow's libImaging/Resample.c, unpinned in the reference's requirements.txt:5);
// it is restated from the published algorithm and pinned by golden vectors made with Pillow 12.2.0.
//
// Byte / integer work, HBM-bound: 921 600 B in and 2 408 448 B out per 640x480 -> 448x448 frame (A0); A9 touches
// at most ksize_v * ksize_h source pixels per keypoint and never materialises the resized image.
#include <math.h>

#include <algorithm>

#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int PREC = 22;   // Pillow PRECISION_BITS = 32 - 8 - 2
constexpr int TX = 64, TY = 16, MAXR = 64;

__device__ __forceinline__ int clip8(int v) {
    v >>= PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t *__restrict__ img, int h, int w, int size,
                                                          const int *__restrict__ bh, const int *__restrict__ ch, int ksh,
                                                          const int *__restrict__ bv, const int *__restrict__ cv, int ksv,
                                                          float *__restrict__ out) {
    __shared__ uint8_t tmp[MAXR][TX][3];
    __shared__ float lut[3][256];
    const int tid = threadIdx.x;
    const long long f = blockIdx.z;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int ty1 = min(y0 + TY, size) - 1;
    const int rmin = bv[2 * y0], rmax = bv[2 * ty1] + bv[2 * ty1 + 1];   // input rows [rmin, rmax)
    const int nrows = rmax - rmin;
    const uint8_t *src = img + f * h * w * 3;
    {   // ToTensor (/255) and Normalize ((x - mean) / std) as a 256-entry table per channel, in fp32
        const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
        for (int c = 0; c < 3; c++) lut[c][tid] = ((float)tid / 255.0f - mean[c]) / sd[c];
    }
    // horizontal pass into LDS (uint8, as Pillow's intermediate image)
    for (int it = tid; it < nrows * TX; it += 256) {
        const int rr = it / TX, xx = it % TX;
        const int ox = x0 + xx;
        if (ox < size) {
            const int xmin = bh[2 * ox], xn = bh[2 * ox + 1];
            const int *k = ch + (long long)ox * ksh;
            const uint8_t *p = src + ((long long)(rmin + rr) * w + xmin) * 3;
            int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
            for (int t = 0; t < xn; t++) {
                const int kk = k[t];
                s0 += p[3 * t] * kk;
                s1 += p[3 * t + 1] * kk;
                s2 += p[3 * t + 2] * kk;
            }
            tmp[rr][xx][0] = (uint8_t)clip8(s0);
            tmp[rr][xx][1] = (uint8_t)clip8(s1);
            tmp[rr][xx][2] = (uint8_t)clip8(s2);
        }
    }
    __syncthreads();
    // vertical pass + normalisation, planar fp32 output
    const long long plane = (long long)size * size;
    for (int it = tid; it < TY * TX * 3; it += 256) {
        const int c = it / (TY * TX), rem = it % (TY * TX), yy = rem / TX, xx = rem % TX;
        const int oy = y0 + yy, ox = x0 + xx;
        if (oy < size && ox < size) {
            const int ymin = bv[2 * oy] - rmin, yn = bv[2 * oy + 1];
            const int *k = cv + (long long)oy * ksv;
            int s = 1 << (PREC - 1);
            for (int t = 0; t < yn; t++) s += tmp[ymin + t][xx][c] * k[t];
            out[(f * 3 + c) * plane + (long long)oy * size + ox] = lut[c][clip8(s)];
        }
    }
}

// Fast path (horizontal taps <= 7, i.e. bilinear down to 1/3 or bicubic down to 2/3 of the input width):
//  - a thread owns ONE output column for the whole tile, so its bounds / coefficients live in registers;
//  - the <= 21 source bytes of a pixel are fetched as one or two range-checked buffer loads (16 + 4 / 8 bytes) at the
//    dword below the first byte and re-aligned with v_alignbyte (byte loads made the first version of this kernel
//    TA-instruction bound at 1.36 TB/s); 32-bit offsets into ONE descriptor over the launch's frames: no 64-bit
//    address arithmetic, no end-of-buffer branches (the descriptor returns 0 beyond the last dword);
//  - every product is an 8-bit sample times a 22-bit coefficient: v_mad_i32_i24 (full rate) - the plain 32-bit
//    multiply hipcc picks for `int * int` is quarter rate and was half of this kernel's VALU time;
//  - the uint8 intermediate row is kept packed (R | G<<8 | B<<16) so the vertical pass reads one dword per tap;
//  - the vertical coefficients of an output row are wave-uniform (a wave owns whole rows): scalar loads.
//  - TAPS (3, 5 or 7) is the compile-time tap count: 640 -> 448 bilinear needs 5, so two of seven tap slots would be zeros.
//  - PATCH = true: the output is not the planar fp32 image but the ViT's patch-embedding operand - bf16 rows of 768 per
//    16 x 16 patch, k = c * 256 + ky * 16 + kx (what im2patch_kernel of vit.hip made from the fp32 image): the same
//    normalised value rounded to bf16 once.  Half the bytes written, and the fp32 image + the im2patch pass disappear.
template <int TAPS, int TYT, int MAXRT, bool PATCH = false>
__global__ __launch_bounds__(256) void preprocess_fast_kernel(const uint8_t *__restrict__ img, unsigned total_bytes, int h, int w,
                                                               int size, const int *__restrict__ bh, const int *__restrict__ ch,
                                                               int ksh, const int *__restrict__ bv, const int *__restrict__ cv,
                                                               int ksv, float *__restrict__ out, int gx, int tiles, int n_frames) {
    __shared__ unsigned tmp[MAXRT][TX];
    __shared__ float lut[3][256];
    const int tid = threadIdx.x;
    // XCD-aware order (1-D grid): workgroup b runs on XCD b % 8 - give every XCD WHOLE frames (frame = 8 (j / tiles) + b % 8,
    // tile = j % tiles, j = b / 8), so that the source rows and columns two neighbouring tiles both filter come from that XCD's
    // L2; with the frame's tiles dealt round-robin over the eight XCDs every L2 fetched its own copy of the overlap (852 MB
    // fetched for 565 MB of pixels at 613 frames)
    const int bq = blockIdx.x >> 3;
    const unsigned f = (unsigned)((bq / tiles) * 8 + (blockIdx.x & 7));
    if ((int)f >= n_frames) return;
    const int tl = bq % tiles;
    const int x0 = (tl % gx) * TX, y0 = (tl / gx) * TYT;
    const int ty1 = min(y0 + TYT, size) - 1;
    const int rmin = bv[2 * y0], rmax = bv[2 * ty1] + bv[2 * ty1 + 1];   // input rows [rmin, rmax)
    const int nrows = rmax - rmin;
    // a dword that holds one valid byte lies inside the allocation's last page: the descriptor covers whole dwords
    const __amdgpu_buffer_rsrc_t irs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img), 0, (int)((total_bytes + 3u) & ~3u), 0x00020000);
    {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
        for (int c = 0; c < 3; c++) lut[c][tid] = ((float)tid / 255.0f - mean[c]) / sd[c];
    }
    const int xx = tid & (TX - 1), q = __builtin_amdgcn_readfirstlane(tid >> 6);   // q = 0..3: the wave
    const int ox = min(x0 + xx, size - 1);
    {
        const int xmin = bh[2 * ox], xn = bh[2 * ox + 1];
        constexpr int NW = (3 * TAPS + 3 + 3) / 4;                          // dwords covering 3*TAPS bytes at any alignment
        int kh[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; t++) kh[t] = t < xn ? ch[ox * ksh + t] : 0;
        const unsigned col0 = f * (unsigned)(h * w * 3) + (unsigned)(rmin * w + xmin) * 3u;
        const unsigned rstride = (unsigned)w * 3u;
        // RB rows are fetched together (the pass is load-latency bound otherwise), then filtered one after the other
        constexpr int RB = 3;
        for (int rr0 = q; rr0 < nrows; rr0 += 4 * RB) {
            unsigned wds[RB][8];
            int shv[RB];
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const int rr = min(rr0 + 4 * i, nrows - 1);
                const unsigned b0 = col0 + (unsigned)rr * rstride;
                const int a0 = (int)(b0 & ~3u);
                shv[i] = (int)(b0 & 3u);
                const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(irs, a0, 0, 0);
                wds[i][0] = lo[0], wds[i][1] = lo[1], wds[i][2] = lo[2], wds[i][3] = lo[3];
                if constexpr (NW == 5) wds[i][4] = __builtin_amdgcn_raw_buffer_load_b32(irs, a0, 16, 0);
                if constexpr (NW == 6) {
                    const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(irs, a0, 16, 0);
                    wds[i][4] = hi[0], wds[i][5] = hi[1];
                }
            }
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const int rr = rr0 + 4 * i;
                if (rr >= nrows) break;
                const int sh = shv[i];
                unsigned al[NW];                                             // bytes b0.. in order
#pragma unroll
                for (int j = 0; j < NW - 1; j++) al[j] = __builtin_amdgcn_alignbyte(wds[i][j + 1], wds[i][j], sh);
                al[NW - 1] = wds[i][NW - 1] >> (8 * sh);
                int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
#pragma unroll
                for (int t = 0; t < TAPS; t++) {
                    const int o = 3 * t;
                    const int r8 = (al[o >> 2] >> (8 * (o & 3))) & 255;
                    const int g8 = (al[(o + 1) >> 2] >> (8 * ((o + 1) & 3))) & 255;
                    const int b8 = (al[(o + 2) >> 2] >> (8 * ((o + 2) & 3))) & 255;
                    s0 += __mul24(r8, kh[t]);
                    s1 += __mul24(g8, kh[t]);
                    s2 += __mul24(b8, kh[t]);
                }
                tmp[rr][xx] = (unsigned)clip8(s0) | ((unsigned)clip8(s1) << 8) | ((unsigned)clip8(s2) << 16);
            }
        }
    }
    __syncthreads();
    const long long plane = (long long)size * size;
    if (x0 + xx < size) {
#pragma unroll
        for (int j = 0; j < TYT / 4; j++) {
            const int oy = y0 + q * (TYT / 4) + j;                           // wave-uniform
            if (oy < size) {
                const int ymin = bv[2 * oy] - rmin, yn = bv[2 * oy + 1];
                const int *k = cv + oy * ksv;
                int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
                for (int t = 0; t < yn; t++) {
                    const unsigned p = tmp[ymin + t][xx];
                    const int kk = k[t];
                    s0 += __mul24((int)(p & 255), kk);
                    s1 += __mul24((int)((p >> 8) & 255), kk);
                    s2 += __mul24((int)((p >> 16) & 255), kk);
                }
                if constexpr (PATCH) {
                    const int G = size >> 4, ox_ = x0 + xx;
                    __bf16 *o = reinterpret_cast<__bf16 *>(out) +
                                (((long long)f * G + (oy >> 4)) * G + (ox_ >> 4)) * 768 + (oy & 15) * 16 + (ox_ & 15);
                    o[0] = (__bf16)lut[0][clip8(s0)];
                    o[256] = (__bf16)lut[1][clip8(s1)];
                    o[512] = (__bf16)lut[2][clip8(s2)];
                } else {
                    float *o = out + (long long)f * 3 * plane + (long long)oy * size + (x0 + xx);
                    o[0] = lut[0][clip8(s0)];
                    o[plane] = lut[1][clip8(s1)];
                    o[2 * plane] = lut[2][clip8(s2)];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void intensity_kernel(const uint8_t *__restrict__ img, int h, int w, int size,
                                                         const int *__restrict__ bh, const int *__restrict__ ch, int ksh,
                                                         const int *__restrict__ bv, const int *__restrict__ cv, int ksv,
                                                         const float *__restrict__ kp_pixel, int K, long long total,
                                                         float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long f = i / K;
    const uint8_t *src = img + f * h * w * 3;
    // numpy round() is half-to-even; clip to the resized image (visualize_matches_sequence.py:93-94)
    int X = (int)__builtin_rintf(kp_pixel[2 * i]), Y = (int)__builtin_rintf(kp_pixel[2 * i + 1]);
    X = min(max(X, 0), size - 1);
    Y = min(max(Y, 0), size - 1);
    const int xmin = bh[2 * X], xn = bh[2 * X + 1], ymin = bv[2 * Y], yn = bv[2 * Y + 1];
    const int *kh = ch + (long long)X * ksh, *kv = cv + (long long)Y * ksv;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    for (int ry = 0; ry < yn; ry++) {
        const uint8_t *p = src + ((long long)(ymin + ry) * w + xmin) * 3;
        int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < xn; t++) {
            const int kk = kh[t];
            s0 += p[3 * t] * kk;
            s1 += p[3 * t + 1] * kk;
            s2 += p[3 * t + 2] * kk;
        }
        const int kk = kv[ry];
        a0 += clip8(s0) * kk;
        a1 += clip8(s1) * kk;
        a2 += clip8(s2) * kk;
    }
    const int R = clip8(a0), Gc = clip8(a1), B = clip8(a2);
    const int L = (R * 19595 + Gc * 38470 + B * 7471 + 0x8000) >> 16;   // Pillow "L" (ITU-R 601-2, 16-bit fixed)
    out[i] = (float)L / 255.0f;
}

double filt_bilinear(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}
double filt_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace

// Pillow precompute_coeffs + normalize_coeffs_8bpc for one axis (host, double precision, same operation order)
extern "C" int sslam_resample_table_host(int in_size, int out_size, int filter, int32_t *bounds, int32_t *coefs,
                                         int coefs_capacity) {
    if (in_size <= 0 || out_size <= 0 || !bounds || !coefs || filter < 0 || filter > 1) return SSLAM_E_INVALID;
    double (*fn)(double) = filter ? filt_bicubic : filt_bilinear;
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter ? 2.0 : 1.0) * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (ksize > SSLAM_MAX_TAPS) return SSLAM_E_UNSUPPORTED;
    if ((long long)ksize * out_size > coefs_capacity) return SSLAM_E_INVALID;
    double k[SSLAM_MAX_TAPS];
    for (int xx = 0; xx < out_size; xx++) {
        const double center = 0.0 + (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; x++) {
            const double wgt = fn((x + xmin - center + 0.5) * ss);
            k[x] = wgt;
            ww += wgt;
        }
        for (x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        for (x = 0; x < ksize; x++)
            coefs[(long long)xx * ksize + x] =
                k[x] < 0 ? (int32_t)(-0.5 + k[x] * (1 << PREC)) : (int32_t)(0.5 + k[x] * (1 << PREC));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

template <bool PATCH>
static int preprocess_launch(const uint8_t *img, int n, int h, int w, int size, const int32_t *bounds_h, const int32_t *coefs_h,
                             int ksize_h, const int32_t *bounds_v, const int32_t *coefs_v, int ksize_v, void *out_v, void *stream) {
    if (!img || !bounds_h || !coefs_h || !bounds_v || !coefs_v || !out_v || n <= 0 || h <= 0 || w <= 0 || size <= 0)
        return SSLAM_E_INVALID;
    if (ksize_h <= 0 || ksize_v <= 0 || ksize_h > SSLAM_MAX_TAPS || ksize_v > SSLAM_MAX_TAPS) return SSLAM_E_INVALID;
    if (PATCH && (size % 16 || ((uintptr_t)out_v & 1))) return SSLAM_E_INVALID;
    // input rows one output tile can need: TY output rows span TY*scale input rows plus the filter support
    if ((long long)(TY * (long long)h + size - 1) / size + ksize_v + 2 > MAXR) return SSLAM_E_UNSUPPORTED;
    // the fast kernel prefers 32-row tiles (less vertical-halo re-filtering) when their input rows fit its LDS buffer
    const bool tall = (32LL * h + size - 1) / size + ksize_v + 2 <= 64;
    // launch groups: grid.z <= 65535 and, for the fast kernel, all byte offsets of a group inside one 32-bit descriptor
    const long long fbytes = (long long)h * w * 3;
    if (fbytes > 0x7fffffffLL) return SSLAM_E_UNSUPPORTED;
    int per_group = (int)std::min<long long>(65535, std::max<long long>(1, 0xfffffff0LL / fbytes));
    // the fast kernel realigns its loads against a dword-aligned group base: with odd frame sizes (h * w * 3 not a multiple
    // of 4) keep every group's base aligned by cutting groups at multiples of 4 frames
    if ((fbytes & 3) && per_group >= 4) per_group &= ~3;
    for (int n0 = 0; n0 < n; n0 += per_group) {
        const int ng = std::min(per_group, n - n0);
        const uint8_t *gi = img + (long long)n0 * fbytes;
        const bool aligned = !((uintptr_t)gi & 3);              // per GROUP: the generic kernel takes an unaligned one
        // group output: planar fp32 (3 * size^2 floats per frame) or bf16 patch rows (3 * size^2 bf16 per frame)
        float *go = PATCH ? reinterpret_cast<float *>(reinterpret_cast<__bf16 *>(out_v) + (long long)n0 * 3 * size * size)
                          : reinterpret_cast<float *>(out_v) + (long long)n0 * 3 * size * size;
        const unsigned gbytes = (unsigned)(ng * fbytes);
        const dim3 grid((size + TX - 1) / TX, (size + TY - 1) / TY, ng);
        const int gx = (size + TX - 1) / TX, tiles_t = gx * ((size + 31) / 32), tiles_s = gx * ((size + TY - 1) / TY);
        const long long nb_t = (long long)((ng + 7) / 8) * 8 * tiles_t, nb_s = (long long)((ng + 7) / 8) * 8 * tiles_s;
        if (nb_t > 0x7fffffffLL || nb_s > 0x7fffffffLL) return SSLAM_E_UNSUPPORTED;
#define FAST(T)                                                                                                        \
    {                                                                                                                  \
        if (tall)                                                                                                      \
            hipLaunchKernelGGL((preprocess_fast_kernel<T, 32, 64, PATCH>), dim3((unsigned)nb_t), dim3(256), 0, (hipStream_t)stream, gi, gbytes, h, w, \
                               size, bounds_h, coefs_h, ksize_h, bounds_v, coefs_v, ksize_v, go, gx, tiles_t, ng);     \
        else                                                                                                           \
            hipLaunchKernelGGL((preprocess_fast_kernel<T, TY, MAXR, PATCH>), dim3((unsigned)nb_s), dim3(256), 0, (hipStream_t)stream, gi, gbytes, h, w, \
                               size, bounds_h, coefs_h, ksize_h, bounds_v, coefs_v, ksize_v, go, gx, tiles_s, ng);     \
    }
        if (ksize_h <= 3 && aligned)
            FAST(3)
        else if (ksize_h <= 5 && aligned)
            FAST(5)
        else if (ksize_h <= 7 && aligned)
            FAST(7)
        else if (PATCH)
            return SSLAM_E_UNSUPPORTED;          // patch rows come from the fast kernel only (callers take the fp32 entry then)
        else
            hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, (hipStream_t)stream, gi, h, w, size, bounds_h, coefs_h,
                               ksize_h, bounds_v, coefs_v, ksize_v, go);
#undef FAST
    }
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

extern "C" int sslam_preprocess_u8(const uint8_t *img, int n, int h, int w, int size, const int32_t *bounds_h,
                                   const int32_t *coefs_h, int ksize_h, const int32_t *bounds_v, const int32_t *coefs_v,
                                   int ksize_v, float *out_chw, void *stream) {
    return preprocess_launch<false>(img, n, h, w, size, bounds_h, coefs_h, ksize_h, bounds_v, coefs_v, ksize_v, out_chw, stream);
}

extern "C" int sslam_preprocess_u8_patches(const uint8_t *img, int n, int h, int w, int size, const int32_t *bounds_h,
                                           const int32_t *coefs_h, int ksize_h, const int32_t *bounds_v, const int32_t *coefs_v,
                                           int ksize_v, void *out_patches_bf16, void *stream) {
    return preprocess_launch<true>(img, n, h, w, size, bounds_h, coefs_h, ksize_h, bounds_v, coefs_v, ksize_v, out_patches_bf16, stream);
}

extern "C" int sslam_keypoint_intensity(const uint8_t *img, int n, int h, int w, int size, const int32_t *bounds_h,
                                        const int32_t *coefs_h, int ksize_h, const int32_t *bounds_v, const int32_t *coefs_v,
                                        int ksize_v, const float *kp_pixel, int K, float *out, void *stream) {
    if (!img || !bounds_h || !coefs_h || !bounds_v || !coefs_v || !kp_pixel || !out || n <= 0 || h <= 0 || w <= 0 ||
        size <= 0 || K <= 0)
        return SSLAM_E_INVALID;
    const long long total = (long long)n * K;
    hipLaunchKernelGGL(intensity_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, h, w,
                       size, bounds_h, coefs_h, ksize_h, bounds_v, coefs_v, ksize_v, kp_pixel, K, total, out);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

// gather_taps.h - bilinear tap set of grid_sample(align_corners=True, padding zeros) for one keypoint: shared by the
// exact (refine.hip) and bf16 (refine_bf16.hip) descriptor kernels.  Reference: dino_backbone.py:114-152.
#pragma once
#include "common.h"

namespace {

// one all-zero feature row: a tap outside the grid (zero padding) reads it instead of being masked element by element
// afterwards - 8 selects per keypoint instead of 32 per 8-channel chunk (every instruction a wave issues besides its
// MFMAs takes issue time from the matrix pipe it shares with the other resident waves, DESIGN_HISTORY.md section 9)
__device__ float g_zero_row[SSLAM_C];

struct Taps {
    const float *src[4];   // the tap's feature row, or g_zero_row for a tap outside the grid
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
    // clamp in float first: far-out-of-range coordinates must not overflow the int conversion
    const int xi = (int)fminf(fmaxf(x0, -2.0f), (float)G), yi = (int)fminf(fmaxf(y0, -2.0f), (float)G);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xs = xi + (k & 1), ys = yi + (k >> 1);
        // a tap whose weight is exactly 0 is not fetched either: it reads the zero row (0 * 0 instead of f * 0: the same sum
        // for finite features).  The selector's keypoints are integer cells, and for two thirds of them the normalise /
        // un-normalise round trip of grid_sample returns the integer exactly - one tap instead of four;
        // most of the others need two.  The gather is the one HBM-bound phase of the descriptor kernels.
        const bool ok = !(xs < 0 || xs >= G || ys < 0 || ys >= G) && t.wt[k] != 0.0f;
        const int xc = min(max(xs, 0), G - 1), yc = min(max(ys, 0), G - 1);
        t.src[k] = ok ? feat_frame + ((long long)yc * G + xc) * SSLAM_C : g_zero_row;
    }
    return t;
}

__device__ __forceinline__ float4 ld4(const Taps &t, int k, int off) {
    return *reinterpret_cast<const float4 *>(t.src[k] + off);
}
__device__ __forceinline__ float4 blend4(const Taps &t, int off) {
    const float4 a = ld4(t, 0, off), b = ld4(t, 1, off), c = ld4(t, 2, off), d = ld4(t, 3, off);
    float4 o;
    o.x = ((a.x * t.wt[0] + b.x * t.wt[1]) + c.x * t.wt[2]) + d.x * t.wt[3];
    o.y = ((a.y * t.wt[0] + b.y * t.wt[1]) + c.y * t.wt[2]) + d.y * t.wt[3];
    o.z = ((a.z * t.wt[0] + b.z * t.wt[1]) + c.z * t.wt[2]) + d.z * t.wt[3];
    o.w = ((a.w * t.wt[0] + b.w * t.wt[1]) + c.w * t.wt[2]) + d.w * t.wt[3];
    return o;
}

}  // namespace

// common.h - shared device helpers for the gfx950 kernels of libsslam_hip.so.
//
// Numerics contract (see oracle/sslam_oracle.h): every kernel evaluates fp32 in one fixed, documented order so
// that the GPU result is bit-identical to the CPU oracle.  Compile with -ffp-contract=off; fused operations are
// written as __builtin_fmaf explicitly.  Division and sqrt are the correctly rounded forms (hipcc default,
// -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sslam_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern long long g_sslam_launches;      // diagnostic launch counter (sslam_launch_count): relaxed atomic adds, so that concurrent
                                        // callers on distinct streams never race on it - it is the library's only mutable state
static inline void sslam_count_launches(long long n) { __atomic_fetch_add(&g_sslam_launches, n, __ATOMIC_RELAXED); }

// TEST-ONLY knobs (A/B timing, forcing a launch form in the parity tests).  Each is read from the environment ONCE, when
// the library is loaded (lib.hip), never per call; tests flip them through sslam_test_set_knob.  Product callers leave
// them alone: every form a knob selects is bit-identical to the default one.
enum sslam_knob_id {
    KNOB_M1_VARIANT,          // SSLAM_M1_VARIANT          1: S per direction (no workspace), 2: S once + 64-bit key reduction
    KNOB_CONV_VARIANT,        // SSLAM_CONV_VARIANT        wave tiling of the stage-form conv
    KNOB_CONV_LATENCY_ROWS,   // SSLAM_CONV_LATENCY_ROWS   row count below which the latency forms run (0: throughput form)
    KNOB_CONV_LAT2_ROWS,      // SSLAM_CONV_LAT2_ROWS      row count below which latency form 2 runs (0: never)
    KNOB_CONV_NO_HALO,        // SSLAM_CONV_NO_HALO        stage form instead of the halo form
    KNOB_CONV_TAIL,           // SSLAM_CONV_TAIL           round size in tiles of the halo form (0: all big tiles)
    KNOB_CONVBF_NO_HALO,      // SSLAM_CONVBF_NO_HALO
    KNOB_CONVBF_TAIL,         // SSLAM_CONVBF_TAIL
    KNOB_CONVBF_VARIANT,      // SSLAM_CONVBF_VARIANT
    KNOB_VIT_NO_FUSED_MLP,    // SSLAM_VIT_NO_FUSED_MLP    the two-launch MLP
    KNOB_BN_FORM,             // SSLAM_BN_FORM             1: three-sweep BatchNorm kernel instead of the register-resident one
    KNOB_VIT_F32_NO_KEY_SPLIT,// SSLAM_VIT_F32_NO_KEY_SPLIT the one-pass attention also for launches of a few frames
    KNOB_RT_STOP,             // SSLAM_RT_STOP             probe builds only
    KNOB_COUNT
};
#define SSLAM_KNOB_UNSET (-0x7fffffffffffffffLL - 1)
extern long long g_sslam_knob[KNOB_COUNT];
static inline long long sslam_knob(int id, long long dflt) { const long long v = g_sslam_knob[id]; return v == SSLAM_KNOB_UNSET ? dflt : v; }
static inline bool sslam_knob_set(int id) { return g_sslam_knob[id] != SSLAM_KNOB_UNSET; }

#define SSLAM_CHECK_LAUNCH()                                   \
    do {                                                       \
        sslam_count_launches(1);                               \
        if (hipGetLastError() != hipSuccess) return SSLAM_E_LAUNCH; \
    } while (0)

// Position of logical k inside its group of 8 in the "KP8" LDS/weight image: (0,2,4,6,1,3,5,7).
// A lane of half h = lane>>5 reading the float4 at physical offset 8g + 4h gets k = 8g + 2s + h, s = 0..3:
// exactly the A[i][k] / B[k][j] operand element of MFMA step s of v_mfma_f32_32x32x2_f32, in natural k order.
__host__ __device__ inline int kp8(int k) { return (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1); }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of C/D register r for lane-half h (32x32 tile): (r&3) + 8*(r>>2) + 4*h
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// sum over the 32 lanes of a half-wave: xor offsets 16, 8, 4, 2, 1 (== oracle butterfly(q, 32))
__device__ __forceinline__ float bfly32(float v) {
    v = v + __shfl_xor(v, 16);
    v = v + __shfl_xor(v, 8);
    v = v + __shfl_xor(v, 4);
    v = v + __shfl_xor(v, 2);
    v = v + __shfl_xor(v, 1);
    return v;
}
// sum over all 64 lanes: xor offsets 32, 16, ..., 1 (== oracle butterfly(a, 64))
__device__ __forceinline__ float bfly64(float v) {
    v = v + __shfl_xor(v, 32);
    return bfly32(v);
}

// canonical exp (oracle ora_expf): Cody-Waite + degree-6 polynomial, fmaf only
__device__ __forceinline__ float sslam_expf(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float e = __builtin_fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    return e * __int_as_float((ni + 127) << 23);
}
__device__ __forceinline__ float sslam_sigmoid(float x) { return 1.0f / (1.0f + sslam_expf(-x)); }

// 8 consecutive logical k values (two float4 loads) -> the two float4 of the KP8 image
__device__ __forceinline__ void kp8_split(const float4 &lo, const float4 &hi, float4 &even, float4 &odd) {
    even = make_float4(lo.x, lo.z, hi.x, hi.z);
    odd = make_float4(lo.y, lo.w, hi.y, hi.w);
}

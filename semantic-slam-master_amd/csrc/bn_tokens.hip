// bn_tokens.hip - A2: drop the CLS/register tokens and apply BatchNorm1d over the (group*cells) x 384 token matrix.
// Replaces DinoBackbone.forward after the ViT call (reference semantic-slam/models/dino_backbone.py:91-106).
//
// Grid (n_groups, 6): one workgroup (4 waves) owns 64 channels of one statistics group.  Lane l of wave w reads the
// float4 at channels 4*(l&15).. of rows r = p + 16*i with p = 4*w + (l>>4): every wave-instruction is four fully
// used 256-B row segments.  Column sums follow the canonical tree of oracle ora_bn_tokens: 16 sequential partials,
// xor-16 / xor-32 butterfly inside the wave, then the four wave sums in order through LDS.
// HBM-bound: reads the tokens (L2/MALL resident for the 2nd and 3rd sweep) and writes cells*384*4 B per frame.
#include "common.h"

namespace {

constexpr int U = 7;   // rows in flight per lane (784 cells = 16 lanes-rows x 49 = 7 x 7)

__device__ __forceinline__ float4 comb16(float4 v, float (*sh)[64], int wave, int lane) {
    // v: this lane's partial P[4*wave + (lane>>4)] for 4 channels
    float4 s;
    s.x = v.x + __shfl_xor(v.x, 16); s.y = v.y + __shfl_xor(v.y, 16);
    s.z = v.z + __shfl_xor(v.z, 16); s.w = v.w + __shfl_xor(v.w, 16);
    s.x = s.x + __shfl_xor(s.x, 32); s.y = s.y + __shfl_xor(s.y, 32);
    s.z = s.z + __shfl_xor(s.z, 32); s.w = s.w + __shfl_xor(s.w, 32);
    __syncthreads();  // previous use of sh is over
    if (lane < 16) {
        sh[wave][4 * lane + 0] = s.x; sh[wave][4 * lane + 1] = s.y;
        sh[wave][4 * lane + 2] = s.z; sh[wave][4 * lane + 3] = s.w;
    }
    __syncthreads();
    const int c = 4 * (lane & 15);
    float4 t;
    t.x = ((sh[0][c + 0] + sh[1][c + 0]) + sh[2][c + 0]) + sh[3][c + 0];
    t.y = ((sh[0][c + 1] + sh[1][c + 1]) + sh[2][c + 1]) + sh[3][c + 1];
    t.z = ((sh[0][c + 2] + sh[1][c + 2]) + sh[2][c + 2]) + sh[3][c + 2];
    t.w = ((sh[0][c + 3] + sh[1][c + 3]) + sh[2][c + 3]) + sh[3][c + 3];
    return t;
}

__global__ __launch_bounds__(256) void bn_tokens_kernel(const float *__restrict__ tokens, int tokens_per_frame,
                                                         int n_prefix, int group, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta,
                                                         const float *__restrict__ run_mean,
                                                         const float *__restrict__ run_var, int train, float eps,
                                                         float *__restrict__ out_feat, float *__restrict__ out_mean,
                                                         float *__restrict__ out_var, uint2 *__restrict__ out_bf16) {
    __shared__ float sh[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, ch0 = blockIdx.y * 64 + 4 * (lane & 15);
    const int cells = tokens_per_frame - n_prefix;
    const int R = group * cells;
    const int p = 4 * wave + (lane >> 4);
    const long long frame0 = (long long)g * group;

    auto src = [&](int r) {
        const int f = r / cells, t = r - f * cells;
        return reinterpret_cast<const float4 *>(tokens + ((frame0 + f) * tokens_per_frame + n_prefix + t) * SSLAM_C + ch0);
    };

    float4 mean, var;
    if (train) {
        // the sweeps are latency-bound unless several rows are in flight per lane: U loads are issued together, then
        // consumed in row order (the canonical sequential sum of oracle ora_bn_tokens is unchanged)
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = p; r < R; r += 16 * U) {
            float4 x[U];
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = *src(min(r + 16 * u, R - 1));
#pragma unroll
            for (int u = 0; u < U; u++)
                if (r + 16 * u < R) { s.x = s.x + x[u].x; s.y = s.y + x[u].y; s.z = s.z + x[u].z; s.w = s.w + x[u].w; }
        }
        const float4 tot = comb16(s, sh, wave, lane);
        const float fr = (float)R;
        mean = make_float4(tot.x / fr, tot.y / fr, tot.z / fr, tot.w / fr);
        s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = p; r < R; r += 16 * U) {
            float4 x[U];
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = *src(min(r + 16 * u, R - 1));
#pragma unroll
            for (int u = 0; u < U; u++)
                if (r + 16 * u < R) {
                    const float dx = x[u].x - mean.x, dy = x[u].y - mean.y, dz = x[u].z - mean.z, dw = x[u].w - mean.w;
                    s.x = __builtin_fmaf(dx, dx, s.x); s.y = __builtin_fmaf(dy, dy, s.y);
                    s.z = __builtin_fmaf(dz, dz, s.z); s.w = __builtin_fmaf(dw, dw, s.w);
                }
        }
        const float4 tot2 = comb16(s, sh, wave, lane);
        var = make_float4(tot2.x / fr, tot2.y / fr, tot2.z / fr, tot2.w / fr);
        if (tid < 16) {
            if (out_mean) *reinterpret_cast<float4 *>(out_mean + (long long)g * SSLAM_C + ch0) = mean;
            if (out_var) *reinterpret_cast<float4 *>(out_var + (long long)g * SSLAM_C + ch0) = var;
        }
    } else {
        mean = *reinterpret_cast<const float4 *>(run_mean + ch0);
        var = *reinterpret_cast<const float4 *>(run_var + ch0);
    }
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + ch0);
    const float4 be = *reinterpret_cast<const float4 *>(beta + ch0);
    float4 al, bs;
    al.x = (1.0f / sqrtf(var.x + eps)) * ga.x; al.y = (1.0f / sqrtf(var.y + eps)) * ga.y;
    al.z = (1.0f / sqrtf(var.z + eps)) * ga.z; al.w = (1.0f / sqrtf(var.w + eps)) * ga.w;
    bs.x = be.x - mean.x * al.x; bs.y = be.y - mean.y * al.y;
    bs.z = be.z - mean.z * al.z; bs.w = be.w - mean.w * al.w;
    for (int r0 = p; r0 < R; r0 += 16 * U) {
        float4 xs[U];
#pragma unroll
        for (int u = 0; u < U; u++) xs[u] = *src(min(r0 + 16 * u, R - 1));
#pragma unroll
        for (int u = 0; u < U; u++) {
        const int r = r0 + 16 * u;
        if (r >= R) break;
        const float4 x = xs[u];
        float4 y;
        y.x = x.x * al.x + bs.x; y.y = x.y * al.y + bs.y; y.z = x.z * al.z + bs.z; y.w = x.w * al.w + bs.w;
        *reinterpret_cast<float4 *>(out_feat + (frame0 * cells + r) * SSLAM_C + ch0) = y;
        if (out_bf16) {   // bf16 copy for the throughput-mode saliency CNN (selector_bf16.hip): 4 values = 8 B per lane
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const f32x2 lo = {y.x, y.y}, hi = {y.z, y.w};
            uint2 o;
            o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
            o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
            out_bf16[((frame0 * cells + r) * SSLAM_C + ch0) / 4] = o;
        }
        }
    }
}

// Register-resident form for group == 1: the three sweeps of the kernel above re-read the tokens from MALL/HBM (a CU's
// resident workgroups cover far more than its L2 share), which makes that kernel bandwidth-bound at 3x the compulsory read
// traffic.  Here every lane keeps its rows in VGPRs (and, for the largest grids, the last NL row steps in LDS) between the
// sweeps: one HBM read, one write.  Same partition (16 partials by row & 15, each a sequential sum in row order) and the
// same combination tree as the sweep kernel, so the results are bit-identical.
//   V  = channels per lane (4: a workgroup owns 64 channels, 2: 32 channels; 16 lanes x V x 4 B is one 256 / 128-byte row
//        segment per wave-quarter, always whole cache lines)
//   NR = row steps (of 16 rows) held in registers, NL = further row steps held in LDS; cells <= 16 * (NR + NL)
//   G = 28: <4, 49, 0>, two workgroups per CU.  G = 40 (1 600 cells): <4, 100, 0> = 400 VGPRs, one wave per SIMD - a lane
//   has 100 independent 16-byte loads in flight, which is all the latency hiding a streaming kernel needs.  G = 60 (3 600
//   cells): a 64-channel slice is 921 KB, more than a CU's 512 KB of VGPRs: <2, 200, 25> keeps a 32-channel slice in 400
//   VGPRs + 51 KB of LDS.
// Addresses are one uniform base per row step plus a 32-bit lane offset (no per-row address registers).
template <int V>
struct VecT;
template <>
struct VecT<4> { typedef float4 T; typedef float N __attribute__((ext_vector_type(4))); };
template <>
struct VecT<2> { typedef float2 T; typedef float N __attribute__((ext_vector_type(2))); };
// the tokens are read exactly once: non-temporal loads keep them from displacing the lines other kernels' operands live in
template <int V>
__device__ __forceinline__ typename VecT<V>::T load_once(const float *p) {
    return __builtin_bit_cast(typename VecT<V>::T, __builtin_nontemporal_load(reinterpret_cast<const typename VecT<V>::N *>(p)));
}

template <int V>
__device__ __forceinline__ void comb16v(const float (&v)[V], float (&t)[V], float (*sh)[64], int wave, int lane) {
    float s[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        s[j] = v[j] + __shfl_xor(v[j], 16);
        s[j] = s[j] + __shfl_xor(s[j], 32);
    }
    __syncthreads();  // previous use of sh is over
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < V; j++) sh[wave][V * lane + j] = s[j];
    }
    __syncthreads();
    const int c = V * (lane & 15);
#pragma unroll
    for (int j = 0; j < V; j++) t[j] = ((sh[0][c + j] + sh[1][c + j]) + sh[2][c + j]) + sh[3][c + j];
}

template <int V, int NR, int NL, int WPS>
__global__ __launch_bounds__(256, WPS) void bn_tokens_reg_kernel(const float *__restrict__ tokens, int tokens_per_frame,
                                                                  int n_prefix, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float eps,
                                                                  float *__restrict__ out_feat, float *__restrict__ out_mean,
                                                                  float *__restrict__ out_var, unsigned *__restrict__ out_bf16) {
    typedef typename VecT<V>::T vec;
    __shared__ float sh[4][64];
    __shared__ vec hold[NL > 0 ? NL : 1][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, ch0 = blockIdx.y * (16 * V) + V * (lane & 15);
    const int R = tokens_per_frame - n_prefix;
    const int p = 4 * wave + (lane >> 4);
    const int full = R / 16, rem = R % 16;          // row steps valid for every lane / lanes valid in step `full`
    const float *src = tokens + ((long long)g * tokens_per_frame + n_prefix) * SSLAM_C;     // uniform
    const unsigned loff = (unsigned)(p * SSLAM_C + ch0);                                    // per lane, floats
    auto valid = [&](int i) { return i < full || (i == full && p < rem); };
    float x[NR][V];
    if (NL > 0) {
        vec t[NL > 0 ? NL : 1];                 // all NL loads in flight together, then the LDS stores
#pragma unroll
        for (int i = 0; i < NL; i++) {
            t[i] = vec{};
            if (valid(NR + i)) t[i] = load_once<V>(src + (size_t)(NR + i) * 16 * SSLAM_C + loff);
        }
#pragma unroll
        for (int i = 0; i < NL; i++) hold[i][tid] = t[i];      // own slots only: no barrier between these stores and the loads below
    }
#pragma unroll
    for (int i = 0; i < NR; i++) {
        vec v = {};
        if (valid(i)) v = load_once<V>(src + (size_t)i * 16 * SSLAM_C + loff);
#pragma unroll
        for (int j = 0; j < V; j++) x[i][j] = reinterpret_cast<const float *>(&v)[j];
    }
    float s[V], tot[V], mean[V], var[V];
#pragma unroll
    for (int j = 0; j < V; j++) s[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NR; i++)
        if (valid(i)) {
#pragma unroll
            for (int j = 0; j < V; j++) s[j] = s[j] + x[i][j];
        }
    if (NL > 0) {
#pragma unroll
        for (int i = NR; i < NR + NL; i++)
            if (valid(i)) {
                const vec v = hold[i - NR][tid];
#pragma unroll
                for (int j = 0; j < V; j++) s[j] = s[j] + reinterpret_cast<const float *>(&v)[j];
            }
    }
    comb16v<V>(s, tot, sh, wave, lane);
    const float fr = (float)R;
#pragma unroll
    for (int j = 0; j < V; j++) { mean[j] = tot[j] / fr; s[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NR; i++)
        if (valid(i)) {
#pragma unroll
            for (int j = 0; j < V; j++) { const float d = x[i][j] - mean[j]; s[j] = __builtin_fmaf(d, d, s[j]); }
        }
    if (NL > 0) {
#pragma unroll
        for (int i = NR; i < NR + NL; i++)
            if (valid(i)) {
                const vec v = hold[i - NR][tid];
#pragma unroll
                for (int j = 0; j < V; j++) { const float d = reinterpret_cast<const float *>(&v)[j] - mean[j]; s[j] = __builtin_fmaf(d, d, s[j]); }
            }
    }
    comb16v<V>(s, tot, sh, wave, lane);
#pragma unroll
    for (int j = 0; j < V; j++) var[j] = tot[j] / fr;
    if (tid < 16) {
#pragma unroll
        for (int j = 0; j < V; j++) {
            if (out_mean) out_mean[(long long)g * SSLAM_C + ch0 + j] = mean[j];
            if (out_var) out_var[(long long)g * SSLAM_C + ch0 + j] = var[j];
        }
    }
    float al[V], bs[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        al[j] = (1.0f / sqrtf(var[j] + eps)) * gamma[ch0 + j];
        bs[j] = beta[ch0 + j] - mean[j] * al[j];
    }
    float *dst = out_feat + (long long)g * R * SSLAM_C;                                      // uniform
    unsigned *dst_bf = out_bf16 ? out_bf16 + (long long)g * R * (SSLAM_C / 2) : nullptr;     // one dword = two bf16
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    auto emit = [&](int i, const float (&xv)[V]) {
        float y[V];
#pragma unroll
        for (int j = 0; j < V; j++) y[j] = xv[j] * al[j] + bs[j];
        __builtin_nontemporal_store(*reinterpret_cast<const typename VecT<V>::N *>(y), reinterpret_cast<typename VecT<V>::N *>(dst + (size_t)i * 16 * SSLAM_C + loff));
        if (dst_bf) {   // bf16 copy for the throughput-mode saliency CNN (selector_bf16.hip)
            unsigned o[V / 2];
#pragma unroll
            for (int j = 0; j < V / 2; j++) {
                const f32x2 pr = {y[2 * j], y[2 * j + 1]};
                o[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
            }
            unsigned *q = dst_bf + (size_t)i * 16 * (SSLAM_C / 2) + loff / 2;
            if (V == 4) *reinterpret_cast<uint2 *>(q) = make_uint2(o[0], o[V / 2 - 1]);
            else *q = o[0];
        }
    };
#pragma unroll
    for (int i = 0; i < NR; i++)
        if (valid(i)) emit(i, x[i]);
    if (NL > 0) {
#pragma unroll
        for (int i = NR; i < NR + NL; i++)
            if (valid(i)) {
                const vec v = hold[i - NR][tid];
                float xv[V];
#pragma unroll
                for (int j = 0; j < V; j++) xv[j] = reinterpret_cast<const float *>(&v)[j];
                emit(i, xv);
            }
    }
}

}  // namespace

static int bn_launch(const float *tokens, int n_frames, int tokens_per_frame, int n_prefix, int group,
                     const float *gamma, const float *beta, const float *run_mean, const float *run_var,
                     int train, float eps, float *out_feat, float *out_mean, float *out_var, void *out_bf16, void *stream) {
    if (!tokens || !gamma || !beta || !out_feat || n_frames <= 0 || group <= 0 || n_prefix < 0 ||
        tokens_per_frame <= n_prefix)
        return SSLAM_E_INVALID;
    if (!train && (!run_mean || !run_var)) return SSLAM_E_INVALID;
    if (n_frames % group) return SSLAM_E_INVALID;
    if (((uintptr_t)tokens | (uintptr_t)out_feat | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)run_mean |
         (uintptr_t)run_var | (uintptr_t)out_mean | (uintptr_t)out_var | (uintptr_t)out_bf16) & 15)
        return SSLAM_E_INVALID;
    if ((long long)group * (tokens_per_frame - n_prefix) > 0x7fffffffLL) return SSLAM_E_UNSUPPORTED;
    const int cells = tokens_per_frame - n_prefix;
    if (train && group == 1 && cells <= 16 * 225 && sslam_knob(KNOB_BN_FORM, 0) != 1) {
        // one HBM read instead of three: every lane keeps its rows on the chip between the sweeps (bn_tokens_reg_kernel)
        hipStream_t st = (hipStream_t)stream;
        if (cells <= 16 * 49)
            hipLaunchKernelGGL((bn_tokens_reg_kernel<4, 49, 0, 2>), dim3(n_frames, SSLAM_C / 64), dim3(256), 0, st, tokens, tokens_per_frame,
                               n_prefix, gamma, beta, eps, out_feat, out_mean, out_var, (unsigned *)out_bf16);
        else if (cells <= 16 * 100)
            hipLaunchKernelGGL((bn_tokens_reg_kernel<4, 100, 0, 1>), dim3(n_frames, SSLAM_C / 64), dim3(256), 0, st, tokens, tokens_per_frame,
                               n_prefix, gamma, beta, eps, out_feat, out_mean, out_var, (unsigned *)out_bf16);
        else
            hipLaunchKernelGGL((bn_tokens_reg_kernel<2, 200, 25, 1>), dim3(n_frames, SSLAM_C / 32), dim3(256), 0, st, tokens, tokens_per_frame,
                               n_prefix, gamma, beta, eps, out_feat, out_mean, out_var, (unsigned *)out_bf16);
        SSLAM_CHECK_LAUNCH();
        return SSLAM_OK;
    }
    hipLaunchKernelGGL(bn_tokens_kernel, dim3(n_frames / group, SSLAM_C / 64), dim3(256), 0, (hipStream_t)stream, tokens,
                       tokens_per_frame, n_prefix, group, gamma, beta, run_mean, run_var, train, eps, out_feat,
                       out_mean, out_var, (uint2 *)out_bf16);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

extern "C" int sslam_bn_tokens(const float *tokens, int n_frames, int tokens_per_frame, int n_prefix, int group,
                               const float *gamma, const float *beta, const float *run_mean, const float *run_var,
                               int train, float eps, float *out_feat, float *out_mean, float *out_var, void *stream) {
    return bn_launch(tokens, n_frames, tokens_per_frame, n_prefix, group, gamma, beta, run_mean, run_var, train, eps, out_feat,
                     out_mean, out_var, nullptr, stream);
}

// the same kernel, additionally writing the bf16 (round-to-nearest-even) copy of out_feat that the bf16-mode saliency CNN reads
extern "C" int sslam_bn_tokens_bf16copy(const float *tokens, int n_frames, int tokens_per_frame, int n_prefix, int group,
                                        const float *gamma, const float *beta, const float *run_mean, const float *run_var,
                                        int train, float eps, float *out_feat, void *out_feat_bf16, float *out_mean,
                                        float *out_var, void *stream) {
    if (!out_feat_bf16) return SSLAM_E_INVALID;
    return bn_launch(tokens, n_frames, tokens_per_frame, n_prefix, group, gamma, beta, run_mean, run_var, train, eps, out_feat,
                     out_mean, out_var, out_feat_bf16, stream);
}

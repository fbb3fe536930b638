// match.hip - M1..M5 core: cosine-similarity GEMM fused with row / column arg-max, then the mutual check,
// thresholds, quality score and ordered compaction of M1.
// Replaces torch.mm + argmax(dim=1) + argmax(dim=0) + mutual/threshold/quality code of
// SequenceMatcher.match_with_quality (reference semantic-slam/visualize_matches_sequence.py:138-197) and the sibling
// matchers (visualize_matches.py:105-109, train.py:423-425, test/test_descriptor_quality.py:116-123,
// test/test_tracking.py:159-160).
//
// sim_argmax: grid (query blocks of 128, direction, pair).  Direction 0 finds for every row of d1 its best row of
// d2 (nn12), direction 1 swaps the roles (nn21): S[i][j] and S^T[j][i] are the same fma chain (a*b commutes), so
// both directions see bit-identical similarities and the K x K matrix is never written to memory.
// The MFMA tile is oriented with the QUERY on the lane (B operand) and the candidates on the accumulator rows
// (A operand): each lane then owns one query and scans its 16 candidate rows per tile in increasing index with a
// strict '>' - a register-local arg-max that keeps the first maximum (torch / numpy semantics); only
// the two half-waves and the two candidate waves are merged at the end (value, then lower index).
// Roofline: MFMA-bound (2 * n1 * n2 * 128 * 2 FLOP per pair incl. both directions; 128 MFLOP at 500 keypoints).
#include "common.h"

namespace {

constexpr int QB = 128;            // queries per workgroup
constexpr int CB = 64;             // candidates per stage
constexpr int LDD = SSLAM_D + 4;   // 132-float rows: 528 B = 33 x 16 B -> conflict-free b128 fragment reads

// rows x 128 floats -> KP8 image, split into a load half (global -> registers) and a store half (registers -> LDS)
// so that the next candidate tile is in flight while the current one is multiplied.
// rows beyond n_valid are zero (they are masked out of the arg-max anyway)
template <int ROWS>
struct Stager {
    static constexpr int ITEMS = ROWS * 16 / 512;
    float4 lo[ITEMS], hi[ITEMS];
    bool ok[ITEMS];         // rows beyond n_valid are zeroed at store time: nothing waits for the loads before the MFMAs
    __device__ __forceinline__ void load(const float *__restrict__ src, int first, int n_valid, int tid) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int it = tid + 512 * i, row = it >> 4, g = it & 15;
            ok[i] = first + row < n_valid;
            const float4 *p = reinterpret_cast<const float4 *>(src + (long long)(ok[i] ? first + row : 0) * SSLAM_D + 8 * g);
            lo[i] = p[0];
            hi[i] = p[1];
        }
    }
    __device__ __forceinline__ void store(float *dst, int tid) const {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int it = tid + 512 * i, row = it >> 4, g = it & 15;
            float4 ev, od;
            const bool k = ok[i];
            kp8_split(make_float4(k ? lo[i].x : 0.f, k ? lo[i].y : 0.f, k ? lo[i].z : 0.f, k ? lo[i].w : 0.f),
                      make_float4(k ? hi[i].x : 0.f, k ? hi[i].y : 0.f, k ? hi[i].z : 0.f, k ? hi[i].w : 0.f), ev, od);
            *reinterpret_cast<float4 *>(dst + row * LDD + 8 * g) = ev;
            *reinterpret_cast<float4 *>(dst + row * LDD + 8 * g + 4) = od;
        }
    }
};

// monotone 64-bit image of (similarity, query index): "larger value, then lower index" is one unsigned compare, so the
// column direction can be reduced with max() in any order (lanes, waves, workgroups) and still give the first maximum
__device__ __forceinline__ unsigned long long sim_key(float v, int i) {
    unsigned u = __float_as_uint(v + 0.0f);                 // -0 -> +0: equal under the float compare of the reference
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned)(~i);
}

// ONEPASS = false: grid (query blocks, 2 directions, pairs) - S is evaluated once per direction, nothing but the outputs
//                  is written (single pairs / small batches: twice the workgroups to spread over the CUs).
// ONEPASS = true:  grid (query blocks, 1, pairs) - S is evaluated ONCE; the column direction (nn21) is reduced over the
//                  32 query lanes of each half-wave by a reduce-scatter butterfly on sim_key()s and merged across waves
//                  and workgroups with a 64-bit atomic max into `keys` (n_pairs x n2, zeroed), decoded by keys_decode_kernel.
template <bool ONEPASS>
__global__ __launch_bounds__(512) void sim_argmax_kernel(const float *__restrict__ desc1, long long stride1, int n1,
                                                          const float *__restrict__ desc2, long long stride2, int n2,
                                                          int *__restrict__ nn12, float *__restrict__ s12,
                                                          int *__restrict__ nn21, float *__restrict__ s21,
                                                          float *__restrict__ second12,
                                                          unsigned long long *__restrict__ keys) {
    __shared__ __attribute__((aligned(16))) float smem[(QB + 2 * CB) * LDD];
    float *Qs = smem, *Cs = smem + QB * LDD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wq = wave & 3, wc = wave >> 2;
    const int dir = blockIdx.y;
    const long long pair = blockIdx.z;
    const float *q = dir == 0 ? desc1 + pair * stride1 : desc2 + pair * stride2;
    const float *c = dir == 0 ? desc2 + pair * stride2 : desc1 + pair * stride1;
    const int nq = dir == 0 ? n1 : n2, nc = dir == 0 ? n2 : n1;
    const int q0 = blockIdx.x * QB;
    if (q0 >= nq) return;
    int *o_idx = (dir == 0 ? nn12 + pair * n1 : nn21 + pair * n2);
    float *o_val = dir == 0 ? (s12 ? s12 + pair * n1 : nullptr) : (s21 ? s21 + pair * n2 : nullptr);
    float *o_sec = (dir == 0 && second12) ? second12 + pair * n1 : nullptr;

    {
        Stager<QB> sq;
        sq.load(q, q0, nq, tid);
        sq.store(Qs, tid);
    }
    Stager<CB> sc;
    sc.load(c, 0, nc, tid);
    sc.store(Cs, tid);
    __syncthreads();

    float best = -INFINITY, second = -INFINITY;   // second: best of the row once the winner is removed
    int besti = 0x7fffffff;
    const int nstage = (nc + CB - 1) / CB;
    const float *B = Qs + (wq * 32 + r) * LDD + 4 * h;
    for (int s = 0; s < nstage; s++) {
        if (s + 1 < nstage) sc.load(c, (s + 1) * CB, nc, tid);     // in flight during the MFMAs below
        const float *A = Cs + (s & 1) * CB * LDD + (wc * 32 + r) * LDD + 4 * h;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.0f;
#pragma unroll
        for (int g = 0; g < SSLAM_D / 8; g++) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(A + 8 * g);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(B + 8 * g);
#pragma unroll
            for (int st = 0; st < 4; st++) acc = mfma32(a[st], b[st], acc);
        }
        const int jbase = s * CB + wc * 32;
#pragma unroll
        for (int e = 0; e < 16; e++) {   // branch-free form of: if (v > best) {second = best; best = v; besti = j;}
            const int j = jbase + crow(e, h);  //                  else if (v > second) second = v;      (rows j >= nc skipped)
            const float v = j < nc ? acc[e] : -INFINITY;
            const bool gt = v > best;
            second = gt ? best : fmaxf(second, v);
            besti = gt ? j : besti;
            best = gt ? v : best;
        }
        if (ONEPASS) {
            const int qi = q0 + wq * 32 + r;
            const bool qok = qi < nq;
            unsigned long long k[8];
            {
                const bool up = (r & 16) != 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    unsigned long long lo = qok ? sim_key(acc[i], qi) : 0ull, hi = qok ? sim_key(acc[i + 8], qi) : 0ull;
                    asm volatile("" : "+v"(lo), "+v"(hi));      // keeps the selects from being folded into a dynamic vector index
                    const unsigned long long keep = up ? hi : lo, send = up ? lo : hi;
                    const unsigned long long o = __shfl_xor(send, 16);
                    k[i] = keep > o ? keep : o;
                }
            }
#pragma unroll
            for (int m = 4; m >= 1; m >>= 1) {
                const bool up = (r & (2 * m)) != 0;
#pragma unroll
                for (int i = 0; i < m; i++) {
                    unsigned long long lo = k[i], hi = k[i + m];
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    const unsigned long long keep = up ? hi : lo, send = up ? lo : hi;
                    const unsigned long long o = __shfl_xor(send, 2 * m);
                    k[i] = keep > o ? keep : o;
                }
            }
            const unsigned long long o = __shfl_xor(k[0], 1);
            const unsigned long long kk = k[0] > o ? k[0] : o;
            const int j = jbase + crow((r >> 1) & 15, h);       // the accumulator row this lane pair ended up holding
            if (!(r & 1) && j < nc) atomicMax(keys + pair * n2 + j, kk);
        }
        if (s + 1 < nstage) sc.store(Cs + ((s + 1) & 1) * CB * LDD, tid);
        __syncthreads();
    }
    // merge the two half-waves (same query, interleaved candidate rows), then the two candidate waves
    {
        const float ov = __shfl_xor(best, 32), os = __shfl_xor(second, 32);
        const int oi = __shfl_xor(besti, 32);
        if (ov > best || (ov == best && oi < besti)) {
            second = fmaxf(best, os);     // the loser's best competes with the winner's runner-up
            best = ov;
            besti = oi;
        } else {
            second = fmaxf(second, ov);
        }
    }
    float *mv = smem;                                     // [2][128]
    int *mi = reinterpret_cast<int *>(smem + 2 * QB);      // [2][128]
    float *ms = smem + 4 * QB;                             // [2][128]
    if (h == 0) {
        mv[wc * QB + wq * 32 + r] = best;
        mi[wc * QB + wq * 32 + r] = besti;
        ms[wc * QB + wq * 32 + r] = second;
    }
    __syncthreads();
    if (tid < QB && q0 + tid < nq) {
        float v = mv[tid], sc = ms[tid];
        int i = mi[tid];
        const float v1 = mv[QB + tid], sc1 = ms[QB + tid];
        const int i1 = mi[QB + tid];
        if (v1 > v || (v1 == v && i1 < i)) {
            sc = fmaxf(v, sc1);
            v = v1;
            i = i1;
        } else {
            sc = fmaxf(sc, v1);
        }
        o_idx[q0 + tid] = i;
        if (o_val) o_val[q0 + tid] = v;
        if (o_sec) o_sec[q0 + tid] = sc;
    }
}

__global__ __launch_bounds__(256) void keys_decode_kernel(const unsigned long long *__restrict__ keys, long long n,
                                                           int *__restrict__ nn21, float *__restrict__ s21) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long kk = keys[i];
    unsigned u = (unsigned)(kk >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    nn21[i] = (int)(~(unsigned)kk);
    if (s21) s21[i] = __uint_as_float(u);
}

// one workgroup per pair: mutual check, thresholds, quality, ordered compaction (ascending idx1)
__global__ __launch_bounds__(256) void match_finalize_kernel(const int *__restrict__ nn12, const float *__restrict__ s12,
                                                              const int *__restrict__ nn21, int n1, int n2,
                                                              const float *__restrict__ sc1, long long ss1,
                                                              const float *__restrict__ sc2, long long ss2,
                                                              const float *__restrict__ in1, const float *__restrict__ in2,
                                                              float w_desc, float w_sal, float t_sal, float t_sim,
                                                              float t_int, long long *__restrict__ matches,
                                                              float *__restrict__ quality, int *__restrict__ count) {
    __shared__ int wave_tot[4];
    __shared__ int running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long p = blockIdx.x;
    nn12 += p * n1;
    s12 += p * n1;
    nn21 += p * n2;
    sc1 += p * ss1;
    sc2 += p * ss2;
    if (in1) in1 += p * ss1;
    if (in2) in2 += p * ss2;
    matches += p * n1 * 2;
    quality += p * n1;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < n1; base += 256) {
        const int i = base + tid;
        bool ok = false;
        int j = 0;
        float sim = 0.f, avg = 0.f;
        if (i < n1) {
            j = nn12[i];
            if (nn21[j] == i) {                                               // :149
                sim = s12[i];
                avg = (sc1[i] + sc2[j]) / 2.0f;                               // :163
                ok = (avg >= t_sal) && (sim >= t_sim);                        // :166-168
                if (in1 && in2) ok = ok && ((in1[i] + in2[j]) / 2.0f >= t_int);   // :171-176
            }
        }
        int inc = ok ? 1 : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        int off0 = running;
        for (int w = 0; w < wave; w++) off0 += wave_tot[w];
        if (ok) {
            const int slot = off0 + inc - 1;
            matches[2 * slot] = i;
            matches[2 * slot + 1] = j;
            quality[slot] = w_desc * sim + w_sal * avg;                       // :189-192
        }
        __syncthreads();
        if (tid == 0) running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (tid == 0) count[p] = running;
}

}  // namespace

extern "C" int sslam_sim_argmax(const float *desc1, long long stride1, int n1, const float *desc2, long long stride2,
                                int n2, int n_pairs, int32_t *nn12, float *s12, int32_t *nn21, float *s21, float *second12,
                                void *stream) {
    if (!desc1 || !desc2 || !nn12 || !nn21 || n1 <= 0 || n2 <= 0 || n_pairs <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)desc1 | (uintptr_t)desc2) & 15 || (stride1 & 3) || (stride2 & 3)) return SSLAM_E_INVALID;
    if (n_pairs > 65535) return SSLAM_E_UNSUPPORTED;
    // variant: 0 = by batch size, 1 = S per direction (no scratch memory), 2 = S once + 64-bit key reduction
    const char *env = getenv("SSLAM_M1_VARIANT");
    const int forced = env ? atoi(env) : 0;
    hipStream_t st = (hipStream_t)stream;
    if (forced == 2 || (forced == 0 && n_pairs >= 16)) {
        unsigned long long *keys = nullptr;
        const size_t bytes = (size_t)n_pairs * n2 * sizeof(unsigned long long);
        if (hipMallocAsync((void **)&keys, bytes, st) != hipSuccess) return SSLAM_E_LAUNCH;   // stream-ordered scratch
        if (hipMemsetAsync(keys, 0, bytes, st) != hipSuccess) {
            (void)hipFreeAsync(keys, st);
            return SSLAM_E_LAUNCH;
        }
        hipLaunchKernelGGL(sim_argmax_kernel<true>, dim3((n1 + QB - 1) / QB, 1, n_pairs), dim3(512), 0, st, desc1, stride1, n1,
                           desc2, stride2, n2, nn12, s12, nn21, s21, second12, keys);
        g_sslam_launches++;
        bool ok = hipGetLastError() == hipSuccess;
        if (ok) {
            const long long n = (long long)n_pairs * n2;
            hipLaunchKernelGGL(keys_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, n, nn21, s21);
            g_sslam_launches++;
            ok = hipGetLastError() == hipSuccess;
        }
        if (hipFreeAsync(keys, st) != hipSuccess) ok = false;     // the scratch is released on the failure paths too
        return ok ? SSLAM_OK : SSLAM_E_LAUNCH;
    }
    const int nmax = n1 > n2 ? n1 : n2;
    hipLaunchKernelGGL(sim_argmax_kernel<false>, dim3((nmax + QB - 1) / QB, 2, n_pairs), dim3(512), 0, st, desc1, stride1, n1,
                       desc2, stride2, n2, nn12, s12, nn21, s21, second12, nullptr);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

extern "C" int sslam_match_finalize(const int32_t *nn12, const float *s12, const int32_t *nn21, int n1, int n2, int n_pairs,
                                    const float *scores1, long long sstride1, const float *scores2, long long sstride2,
                                    const float *intensity1, const float *intensity2, float w_desc, float w_sal,
                                    float min_saliency, float min_sim, float min_intensity, int64_t *matches,
                                    float *quality, int32_t *count, void *stream) {
    if (!nn12 || !s12 || !nn21 || !scores1 || !scores2 || !matches || !quality || !count || n1 <= 0 || n2 <= 0 ||
        n_pairs <= 0)
        return SSLAM_E_INVALID;
    hipLaunchKernelGGL(match_finalize_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, nn12, s12, nn21, n1, n2,
                       scores1, sstride1, scores2, sstride2, intensity1, intensity2, w_desc, w_sal, min_saliency, min_sim,
                       min_intensity, (long long *)matches, quality, count);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

// select.hip - A4 + A5 (+ A8): NMS, percentile threshold, branchy top-k / padding - one workgroup per frame, with
// the whole data-dependent control flow on the device (the reference syncs to the host at every .item() / nonzero).
// Replaces KeypointSelector.select_keypoints and _apply_nms (reference semantic-slam/models/keypoint_selector.py:
// 69-226) and DinoBackbone.patch_to_pixel (dino_backbone.py:154-165).
//
// Ordering rule: value descending, flat index ascending.  One bitonic sort of 64-bit keys
// (orderable-float bits << 32 | ~index) of the raw saliency gives, in LDS: the order statistics for every
// torch.quantile call, the "top-k of raw saliency" lists of the padding branches, and - by a prefix sum over a
// membership flag walked in sorted order - the top-k of any subset (NMS survivors above a threshold).
// Integer / compare work only, apart from the quantile lerp: bit-exact against the oracle by construction.
#include "common.h"

namespace {

constexpr int NT = 256;       // threads
constexpr int MAXN = 4096;    // cells per frame supported (64x64 grid)

__device__ __forceinline__ unsigned orderable(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Scan {
    int wave_tot[NT / 64];
    int total;
};

// exclusive prefix sum of one int per thread (thread order), also returns the block total
__device__ __forceinline__ int block_excl_scan(int v, Scan &sc, int tid, int &total) {
    const int lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    __syncthreads();
    if (lane == 63) sc.wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        if (w < wave) base += sc.wave_tot[w];
        tot += sc.wave_tot[w];
    }
    total = tot;
    return base + inc - v;
}

// torch.quantile(linear) on the descending-sorted keys (oracle quantile_sorted_desc)
__device__ __forceinline__ float key_val(unsigned long long k) {
    const unsigned u = (unsigned)(k >> 32);
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ float quantile_desc(const unsigned long long *keys, int n, float q) {
    const float rank = q * (float)(n - 1);
    const float lo = floorf(rank), hi = ceilf(rank);
    const float w = rank - lo;
    const float a = key_val(keys[n - 1 - (int)lo]), b = key_val(keys[n - 1 - (int)hi]);
    const float diff = b - a;
    return (w < 0.5f) ? __builtin_fmaf(w, diff, a) : __builtin_fmaf(-diff, 1.0f - w, b);
}

__global__ __launch_bounds__(NT) void select_keypoints_kernel(const float *__restrict__ sal_all, int G, int K, int radius,
                                                               float pct, float *__restrict__ kp_xy,
                                                               float *__restrict__ scores, int *__restrict__ idx_out,
                                                               float *__restrict__ kp_pixel, int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int n = G * G;
    int P = 1;
    while (P < n) P <<= 1;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(dyn);            // P
    float *sal = reinterpret_cast<float *>(keys + P);                                   // n
    float *nmsv = sal + n;                                                              // n
    __shared__ Scan sc;
    __shared__ float s_thr[5];
    __shared__ int s_cnt[5];

    const int tid = threadIdx.x;
    const long long f = blockIdx.x;
    const float *src = sal_all + f * n;
    float *o_kp = kp_xy + f * K * 2, *o_sc = scores + f * K;
    int *o_idx = idx_out ? idx_out + f * K : nullptr;
    float *o_px = kp_pixel ? kp_pixel + f * K * 2 : nullptr;

    for (int i = tid; i < P; i += NT) {
        if (i < n) {
            const float v = src[i];
            sal[i] = v;
            keys[i] = ((unsigned long long)orderable(v) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        } else {
            keys[i] = 0ull;  // below every real key (real keys have a non-zero low word)
        }
    }
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool desc = (i & k) == 0;
                if (desc ? (a < b) : (a > b)) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }

    // thresholds (keypoint_selector.py:105-109, 139-141)
    if (tid < 5) {
        const float q = tid == 0 ? pct : (tid == 1 ? 0.40f : (tid == 2 ? 0.30f : (tid == 3 ? 0.20f : 0.10f)));
        const float fl = tid == 0 ? 0.1f : 0.05f;
        s_thr[tid] = fmaxf(quantile_desc(keys, n, q), fl);
        s_cnt[tid] = 0;
    }
    // NMS (:209-226): max over the (2r+1)^2 window clipped to the grid; survivors keep their value, others 0
    for (int i = tid; i < n; i += NT) {
        const int y = i / G, x = i - y * G;
        const float v = sal[i];
        float mx = v;
        if (radius > 0) {
            mx = -INFINITY;
            const int y0 = max(y - radius, 0), y1 = min(y + radius, G - 1);
            const int x0 = max(x - radius, 0), x1 = min(x + radius, G - 1);
            for (int yy = y0; yy <= y1; yy++)
                for (int xx = x0; xx <= x1; xx++) mx = fmaxf(mx, sal[yy * G + xx]);
        }
        nmsv[i] = radius > 0 ? v * (v == mx ? 1.0f : 0.0f) : v;
    }
    __syncthreads();
    // counts: |V| and the four "additional" candidate sets (:115-117, :143-147)
    {
        int c[5] = {0, 0, 0, 0, 0};
        for (int i = tid; i < n; i += NT) {
            const float v = nmsv[i];
            const bool valid = v > s_thr[0];
            c[0] += valid;
#pragma unroll
            for (int t = 1; t < 5; t++) c[t] += (!valid && v > s_thr[t]);
        }
#pragma unroll
        for (int t = 0; t < 5; t++) {
            int v = c[t];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if ((tid & 63) == 0) atomicAdd(&s_cnt[t], v);
        }
    }
    __syncthreads();
    const int nv = s_cnt[0];
    const float thr = s_thr[0];

    auto emit = [&](int slot, int cell, float score) {
        if (slot < K) {
            const float x = (float)(cell % G), y = (float)(cell / G);
            o_kp[2 * slot] = x;
            o_kp[2 * slot + 1] = y;
            o_sc[slot] = score;
            if (o_idx) o_idx[slot] = cell;
            if (o_px) {
                o_px[2 * slot] = x * 16.0f + 8.0f;
                o_px[2 * slot + 1] = y * 16.0f + 8.0f;
            }
        }
    };
    // walk `count` positions (thread tid owns a contiguous chunk), emit the flagged ones to slots base + rank
    auto compact = [&](int count, int base, int limit, auto flag_of, auto cell_of, auto score_of) {
        const int chunk = (count + NT - 1) / NT;
        const int b = tid * chunk, e = min(b + chunk, count);
        int local = 0;
        for (int i = b; i < e; i++) local += flag_of(i) ? 1 : 0;
        int total;
        int slot = block_excl_scan(local, sc, tid, total);
        for (int i = b; i < e; i++)
            if (flag_of(i)) {
                if (slot < limit) emit(base + slot, cell_of(i), score_of(i));
                slot++;
            }
    };
    auto sorted_cell = [&](int pos) { return (int)(0xffffffffu - (unsigned)(keys[pos] & 0xffffffffu)); };

    int st = 0, filled = K;
    if (nv >= K) {
        // :120-128  top-K of the survivors above the threshold
        compact(n, 0, K, [&](int p) { return nmsv[sorted_cell(p)] > thr; }, sorted_cell,
                [&](int p) { return key_val(keys[p]); });
    } else if (nv > 0) {
        // :130-134  all survivors first, in row-major order
        compact(n, 0, K, [&](int i) { return nmsv[i] > thr; }, [&](int i) { return i; }, [&](int i) { return nmsv[i]; });
        const int remaining = K - nv;
        int tsel = 0;
#pragma unroll
        for (int t = 4; t >= 1; t--)
            if (s_cnt[t] >= remaining) tsel = t;  // first percentile (0.40, 0.30, ..) that yields enough (:139-156)
        if (tsel) {
            const float lower = s_thr[tsel];
            compact(n, nv, remaining,
                    [&](int p) { const float v = nmsv[sorted_cell(p)]; return v > lower && !(v > thr); }, sorted_cell,
                    [&](int p) { return key_val(keys[p]); });
        } else {
            // :157-173  pad with the highest raw saliencies (survivors re-appear here: SURVEY H2)
            if (remaining > n) st = 1;
            const int m = min(remaining, n);
            for (int j = tid; j < m; j += NT) emit(nv + j, sorted_cell(j), key_val(keys[j]));
            filled = nv + m;
        }
    } else {
        // :174-184  nothing above the threshold: top-K of the raw saliency
        if (K > n) st = 1;
        const int m = min(K, n);
        for (int j = tid; j < m; j += NT) emit(j, sorted_cell(j), key_val(keys[j]));
        filled = m;
    }
    // :186-199  fewer than K points (only when torch.topk would have raised, status = 1): repeat the best one.  The first
    // slot holding the maximum score is the lowest-index cell of the global maximum in both branches (it is an NMS
    // survivor above the threshold whenever any cell is), i.e. position 0 of the canonical order.
    if (st)
        for (int j = filled + tid; j < K; j += NT) emit(j, sorted_cell(0), key_val(keys[0]));
    if (tid == 0) status[f] = st;
}

}  // namespace

extern "C" int sslam_select_keypoints(const float *sal, int n_frames, int G, int K, int nms_radius,
                                      double min_score_percentile, float *kp_xy, float *scores, int32_t *idx,
                                      float *kp_pixel, int32_t *status, void *stream) {
    if (!sal || !kp_xy || !scores || !status || n_frames <= 0 || G <= 0 || K <= 0) return SSLAM_E_INVALID;
    const int n = G * G;
    if (n > MAXN || K > MAXN || nms_radius < 0 || nms_radius > 8) return SSLAM_E_UNSUPPORTED;
    int P = 1;
    while (P < n) P <<= 1;
    const size_t shmem = (size_t)P * 8 + (size_t)n * 8;
    hipLaunchKernelGGL(select_keypoints_kernel, dim3(n_frames), dim3(NT), shmem, (hipStream_t)stream, sal, G, K,
                       nms_radius, (float)min_score_percentile, kp_xy, scores, idx, kp_pixel, status);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

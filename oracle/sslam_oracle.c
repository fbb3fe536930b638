/*
 * sslam_oracle.c - CPU restatement of the reference's extraction + matching path (see sslam_oracle.h).
 * TEST INFRASTRUCTURE ONLY; never linked into or called from the product path.
 *
 * Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fno-math-errno -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: a*b+c written as two operations must stay two roundings; every fused operation
 * is written as fmaf() explicitly.
 */
#include "sslam_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define ORA_AVX2 1
#endif

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ora_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------------
 * canonical exp / sigmoid (torch.sigmoid at keypoint_selector.py:62 is 1/(1+exp(-x)); the exp here is a
 * Cody-Waite reduction + degree-6 polynomial evaluated with fmaf, <= ~1 ulp, identical on host and device)
 * ---------------------------------------------------------------------------------------------------- */
float ora_expf(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float e = fmaf(p, r2, r) + 1.0f;
    int32_t ni = (int32_t)n; /* in [-126, 127] after the clamp */
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(ni + 127) << 23;
    return e * s.f;
}

float ora_sigmoid(float x) { return 1.0f / (1.0f + ora_expf(-x)); }

/* butterfly sum over a power-of-two array (xor offsets n/2, n/4, .. 1); every slot ends with the total.
 * This is exactly what a chain of __shfl_xor additions produces on the device. */
static float butterfly(float *a, int n) {
    float t[64];
    for (int off = n >> 1; off >= 1; off >>= 1) {
        for (int i = 0; i < n; i++) t[i] = a[i] + a[i ^ off];
        memcpy(a, t, (size_t)n * sizeof(float));
    }
    return a[0];
}

/* ------------------------------------------------------------------------------------------------------
 * chained-FMA GEMM micro-kernel:  C[i][j] = fmaf-chain over k of A_i[k] * Bt[k][j], starting from bias[j]
 * (or 0).  A rows are given as `nseg` segment pointers of `seglen` floats each (a NULL segment is a row of
 * zeros, still multiplied through the chain, as the zero-padded conv taps are on the device).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    const float *const *seg; /* [rows][nseg] */
    int nseg, seglen;
} arows_t;

static const float ZEROS[ORA_C] = {0};

/* Blocking: K is walked one segment at a time (the accumulators round-trip through C between segments, which
 * continues the SAME chain: a stored fp32 is reloaded unchanged), B is re-packed into contiguous column panels. */
static void micro_4xN(const float *const a[4], int klen, const float *bp, int w, float *c, int ldc, int mr,
                      const float *init /* bias slice or NULL */, int first) {
#ifdef ORA_AVX2
#define ORA_TILE(NV)                                                                                           \
    {                                                                                                          \
        __m256 acc[4][NV];                                                                                     \
        for (int i = 0; i < 4; i++)                                                                            \
            for (int v = 0; v < NV; v++)                                                                       \
                acc[i][v] = first ? (init ? _mm256_loadu_ps(init + 8 * v) : _mm256_setzero_ps())               \
                                  : _mm256_loadu_ps(c + (size_t)(i < mr ? i : 0) * ldc + 8 * v);               \
        const float *b = bp;                                                                                   \
        for (int k = 0; k < klen; k++, b += 8 * NV) {                                                          \
            __m256 bv[NV];                                                                                     \
            for (int v = 0; v < NV; v++) bv[v] = _mm256_loadu_ps(b + 8 * v);                                   \
            const __m256 x0 = _mm256_broadcast_ss(a[0] + k), x1 = _mm256_broadcast_ss(a[1] + k);               \
            const __m256 x2 = _mm256_broadcast_ss(a[2] + k), x3 = _mm256_broadcast_ss(a[3] + k);               \
            for (int v = 0; v < NV; v++) {                                                                     \
                acc[0][v] = _mm256_fmadd_ps(x0, bv[v], acc[0][v]);                                             \
                acc[1][v] = _mm256_fmadd_ps(x1, bv[v], acc[1][v]);                                             \
                acc[2][v] = _mm256_fmadd_ps(x2, bv[v], acc[2][v]);                                             \
                acc[3][v] = _mm256_fmadd_ps(x3, bv[v], acc[3][v]);                                             \
            }                                                                                                  \
        }                                                                                                      \
        for (int i = 0; i < mr; i++)                                                                           \
            for (int v = 0; v < NV; v++) _mm256_storeu_ps(c + (size_t)i * ldc + 8 * v, acc[i][v]);             \
        return;                                                                                                \
    }
    if (w == 24) ORA_TILE(3)
    if (w == 16) ORA_TILE(2)
    if (w == 8) ORA_TILE(1)
#undef ORA_TILE
#endif
    for (int i = 0; i < mr; i++)
        for (int j = 0; j < w; j++) {
            float acc = first ? (init ? init[j] : 0.0f) : c[(size_t)i * ldc + j];
            for (int k = 0; k < klen; k++) acc = fmaf(a[i][k], bp[(size_t)k * w + j], acc);
            c[(size_t)i * ldc + j] = acc;
        }
}

static void chain_gemm(const arows_t *A, int rows, const float *Bt, int N, const float *bias, float *C, int ldc) {
    const int nseg = A->nseg, seglen = A->seglen, K = nseg * seglen;
    const int KC = 384;                                   /* K block: accumulators round-trip through C in between */
    const int nblk = (K + KC - 1) / KC;
    /* column panels: widths 24,24,...,then 16 / 8 / remainder */
    int pw[512], po[512], np = 0;
    for (int j = 0; j < N;) {
        const int left = N - j;
        const int w = left >= 24 ? 24 : (left >= 16 ? 16 : (left >= 8 ? 8 : left));
        po[np] = j; pw[np] = w; np++; j += w;
    }
    /* Bp[blk][panel][k in block][w] */
    float *Bp = (float *)malloc((size_t)K * N * sizeof(float));
    size_t *boff = (size_t *)malloc((size_t)nblk * np * sizeof(size_t));
    {
        size_t off = 0;
        for (int b = 0; b < nblk; b++) {
            const int k0 = b * KC, kl = k0 + KC <= K ? KC : K - k0;
            for (int p = 0; p < np; p++) {
                boff[(size_t)b * np + p] = off;
                for (int k = 0; k < kl; k++)
                    memcpy(Bp + off + (size_t)k * pw[p], Bt + (size_t)(k0 + k) * N + po[p], (size_t)pw[p] * sizeof(float));
                off += (size_t)kl * pw[p];
            }
        }
    }
    const int MC = 32;
#pragma omp parallel
    {
        /* segmented rows (the conv's shifted / zero-padded taps) are packed into contiguous K-ordered rows per block */
        float *Ap = nseg > 1 ? (float *)malloc((size_t)MC * K * sizeof(float)) : NULL;
#pragma omp for schedule(dynamic, 1)
        for (int rb = 0; rb < rows; rb += MC) {
            const int rend = rb + MC < rows ? rb + MC : rows;
            if (Ap)
                for (int i = rb; i < rend; i++)
                    for (int sg = 0; sg < nseg; sg++) {
                        const float *q = A->seg[(size_t)i * nseg + sg];
                        memcpy(Ap + (size_t)(i - rb) * K + (size_t)sg * seglen, q ? q : ZEROS, (size_t)seglen * sizeof(float));
                    }
            for (int b = 0; b < nblk; b++) {
                const int k0 = b * KC, kl = k0 + KC <= K ? KC : K - k0;
                for (int p = 0; p < np; p++)
                    for (int i0 = rb; i0 < rend; i0 += 4) {
                        const int mr = rend - i0 < 4 ? rend - i0 : 4;
                        const float *a[4];
                        for (int i = 0; i < 4; i++) {
                            const int ri = i0 + (i < mr ? i : 0);
                            a[i] = (Ap ? Ap + (size_t)(ri - rb) * K : A->seg[ri]) + k0;
                        }
                        micro_4xN(a, kl, Bp + boff[(size_t)b * np + p], pw[p], C + (size_t)i0 * ldc + po[p], ldc, mr,
                                  bias ? bias + po[p] : NULL, b == 0);
                    }
            }
        }
        free(Ap);
    }
    free(boff);
    free(Bp);
}

/* Bt[k][n] = W[n][k] */
static float *transpose(const float *W, int N, int K) {
    float *Bt = (float *)malloc((size_t)N * K * sizeof(float));
    for (int n = 0; n < N; n++)
        for (int k = 0; k < K; k++) Bt[(size_t)k * N + n] = W[(size_t)n * K + k];
    return Bt;
}

/* ------------------------------------------------------------------------------------------------------
 * A2  BatchNorm1d over tokens (dino_backbone.py:91-106)
 * Column reduction order (R = group*cells rows of one statistics group, channel c):
 *   16 partial sums P[p], p = r mod 16, each a sequential sum in increasing r;
 *   S_w = (P[4w] + P[4w+1]) + (P[4w+2] + P[4w+3]) for w = 0..3;  total = ((S_0 + S_1) + S_2) + S_3.
 *   mean = total / R;  second pass on d = x - mean with P[p] = fmaf(d, d, P[p]);  var = total2 / R (biased).
 *   invstd = 1/sqrtf(var + eps); alpha = invstd*gamma; beta' = beta - mean*alpha;  y = x*alpha + beta'.
 * ---------------------------------------------------------------------------------------------------- */
static float comb16(const float *P) {
    float S[4];
    for (int w = 0; w < 4; w++) S[w] = (P[4 * w] + P[4 * w + 1]) + (P[4 * w + 2] + P[4 * w + 3]);
    return ((S[0] + S[1]) + S[2]) + S[3];
}

void ora_bn_tokens(const float *tokens, int n_frames, int tokens_per_frame, int n_prefix, int group,
                   const float *gamma, const float *beta, const float *run_mean, const float *run_var,
                   int train, float eps, float *out_feat, float *out_mean, float *out_var) {
    const int cells = tokens_per_frame - n_prefix;
    const int n_groups = n_frames / group;
    const int R = group * cells;
#pragma omp parallel for schedule(static)
    for (int g = 0; g < n_groups; g++) {
        float mean[ORA_C], var[ORA_C];
        if (train) {
            static const int dummy = 0;
            (void)dummy;
            float (*P)[ORA_C] = (float (*)[ORA_C])malloc(16 * sizeof(*P));
            memset(P, 0, 16 * sizeof(*P));
            for (int r = 0; r < R; r++) {
                const int f = g * group + r / cells, t = n_prefix + r % cells;
                const float *__restrict x = tokens + ((size_t)f * tokens_per_frame + t) * ORA_C;
                float *__restrict p = P[r & 15];
                for (int c = 0; c < ORA_C; c++) p[c] = p[c] + x[c];
            }
            for (int c = 0; c < ORA_C; c++) {
                float q[16];
                for (int p = 0; p < 16; p++) q[p] = P[p][c];
                mean[c] = comb16(q) / (float)R;
            }
            memset(P, 0, 16 * sizeof(*P));
            for (int r = 0; r < R; r++) {
                const int f = g * group + r / cells, t = n_prefix + r % cells;
                const float *__restrict x = tokens + ((size_t)f * tokens_per_frame + t) * ORA_C;
                float *__restrict p = P[r & 15];
                for (int c = 0; c < ORA_C; c++) {
                    const float d = x[c] - mean[c];
                    p[c] = fmaf(d, d, p[c]);
                }
            }
            for (int c = 0; c < ORA_C; c++) {
                float q[16];
                for (int p = 0; p < 16; p++) q[p] = P[p][c];
                var[c] = comb16(q) / (float)R;
            }
            free(P);
            if (out_mean) memcpy(out_mean + (size_t)g * ORA_C, mean, sizeof(mean));
            if (out_var) memcpy(out_var + (size_t)g * ORA_C, var, sizeof(var));
        } else {
            memcpy(mean, run_mean, sizeof(mean));
            memcpy(var, run_var, sizeof(var));
        }
        float alpha[ORA_C], bshift[ORA_C];
        for (int c = 0; c < ORA_C; c++) {
            const float invstd = 1.0f / sqrtf(var[c] + eps);
            alpha[c] = invstd * gamma[c];
            bshift[c] = beta[c] - mean[c] * alpha[c];
        }
        for (int r = 0; r < R; r++) {
            const int f = g * group + r / cells, t = r % cells;
            const float *__restrict x = tokens + ((size_t)f * tokens_per_frame + n_prefix + t) * ORA_C;
            float *__restrict y = out_feat + ((size_t)f * cells + t) * ORA_C;
            for (int c = 0; c < ORA_C; c++) y[c] = x[c] * alpha[c] + bshift[c];
        }
    }
}

/* ------------------------------------------------------------------------------------------------------
 * A3  saliency CNN (keypoint_selector.py:45-67)
 *   hidden[n] = fmaf chain from b1[n] over k = (chunk*9 + tap)*32 + cc  (channel c = chunk*32 + cc, tap = ky*3 + kx;
 *   the nine taps of a 32-channel slice are consecutive; zero taps included), then ReLU.
 *   logit: p[n] = hidden[n] * w2[n];  for each 64-wide slab s: q[c] = p[64s + c] + p[64s + 32 + c] (c < 32),
 *   T_s = butterfly32(q);  logit = ((b2 + T_0) + T_1) + ...;  saliency = 1 / (1 + ora_expf(-logit)).
 * ---------------------------------------------------------------------------------------------------- */
void ora_selector_saliency(const float *feat, int n_frames, int G, const float *w1, const float *b1,
                           const float *w2, const float *b2, int hs, float *sal) {
    const int cells = G * G, K = 9 * ORA_C, NSEG = 108, SEGLEN = 32;
    /* canonical k order: k = (chunk*9 + tap)*32 + cc with channel c = chunk*32 + cc, tap = ky*3 + kx */
    float *Bt = (float *)malloc((size_t)K * hs * sizeof(float));
    for (int n = 0; n < hs; n++)
        for (int c = 0; c < ORA_C; c++)
            for (int t = 0; t < 9; t++)
                Bt[((size_t)((c / SEGLEN) * 9 + t) * SEGLEN + c % SEGLEN) * hs + n] = w1[((size_t)n * ORA_C + c) * 9 + t];
    const int rows = n_frames * cells;
    const int RB = 4096;   /* row blocks bound the hidden buffer and the segment table */
    const float **seg = (const float **)malloc((size_t)RB * NSEG * sizeof(float *));
    float *hid = (float *)malloc((size_t)RB * hs * sizeof(float));
    for (int r0 = 0; r0 < rows; r0 += RB) {
        const int nr = rows - r0 < RB ? rows - r0 : RB;
        for (int i = 0; i < nr; i++) {
            const int m = r0 + i, f = m / cells, cell = m % cells, y = cell / G, x = cell % G;
            for (int t = 0; t < 9; t++) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const float *row = (yy < 0 || yy >= G || xx < 0 || xx >= G) ? NULL : feat + (((size_t)f * G + yy) * G + xx) * ORA_C;
                for (int ch = 0; ch < ORA_C / SEGLEN; ch++) seg[(size_t)i * NSEG + ch * 9 + t] = row ? row + ch * SEGLEN : NULL;
            }
        }
        arows_t Ab = {seg, NSEG, SEGLEN};
        chain_gemm(&Ab, nr, Bt, hs, b1, hid, hs);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < nr; i++) {
            const float *h = hid + (size_t)i * hs;
            float logit = b2[0];
            for (int s = 0; s < hs / 64; s++) {
                float q[32];
                for (int c = 0; c < 32; c++) {
                    const float h0 = h[64 * s + c] > 0.0f ? h[64 * s + c] : 0.0f;
                    const float h1 = h[64 * s + 32 + c] > 0.0f ? h[64 * s + 32 + c] : 0.0f;
                    q[c] = h0 * w2[64 * s + c] + h1 * w2[64 * s + 32 + c];
                }
                logit = logit + butterfly(q, 32);
            }
            sal[r0 + i] = ora_sigmoid(logit);
        }
    }
    free(hid);
    free(seg);
    free(Bt);
}

/* ------------------------------------------------------------------------------------------------------
 * A4  NMS (keypoint_selector.py:209-226): max_pool2d(k = 2r+1, stride 1, pad r with -inf), keep sal==pooled
 * ---------------------------------------------------------------------------------------------------- */
void ora_nms(const float *sal, int G, int radius, float *out) {
    if (radius == 0) {
        memcpy(out, sal, (size_t)G * G * sizeof(float));
        return;
    }
    for (int y = 0; y < G; y++)
        for (int x = 0; x < G; x++) {
            float mx = -INFINITY;
            for (int yy = y - radius; yy <= y + radius; yy++)
                for (int xx = x - radius; xx <= x + radius; xx++)
                    if (yy >= 0 && yy < G && xx >= 0 && xx < G && sal[yy * G + xx] > mx) mx = sal[yy * G + xx];
            const float v = sal[y * G + x];
            out[y * G + x] = v * (v == mx ? 1.0f : 0.0f);
        }
}

/* value descending, index ascending (SURVEY H3 canonical top-k order) */
typedef struct { float v; int32_t i; } vi_t;
static int cmp_desc(const void *a, const void *b) {
    const vi_t *x = (const vi_t *)a, *y = (const vi_t *)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* torch.quantile(v, q, interpolation='linear') for a 1-D fp32 tensor:
 *   rank = q(as fp32) * (n-1) in fp32; lo = floor, hi = ceil, w = rank - lo;  result = lerp(v[lo], v[hi], w)
 *   with ATen's lerp evaluated fused (matches torch 2.10 CPU bit-for-bit on the quantile fixtures):
 *   w < 0.5 ? fmaf(w, b-a, a) : fmaf(-(b-a), 1-w, b). */
static float quantile_sorted_desc(const vi_t *desc, int n, double q) {
    const float rank = (float)q * (float)(n - 1);
    const float lo = floorf(rank), hi = ceilf(rank);
    const float w = rank - lo;
    const float a = desc[n - 1 - (int)lo].v, b = desc[n - 1 - (int)hi].v;
    const float diff = b - a;
    return (w < 0.5f) ? fmaf(w, diff, a) : fmaf(-diff, 1.0f - w, b);
}

float ora_quantile(const float *v, int n, double q) {
    vi_t *d = (vi_t *)malloc((size_t)n * sizeof(vi_t));
    for (int i = 0; i < n; i++) { d[i].v = v[i]; d[i].i = i; }
    qsort(d, (size_t)n, sizeof(vi_t), cmp_desc);
    const float r = quantile_sorted_desc(d, n, q);
    free(d);
    return r;
}

/* ------------------------------------------------------------------------------------------------------
 * A5  select_keypoints (keypoint_selector.py:69-207), one frame
 * ---------------------------------------------------------------------------------------------------- */
static int select_one(const float *sal, int G, int K, int radius, double pct, float *kp, float *sc, int32_t *idx) {
    const int n = G * G;
    vi_t *desc = (vi_t *)malloc((size_t)n * sizeof(vi_t));    /* raw saliency, canonical descending order */
    float *nms = (float *)malloc((size_t)n * sizeof(float));
    vi_t *cand = (vi_t *)malloc((size_t)n * sizeof(vi_t));
    unsigned char *valid = (unsigned char *)malloc((size_t)n);
    int status = 0, cnt = 0;
    for (int i = 0; i < n; i++) { desc[i].v = sal[i]; desc[i].i = i; }
    qsort(desc, (size_t)n, sizeof(vi_t), cmp_desc);
    /* :105-109  threshold = max(quantile(sal, pct).item(), 0.1), compared in fp32 */
    const float thr = fmaxf(quantile_sorted_desc(desc, n, pct), 0.1f);
    ora_nms(sal, G, radius, nms);                                                   /* :112 */
    int nv = 0;
    for (int i = 0; i < n; i++) { valid[i] = nms[i] > thr; nv += valid[i]; }          /* :115-117 */
#define EMIT(cell, score) do { if (cnt < K) { idx[cnt] = (cell); kp[2 * cnt] = (float)((cell) % G); \
        kp[2 * cnt + 1] = (float)((cell) / G); sc[cnt] = (score); cnt++; } } while (0)
    if (nv >= K) {                                                                  /* :120-128 top-K of V */
        int m = 0;
        for (int i = 0; i < n; i++) if (valid[i]) { cand[m].v = nms[i]; cand[m].i = i; m++; }
        qsort(cand, (size_t)m, sizeof(vi_t), cmp_desc);
        for (int j = 0; j < K; j++) EMIT(cand[j].i, cand[j].v);
    } else if (nv > 0) {                                                            /* :130-173 */
        for (int i = 0; i < n; i++) if (valid[i]) EMIT(i, nms[i]);                    /* row-major order */
        const int remaining = K - nv;
        static const double pcts[4] = {0.40, 0.30, 0.20, 0.10};
        int done = 0;
        for (int t = 0; t < 4 && !done; t++) {                                      /* :139-156 */
            const float lower = fmaxf(quantile_sorted_desc(desc, n, pcts[t]), 0.05f);
            int m = 0;
            for (int i = 0; i < n; i++) if (nms[i] > lower && !valid[i]) { cand[m].v = nms[i]; cand[m].i = i; m++; }
            if (m >= remaining) {
                qsort(cand, (size_t)m, sizeof(vi_t), cmp_desc);
                for (int j = 0; j < remaining; j++) EMIT(cand[j].i, cand[j].v);
                done = 1;
            }
        }
        if (!done) {                                                                /* :157-173 pad with top raw */
            if (remaining > n) status = 1;                                          /* torch.topk would raise */
            for (int j = 0; j < remaining && j < n; j++) EMIT(desc[j].i, desc[j].v);
        }
    } else {                                                                        /* :174-184 */
        if (K > n) status = 1;
        for (int j = 0; j < K && j < n; j++) EMIT(desc[j].i, desc[j].v);
    }
    if (cnt < K && cnt > 0) {                                                       /* :190-199 (unreachable unless status) */
        int best = 0;
        for (int j = 1; j < cnt; j++) if (sc[j] > sc[best]) best = j;
        const int bi = idx[best];
        const float bs = sc[best];
        while (cnt < K) EMIT(bi, bs);
    }
#undef EMIT
    free(valid); free(cand); free(nms); free(desc);
    return status;
}

void ora_select_keypoints(const float *sal, int n_frames, int G, int K, int radius, double pct, float *kp_xy,
                          float *scores, int32_t *idx, int32_t *status) {
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < n_frames; f++) {
        const int st = select_one(sal + (size_t)f * G * G, G, K, radius, pct, kp_xy + (size_t)f * K * 2,
                                  scores + (size_t)f * K, idx + (size_t)f * K);
        if (status) status[f] = st;
    }
}

/* ------------------------------------------------------------------------------------------------------
 * A6  bilinear feature gather (dino_backbone.py:131-152); grid_sample(bilinear, align_corners=True, zeros)
 *   xn = 2*x/(W-1) - 1;  ix = (xn + 1) * ((W-1)/2);  x0 = floor(ix); w = ix - x0; e = 1 - w  (same for y: n, s)
 *   weights nw = s*e, ne = s*w, sw = n*e, se = n*w;  out = ((v_nw*nw + v_ne*ne) + v_sw*sw) + v_se*se,
 *   out-of-range taps read 0.
 * ---------------------------------------------------------------------------------------------------- */
void ora_gather(const float *feat, int n_frames, int G, const float *kp_xy, int K, float *out) {
    const float gm1 = (float)(G - 1), half = gm1 / 2.0f;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n_frames * K; r++) {
        const int f = r / K;
        const float x = kp_xy[2 * (size_t)r], y = kp_xy[2 * (size_t)r + 1];
        const float xn = 2.0f * x / gm1 - 1.0f, yn = 2.0f * y / gm1 - 1.0f;
        const float ix = (xn + 1.0f) * half, iy = (yn + 1.0f) * half;
        const float x0 = floorf(ix), y0 = floorf(iy);
        const float w = ix - x0, e = 1.0f - w, n = iy - y0, s = 1.0f - n;
        const float wt[4] = {s * e, s * w, n * e, n * w};
        const int xs[4] = {(int)x0, (int)x0 + 1, (int)x0, (int)x0 + 1};
        const int ys[4] = {(int)y0, (int)y0, (int)y0 + 1, (int)y0 + 1};
        const float *src[4];
        for (int t = 0; t < 4; t++)
            src[t] = (xs[t] < 0 || xs[t] >= G || ys[t] < 0 || ys[t] >= G)
                         ? ZEROS : feat + (((size_t)f * G + ys[t]) * G + xs[t]) * ORA_C;
        float *o = out + (size_t)r * ORA_C;
        for (int c = 0; c < ORA_C; c++)
            o[c] = ((src[0][c] * wt[0] + src[1][c] * wt[1]) + src[2][c] * wt[2]) + src[3][c] * wt[3];
    }
}

/* ------------------------------------------------------------------------------------------------------
 * A7  descriptor MLP (descriptor_refiner.py:58-126)
 *   Row reductions follow the accumulator layout of the GPU kernel, which evaluates the products TRANSPOSED (a lane owns one
 *   activation row; 4 waves x 96 columns; per wave two half-waves h, each holding columns slab + 32t + crow(e,h),
 *   crow(e,h) = (e&3) + 8(e>>2) + 4h, t = 0..2, e = 0..15):
 *     P[w][h] = sequential sum over (t, e) from 0;  W[w] = P[w][0] + P[w][1];  total = ((W0 + W1) + W2) + W3.
 *   LayerNorm over 384: mean = total(x) / 384; variance likewise on fmaf(d, d, .) with d = x - mean;
 *   rstd = 1/sqrtf(var + eps); y = fmaf((x - mean) * rstd, gamma, beta).
 *   L2 normalise over 128 (F.normalize, eps 1e-12): 4 waves x 32 columns (t = 0 only), fmaf(v, v, .) chains, same tree;
 *   out = v / max(sqrtf(total), 1e-12).
 * ---------------------------------------------------------------------------------------------------- */
static inline int crow_(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

/* ((W0 + W1) + W2) + W3 over `tiles` 32-column tiles per wave; mode 0: sum x, 1: sum (x - mean)^2, 2: sum x^2 */
static float slab_total(const float *x, int tiles, float mean, int mode) {
    float W[4];
    for (int w = 0; w < 4; w++) {
        float P[2];
        for (int h = 0; h < 2; h++) {
            float s = 0.0f;
            for (int t = 0; t < tiles; t++)
                for (int e = 0; e < 16; e++) {
                    const float v = x[w * 32 * tiles + t * 32 + crow_(e, h)];
                    if (mode == 0) s = s + v;
                    else if (mode == 1) { const float d = v - mean; s = fmaf(d, d, s); }
                    else s = fmaf(v, v, s);
                }
            P[h] = s;
        }
        W[w] = P[0] + P[1];
    }
    return ((W[0] + W[1]) + W[2]) + W[3];
}

static void layernorm384(const float *x, const float *g, const float *b, float *y) {
    const float mean = slab_total(x, 3, 0.0f, 0) / 384.0f;
    const float var = slab_total(x, 3, mean, 1) / 384.0f;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    for (int c = 0; c < 384; c++) y[c] = fmaf((x[c] - mean) * rstd, g[c], b[c]);
}

void ora_refine(const float *x, int rows, const float *const *w, int n_blocks, int d_out, float *desc) {
    const int H = 384;
    float *cur = (float *)malloc((size_t)rows * H * sizeof(float));
    float *t1 = (float *)malloc((size_t)rows * H * sizeof(float));
    float *t2 = (float *)malloc((size_t)rows * H * sizeof(float));
    const float **seg = (const float **)malloc((size_t)rows * sizeof(float *));
    arows_t A = {seg, 1, H};
    float *Bt;
    /* input_proj + ReLU (:76) */
    for (int r = 0; r < rows; r++) seg[r] = x + (size_t)r * ORA_C;
    Bt = transpose(w[0], H, ORA_C);
    chain_gemm(&A, rows, Bt, H, w[1], cur, H);
    free(Bt);
    for (size_t i = 0; i < (size_t)rows * H; i++) cur[i] = cur[i] > 0.0f ? cur[i] : 0.0f;
    for (int blk = 0; blk < n_blocks; blk++) {                                       /* :108-126 */
        const float *const *p = w + 2 + 8 * blk;
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rows; r++) layernorm384(cur + (size_t)r * H, p[0], p[1], t1 + (size_t)r * H);
        for (int r = 0; r < rows; r++) seg[r] = t1 + (size_t)r * H;
        Bt = transpose(p[2], H, H);
        chain_gemm(&A, rows, Bt, H, p[3], t2, H);
        free(Bt);
        for (size_t i = 0; i < (size_t)rows * H; i++) t2[i] = t2[i] > 0.0f ? t2[i] : 0.0f;
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rows; r++) layernorm384(t2 + (size_t)r * H, p[4], p[5], t1 + (size_t)r * H);
        Bt = transpose(p[6], H, H);
        chain_gemm(&A, rows, Bt, H, p[7], t2, H);
        free(Bt);
        for (size_t i = 0; i < (size_t)rows * H; i++) {
            const float v = t2[i] + cur[i];
            cur[i] = v > 0.0f ? v : 0.0f;
        }
    }
    const float *const *po = w + 2 + 8 * n_blocks;
    for (int r = 0; r < rows; r++) seg[r] = cur + (size_t)r * H;
    Bt = transpose(po[0], d_out, H);
    chain_gemm(&A, rows, Bt, d_out, po[1], desc, d_out);
    free(Bt);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        float *v = desc + (size_t)r * d_out;
        const float ss = slab_total(v, d_out / 128, 0.0f, 2);
        const float den = fmaxf(sqrtf(ss), 1e-12f);
        for (int c = 0; c < d_out; c++) v[c] = v[c] / den;
    }
    free(seg); free(t2); free(t1); free(cur);
}

void ora_patch_to_pixel(const float *kp, int n, float *out) {     /* dino_backbone.py:164 */
    for (int i = 0; i < n; i++) out[i] = kp[i] * 16.0f + 8.0f;
}
void ora_pixel_to_patch(const float *px, int n, float *out) {     /* dino_backbone.py:177 */
    for (int i = 0; i < n; i++) out[i] = (px[i] - 8.0f) / 16.0f;
}

/* ------------------------------------------------------------------------------------------------------
 * similarity + arg-max (visualize_matches_sequence.py:144-148 and the four sibling matchers)
 *   S[i][j] = fmaf chain from 0 over k = 0..d-1 of d1[i][k]*d2[j][k]; arg-max = first maximal index.
 * ---------------------------------------------------------------------------------------------------- */
void ora_sim_argmax(const float *d1, int n, const float *d2, int m, int d, int32_t *nn12, float *s12,
                    int32_t *nn21, float *s21) {
    float *Bt = transpose(d2, m, d);
    const int RB = 64;
    const int nblk = (n + RB - 1) / RB;
    float *cpart = (float *)malloc((size_t)nblk * m * sizeof(float));
    int32_t *ipart = (int32_t *)malloc((size_t)nblk * m * sizeof(int32_t));
    const float **seg = (const float **)malloc((size_t)n * sizeof(float *));
    for (int i = 0; i < n; i++) seg[i] = d1 + (size_t)i * d;
#pragma omp parallel
    {
        float *S = (float *)malloc((size_t)RB * m * sizeof(float));
#pragma omp for schedule(static)
        for (int b = 0; b < nblk; b++) {
            const int r0 = b * RB, nr = n - r0 < RB ? n - r0 : RB;
            arows_t A = {seg + r0, 1, d};
            /* chain_gemm has its own parallel-for; nested regions run serially inside this one */
            chain_gemm(&A, nr, Bt, m, NULL, S, m);
            for (int j = 0; j < m; j++) { cpart[(size_t)b * m + j] = -INFINITY; ipart[(size_t)b * m + j] = r0; }
            for (int i = 0; i < nr; i++) {
                const float *row = S + (size_t)i * m;
                float best = row[0];
                int bj = 0;
                for (int j = 1; j < m; j++) if (row[j] > best) { best = row[j]; bj = j; }
                nn12[r0 + i] = bj;
                s12[r0 + i] = best;
                for (int j = 0; j < m; j++)
                    if (row[j] > cpart[(size_t)b * m + j]) { cpart[(size_t)b * m + j] = row[j]; ipart[(size_t)b * m + j] = r0 + i; }
            }
        }
        free(S);
    }
    for (int j = 0; j < m; j++) {
        float best = cpart[j];
        int bi = ipart[j];
        for (int b = 1; b < nblk; b++)
            if (cpart[(size_t)b * m + j] > best) { best = cpart[(size_t)b * m + j]; bi = ipart[(size_t)b * m + j]; }
        nn21[j] = bi;
        if (s21) s21[j] = best;
    }
    free(seg); free(ipart); free(cpart); free(Bt);
}

/* the full similarity matrix in the canonical order (tests of M2 / M4 derive second-best values from it) */
void ora_sim_matrix(const float *d1, int n, const float *d2, int m, int d, float *S) {
    float *Bt = transpose(d2, m, d);
    const float **seg = (const float **)malloc((size_t)n * sizeof(float *));
    for (int i = 0; i < n; i++) seg[i] = d1 + (size_t)i * d;
    arows_t A = {seg, 1, d};
    chain_gemm(&A, n, Bt, m, NULL, S, m);
    free(seg);
    free(Bt);
}

/* M1 (visualize_matches_sequence.py:106-197).  python-float thresholds meet fp32 tensors, so every comparison
 * and product is done in fp32 with the scalar rounded to fp32 first. */
int ora_match_with_quality(const float *d1, int n, const float *d2, int m, int d, const float *sc1,
                           const float *sc2, double saliency_weight, double min_saliency, double min_sim,
                           const float *int1, const float *int2, double min_intensity, int64_t *matches,
                           float *quality) {
    int32_t *nn12 = (int32_t *)malloc((size_t)n * sizeof(int32_t)), *nn21 = (int32_t *)malloc((size_t)m * sizeof(int32_t));
    float *s12 = (float *)malloc((size_t)n * sizeof(float));
    ora_sim_argmax(d1, n, d2, m, d, nn12, s12, nn21, NULL);
    const float w_desc = (float)(1.0 - saliency_weight), w_sal = (float)saliency_weight;
    const float t_sal = (float)min_saliency, t_sim = (float)min_sim, t_int = (float)min_intensity;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        const int j = nn12[i];
        if (nn21[j] != i) continue;                                      /* :149 mutual */
        const float sim = s12[i];
        const float avg_sal = (sc1[i] + sc2[j]) / 2.0f;                   /* :163 */
        int ok = (avg_sal >= t_sal) && (sim >= t_sim);                    /* :166-168 */
        if (int1 && int2) ok = ok && ((int1[i] + int2[j]) / 2.0f >= t_int); /* :171-176 */
        if (!ok) continue;
        matches[2 * cnt] = i;
        matches[2 * cnt + 1] = j;
        quality[cnt] = w_desc * sim + w_sal * avg_sal;                    /* :189-192 */
        cnt++;
    }
    free(s12); free(nn21); free(nn12);
    return cnt;
}

/* ------------------------------------------------------------------------------------------------------
 * A0 / A9  Pillow resampling (third-party: Pillow, unpinned in requirements.txt:5; pinned here by goldens made
 * with Pillow 12.2.0).  Restates the published algorithm of libImaging/Resample.c: per-axis coefficient
 * tables in double, converted to 22-bit fixed point; horizontal pass then vertical pass, each rounding to
 * uint8 ((acc + 2^21) >> 22, clamped).
 * ---------------------------------------------------------------------------------------------------- */
#define PREC_BITS 22

static double filt_bilinear(double x) { if (x < 0.0) x = -x; return x < 1.0 ? 1.0 - x : 0.0; }
static double filt_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

/* returns ksize; bounds[2*xx] = first input index, bounds[2*xx+1] = tap count; kk[xx*ksize + t] fixed point */
static int precompute_coeffs(int in_size, int out_size, int filter, int **bounds_out, int32_t **kk_out) {
    double (*fn)(double) = filter ? filt_bicubic : filt_bilinear;
    const double fsupport = filter ? 2.0 : 1.0;
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = fsupport * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    int *bounds = (int *)malloc((size_t)out_size * 2 * sizeof(int));
    int32_t *kk = (int32_t *)malloc((size_t)out_size * ksize * sizeof(int32_t));
    double *k = (double *)malloc((size_t)ksize * sizeof(double));
    for (int xx = 0; xx < out_size; xx++) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; x++) {
            const double w = fn((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        for (x = 0; x < ksize; x++)
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int32_t)(-0.5 + k[x] * (1 << PREC_BITS)) : (int32_t)(0.5 + k[x] * (1 << PREC_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(k);
    *bounds_out = bounds;
    *kk_out = kk;
    return ksize;
}

static inline uint8_t clip8(int32_t v) {
    v >>= PREC_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* full two-pass resize of an RGB uint8 image to (size, size) */
static void resize_rgb_u8(const uint8_t *img, int h, int w, int size, int filter, uint8_t *out) {
    int *bh, *bv;
    int32_t *kh, *kv;
    const int ksh = precompute_coeffs(w, size, filter, &bh, &kh);
    const int ksv = precompute_coeffs(h, size, filter, &bv, &kv);
    const uint8_t *src = img;
    uint8_t *tmp = NULL;
    if (w != size) { /* horizontal pass: (h, size, 3) */
        tmp = (uint8_t *)malloc((size_t)h * size * 3);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; y++)
            for (int xx = 0; xx < size; xx++) {
                const int xmin = bh[2 * xx], xn = bh[2 * xx + 1];
                const int32_t *k = kh + (size_t)xx * ksh;
                int32_t s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
                for (int x = 0; x < xn; x++) {
                    const uint8_t *p = img + ((size_t)y * w + xmin + x) * 3;
                    s0 += p[0] * k[x]; s1 += p[1] * k[x]; s2 += p[2] * k[x];
                }
                uint8_t *o = tmp + ((size_t)y * size + xx) * 3;
                o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
            }
        src = tmp;
    }
    if (h != size) {
#pragma omp parallel for schedule(static)
        for (int yy = 0; yy < size; yy++) {
            const int ymin = bv[2 * yy], yn = bv[2 * yy + 1];
            const int32_t *k = kv + (size_t)yy * ksv;
            for (int x = 0; x < size * 3; x++) {
                int32_t s = 1 << (PREC_BITS - 1);
                for (int y = 0; y < yn; y++) s += src[((size_t)(ymin + y) * size) * 3 + x] * k[y];
                out[(size_t)yy * size * 3 + x] = clip8(s);
            }
        }
    } else {
        memcpy(out, src, (size_t)size * size * 3);
    }
    free(tmp); free(kv); free(bv); free(kh); free(bh);
}

void ora_resize_rgb(const uint8_t *img, int h, int w, int size, int filter, uint8_t *resized, float *chw) {
    static const float MEAN[3] = {0.485f, 0.456f, 0.406f}, STD[3] = {0.229f, 0.224f, 0.225f};
    uint8_t *rs = resized ? resized : (uint8_t *)malloc((size_t)size * size * 3);
    resize_rgb_u8(img, h, w, size, filter, rs);
    if (chw) {
        const size_t plane = (size_t)size * size;
        for (int c = 0; c < 3; c++)
            for (size_t i = 0; i < plane; i++)
                chw[c * plane + i] = ((float)rs[i * 3 + c] / 255.0f - MEAN[c]) / STD[c];   /* ToTensor, Normalize */
    }
    if (!resized) free(rs);
}

static inline uint8_t luma(const uint8_t *p) {                       /* Pillow "L": ITU-R 601-2, 16-bit fixed */
    return (uint8_t)((p[0] * 19595 + p[1] * 38470 + p[2] * 7471 + 0x8000) >> 16);
}

void ora_gray_resized(const uint8_t *img, int h, int w, int size, uint8_t *gray) {
    uint8_t *rs = (uint8_t *)malloc((size_t)size * size * 3);
    resize_rgb_u8(img, h, w, size, 1, rs);
    for (size_t i = 0; i < (size_t)size * size; i++) gray[i] = luma(rs + 3 * i);
    free(rs);
}

void ora_intensity(const uint8_t *img, int h, int w, int size, const float *kp_pixel, int K, float *out) {
    uint8_t *gray = (uint8_t *)malloc((size_t)size * size);
    ora_gray_resized(img, h, w, size, gray);
    for (int i = 0; i < K; i++) {
        /* numpy round = half-to-even (rintf under the default rounding mode), then clip (:93-94) */
        int x = (int)rintf(kp_pixel[2 * i]), y = (int)rintf(kp_pixel[2 * i + 1]);
        x = x < 0 ? 0 : (x > size - 1 ? size - 1 : x);
        y = y < 0 ? 0 : (y > size - 1 ? size - 1 : y);
        out[i] = (float)gray[(size_t)y * size + x] / 255.0f;
    }
    free(gray);
}

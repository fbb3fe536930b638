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
// the two half-waves are merged at the end (value, then lower index).
// A workgroup is 4 waves = 128 queries; every wave keeps ITS 32 queries in registers for the whole kernel (the B operand
// of v_mfma_f32_32x32x2_f32 is one VGPR per k pair: 64 VGPRs for d = 128), so LDS holds only the double-buffered
// candidate tiles (2 x 33 KB) and TWO workgroups share a CU.  The reduction of a stage (row scan + the 64-bit key butterfly of
// the single-evaluation form: ~550 vector instructions per wave after the round-3 diet, 1 100 before) follows its 128 MFMAs
// in every wave of a workgroup at the same time; on this chip the fp32 matrix time and the vector time of a SIMD ADD UP (PMC),
// so what counts is the number of vector instructions per MFMA, not how they are overlapped - see red_step.
// Roofline: MFMA-bound (2 * n1 * n2 * 128 * 2 FLOP per pair incl. both directions; 128 MFLOP at 500 keypoints).
#include "common.h"

namespace {

constexpr int QB = 128;            // queries per workgroup (32 per wave)
constexpr int CB = 64;             // candidates per stage (two MFMA tiles per wave)
constexpr int NTM = 256;           // threads per workgroup
constexpr int LDD = SSLAM_D + 4;   // 132-float rows: 528 B = 33 x 16 B -> conflict-free b128 fragment reads

// rows x 128 floats -> KP8 image, split into a load half (global -> registers) and a store half (registers -> LDS)
// so that the next candidate tile is in flight while the current one is multiplied.
// rows beyond n_valid are zero (they are masked out of the arg-max anyway)
template <int ROWS>
struct Stager {
    static constexpr int ITEMS = ROWS * 16 / NTM;
    float4 lo[ITEMS], hi[ITEMS];
    bool ok[ITEMS];         // rows beyond n_valid are zeroed at store time: nothing waits for the loads before the MFMAs
    __device__ __forceinline__ void load(const float *__restrict__ src, int first, int n_valid, int tid) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int it = tid + NTM * i, row = it >> 4, g = it & 15;
            ok[i] = first + row < n_valid;
            const float4 *p = reinterpret_cast<const float4 *>(src + (long long)(ok[i] ? first + row : 0) * SSLAM_D + 8 * g);
            lo[i] = p[0];
            hi[i] = p[1];
        }
    }
    __device__ __forceinline__ void store(float *dst, int tid) const {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int it = tid + NTM * i, row = it >> 4, g = it & 15;
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
// column direction can be reduced with max() in any order (lanes, waves, workgroups) and still give the first maximum.
// qoff = +0 for a lane that holds a query (the add also turns -0 into +0: equal under the float compare of the reference),
// -inf for a lane beyond the last query: its key is then below every key of a finite similarity and never wins - no select.
// nqi = ~(query index), the low word.
__device__ __forceinline__ unsigned long long sim_key(float v, float qoff, unsigned nqi) {
    unsigned u = __float_as_uint(v + qoff);
    u ^= (unsigned)((int)u >> 31) | 0x80000000u;
    return ((unsigned long long)u << 32) | nqi;
}

// The reduction of one finished 32 x 32 tile, cut into 24 micro-steps so that the kernel can issue a few of them behind
// every group of MFMAs of the NEXT stage (software pipelining by hand; the scheduler left alone puts all MFMAs first):
//   0..7    row scan of accumulator rows 2 STEP, 2 STEP + 1: register-local first-max (+ runner-up) per query lane
//   8..15   column butterfly, xor 16: builds the 64-bit keys of rows i, i + 8 and keeps one of them (reduce-scatter)
//   16..19  xor 8 (4 exchanges), 20..21 xor 4, 22 xor 2, 23 xor 1 -> kk = key of accumulator row crow((r >> 1) & 15, h)
// Every vector instruction here takes issue time the fp32 MFMAs of the same SIMD cannot use (PMC: matrix time + vector time
// add up to the kernel time), so the steps are written for instruction COUNT:
//   * the row scan is compare + two selects per score (+ one v_med3 for the runner-up, only in the form that returns it):
//     the winner's row is kept as a stage-local literal (32 CT + its accumulator row, an inline constant of the select) and
//     turned into a candidate index once per stage; rows beyond nc are masked only in the LAST stage (MASKED);
//   * the xor-16 exchange is v_permlane16_swap on the two key halves: afterwards every lane holds (own key, partner's key)
//     of the row it keeps - no keep / send selects, no LDS crossbar.
struct RowBest {
    float best, second;
    int besti, li;          // li: literal of the stage's winner so far, -1 = the running best is from an earlier stage
};
template <int STEP, int CT, bool ONEPASS, bool MASKED, bool SECOND>
__device__ __forceinline__ void red_step(const f32x16 &acc, int sbase, int nc, int r, int h, float qoff, unsigned nqi, RowBest &rb,
                                         unsigned long long (&k)[8], unsigned long long &kk, int &jj) {
    if constexpr (STEP < 8) {
        if constexpr (STEP == 0 && CT == 0) rb.li = -1;
#pragma unroll
        for (int e = 2 * STEP; e < 2 * STEP + 2; e++) {   // branch-free form of: if (v > best) {second = best; best = v; besti = j;}
                                                          //                  else if (v > second) second = v;      (rows j >= nc skipped)
            const int lit = 32 * CT + (e & 3) + 8 * (e >> 2);      // j = sbase + lit + 4 h, increasing in (CT, e)
            float v = acc[e];
            if constexpr (MASKED) v = sbase + lit + 4 * h < nc ? v : -INFINITY;
            const bool gt = v > rb.best;
            // best >= second always: the median of (v, best, second) is best if v beats it, else max(second, v)
            if constexpr (SECOND) rb.second = __builtin_amdgcn_fmed3f(v, rb.best, rb.second);
            rb.li = gt ? lit : rb.li;
            rb.best = gt ? v : rb.best;
        }
        if constexpr (STEP == 7 && CT == 1) rb.besti = rb.li >= 0 ? sbase + 4 * h + rb.li : rb.besti;
    } else if constexpr (!ONEPASS) {
    } else if constexpr (STEP < 16) {
        constexpr int i = STEP - 8;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned long long lo = sim_key(acc[i], qoff, nqi), hi = sim_key(acc[i + 8], qoff, nqi);
        // rows of 16 lanes: the odd rows of the first operand change places with the even rows of the second, so a lane of an
        // even row (keeps i) ends up with (own lo, partner's lo) and a lane of an odd row (keeps i + 8) with (partner's hi, own hi)
        const u32x2 w0 = __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)hi, false, false);
        const u32x2 w1 = __builtin_amdgcn_permlane16_swap((unsigned)(lo >> 32), (unsigned)(hi >> 32), false, false);
        const unsigned long long a = ((unsigned long long)w1[0] << 32) | w0[0], b = ((unsigned long long)w1[1] << 32) | w0[1];
        k[i] = a > b ? a : b;
    } else if constexpr (STEP < 23) {
        constexpr int m = STEP < 20 ? 4 : (STEP < 22 ? 2 : 1);
        constexpr int i = STEP < 20 ? STEP - 16 : (STEP < 22 ? STEP - 20 : 0);
        const bool up = (r & (2 * m)) != 0;
        unsigned long long lo = k[i], hi = k[i + m];
        asm("" : "+v"(lo), "+v"(hi));          // keeps the selects from being folded into a dynamic vector index
        const unsigned long long keep = up ? hi : lo, send = up ? lo : hi;
        const unsigned long long o = __shfl_xor(send, 2 * m);
        k[i] = keep > o ? keep : o;
    } else {
        const unsigned long long o = __shfl_xor(k[0], 1);
        kk = k[0] > o ? k[0] : o;
        jj = sbase + 32 * CT + crow((r >> 1) & 15, h);       // the accumulator row this lane pair ended up holding
    }
}
// micro-steps [FIRST, FIRST + N) of the 48 of a stage (two tiles)
template <int FIRST, int N, bool ONEPASS, bool MASKED, bool SECOND>
__device__ __forceinline__ void red_steps(const f32x16 (&acc)[2], int s, int nc, int r, int h, float qoff, unsigned nqi, RowBest &rb,
                                          unsigned long long (&k)[2][8], unsigned long long (&kk)[2], int (&jj)[2]) {
    if constexpr (N > 0) {
        constexpr int ct = FIRST / 24;
        red_step<FIRST % 24, ct, ONEPASS, MASKED, SECOND>(acc[ct], s * CB, nc, r, h, qoff, nqi, rb, k[ct], kk[ct], jj[ct]);
        red_steps<FIRST + 1, N - 1, ONEPASS, MASKED, SECOND>(acc, s, nc, r, h, qoff, nqi, rb, k, kk, jj);
    }
}

// ONEPASS = false: grid (query blocks, 2 directions, pairs) - S is evaluated once per direction, nothing but the outputs
//                  is written (single pairs / small batches: twice the workgroups to spread over the CUs).
// ONEPASS = true:  grid (query blocks, 1, pairs) - S is evaluated ONCE; the column direction (nn21) is reduced over the
//                  32 query lanes of each half-wave by a reduce-scatter butterfly on sim_key()s and merged across waves
//                  and workgroups with a 64-bit atomic max into `keys` (n_pairs x n2, zeroed), decoded by keys_decode_kernel.
// SECOND: also the runner-up of the row direction (second12; the ratio-test matchers M2 / M4).
template <bool ONEPASS, bool SECOND>
__global__ __launch_bounds__(NTM, 2) void sim_argmax_kernel(const float *__restrict__ desc1, long long stride1, int n1,
                                                            const float *__restrict__ desc2, long long stride2, int n2,
                                                            int *__restrict__ nn12, float *__restrict__ s12,
                                                            int *__restrict__ nn21, float *__restrict__ s21,
                                                            float *__restrict__ second12,
                                                            unsigned long long *__restrict__ keys, int n_pairs, int qblocks) {
    __shared__ __attribute__((aligned(16))) float Cs[2 * CB * LDD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int dir = blockIdx.y;
    // XCD-aware order (speed only): the query blocks of a pair stream the same candidates; blocks b and b + 8 share an XCD
    // (round-robin dispatch), so pair p goes to XCD p % 8 and its query blocks take consecutive slots there - the candidate
    // tiles come from that XCD's L2 after the first block (PMC before: 785 MB fetched for 157 MB of descriptors)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long pair = (long long)(slot / qblocks) * 8 + xcd;
    if (pair >= n_pairs) return;
    const float *q = dir == 0 ? desc1 + pair * stride1 : desc2 + pair * stride2;
    const float *c = dir == 0 ? desc2 + pair * stride2 : desc1 + pair * stride1;
    const int nq = dir == 0 ? n1 : n2, nc = dir == 0 ? n2 : n1;
    const int q0 = (slot % qblocks) * QB;
    if (q0 >= nq) return;
    int *o_idx = (dir == 0 ? nn12 + pair * n1 : nn21 + pair * n2);
    float *o_val = dir == 0 ? (s12 ? s12 + pair * n1 : nullptr) : (s21 ? s21 + pair * n2 : nullptr);
    float *o_sec = (dir == 0 && second12) ? second12 + pair * n1 : nullptr;

    Stager<CB> sc;
    sc.load(c, 0, nc, tid);
    // this lane's query as the B operand of all 64 MFMA steps: step i multiplies k = 2 i + h
    const int qi = q0 + wave * 32 + r;
    const bool qok = qi < nq;
    float qreg[SSLAM_D / 2];
    {
        // rows beyond nq re-read row 0 and are multiplied by 0 (finite data; exact for the valid rows): a select on qok around
        // the loads makes hipcc branch around EACH load and wait for it - 32 serial memory round trips in the prologue
        const float4 *qp = reinterpret_cast<const float4 *>(q + (long long)(qok ? qi : 0) * SSLAM_D);
        const float qm = qok ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < SSLAM_D / 4; i++) {
            const float4 v = qp[i];
            qreg[2 * i] = (h ? v.y : v.x) * qm;
            qreg[2 * i + 1] = (h ? v.w : v.z) * qm;
        }
    }
    sc.store(Cs, tid);
    __syncthreads();

    RowBest rb = {-INFINITY, -INFINITY, 0x7fffffff, -1};     // second: best of the row once the winner is removed
    const float qoff = qok ? 0.0f : -INFINITY;
    const unsigned nqi = ~(unsigned)qi;
    const int nstage = (nc + CB - 1) / CB;

    // A stage = 128 MFMAs (two 32 x 32 tiles against the wave's queries) + its reduction (row scan; in the single-evaluation
    // form also the column butterfly on 64-bit keys): ~550 vector instructions that depend on the finished tiles.  Run one
    // after the other, the matrix pipe idles through every reduction - and a second workgroup on the CU does not fill the gap:
    // two waves sharing a pipe fairly finish their MFMA phases together, so they also reduce together (PMC: pipe 54 % busy,
    // MFMA + VALU time adding up to the kernel time).  The loop is therefore software-pipelined INSIDE the wave, by hand:
    // iteration s multiplies stage s and, behind every group of 8 MFMAs (512 cycles of matrix work), issues 3 of the 48
    // micro-steps that reduce stage s - 1 (held in 32 registers).  sched_barrier pins that order (left alone, hipcc issues
    // the 128 MFMAs back to back and the reduction after them; sched_group_barrier pipelines were not honoured here).
#define M1_MMA(g_, A_, acc_)                                                                                          \
    {                                                                                                                 \
        /* KP8 image: the float4 at 8 g + 4 h holds k = 8 g + 2 st + h, st = 0..3 -> MFMA step 4 g + st */            \
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>((A_) + 8 * (g_));                                          \
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>((A_) + 32 * LDD + 8 * (g_));                                \
        _Pragma("unroll") for (int st = 0; st < 4; st++) {                                                            \
            acc_[0] = mfma32(a0[st], qreg[4 * (g_) + st], acc_[0]);                                                   \
            acc_[1] = mfma32(a1[st], qreg[4 * (g_) + st], acc_[1]);                                                   \
        }                                                                                                             \
    }
    auto commit = [&](const unsigned long long (&kk)[2], const int (&jj)[2]) {
        if (ONEPASS) {
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
                if (!(r & 1) && jj[ct] < nc) atomicMax(keys + pair * n2 + jj[ct], kk[ct]);
        }
    };

    f32x16 held[2];
#pragma unroll
    for (int e = 0; e < 16; e++) held[0][e] = held[1][e] = 0.0f;
    {
        const float *A = Cs + r * LDD + 4 * h;
#pragma unroll
        for (int g = 0; g < SSLAM_D / 8; g++) M1_MMA(g, A, held)
    }
    if (nstage > 1) {
        sc.load(c, CB, nc, tid);
        sc.store(Cs + CB * LDD, tid);
    }
    __syncthreads();
    unsigned long long kq[2][8];
    for (int s = 1; s < nstage; s++) {
        if (s + 1 < nstage) sc.load(c, (s + 1) * CB, nc, tid);     // in flight during the MFMAs below
        const float *A = Cs + (s & 1) * CB * LDD + r * LDD + 4 * h;
        f32x16 acc[2];
        unsigned long long kk[2] = {0ull, 0ull};
        int jj[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 16; e++) acc[0][e] = acc[1][e] = 0.0f;
#define M1_SLOT(g_)                                                                                                   \
        M1_MMA(g_, A, acc)                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        red_steps<3 * (g_), 3, ONEPASS, false, SECOND>(held, s - 1, nc, r, h, qoff, nqi, rb, kq, kk, jj);                              \
        __builtin_amdgcn_sched_barrier(0);
        M1_SLOT(0) M1_SLOT(1) M1_SLOT(2) M1_SLOT(3) M1_SLOT(4) M1_SLOT(5) M1_SLOT(6) M1_SLOT(7)
        M1_SLOT(8) M1_SLOT(9) M1_SLOT(10) M1_SLOT(11) M1_SLOT(12) M1_SLOT(13) M1_SLOT(14) M1_SLOT(15)
#undef M1_SLOT
        held[0] = acc[0];
        held[1] = acc[1];
        if (s + 1 < nstage) sc.store(Cs + ((s + 1) & 1) * CB * LDD, tid);
        // the atomics go out AFTER the tile stores: memory operations complete in order in the wave's counter, and the wait for the
        // tile's loads in front of the stores would otherwise also wait for two memory-side atomics (a round trip to the
        // memory-side atomic unit) in every stage; issued here they are in flight across the barrier and the next stage
        __builtin_amdgcn_sched_barrier(0);
        commit(kk, jj);
        __syncthreads();
    }
#undef M1_MMA
    {
        unsigned long long kk[2] = {0ull, 0ull};
        int jj[2] = {0, 0};
        red_steps<0, 48, ONEPASS, true, SECOND>(held, nstage - 1, nc, r, h, qoff, nqi, rb, kq, kk, jj);    // the last stage: rows beyond nc masked
        commit(kk, jj);
    }
    float best = rb.best, second = rb.second;
    int besti = rb.besti;
    // merge the two half-waves (same query, interleaved candidate rows)
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
    if (h == 0 && qok) {
        // finite descriptors always leave a candidate index here; a row of NaN similarities (non-finite input: results
        // undefined, see sslam_hip.h) never beats -inf and would leave the initial 0x7fffffff - kept in range for the callers
        o_idx[qi] = (unsigned)besti < (unsigned)nc ? besti : 0;
        if (o_val) o_val[qi] = best;
        if (o_sec) o_sec[qi] = second;
    }
}

__global__ __launch_bounds__(256) void keys_decode_kernel(const unsigned long long *__restrict__ keys, long long n, int n1,
                                                           int *__restrict__ nn21, float *__restrict__ s21) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long kk = keys[i];
    unsigned u = (unsigned)(kk >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    // with finite similarities a lane that holds a query always wins (a lane beyond the last query contributes the key of
    // -inf with an index >= n1); NaN / Inf descriptors can let such a key, or the zero fill, win: keep the index in range
    const unsigned qi = ~(unsigned)kk;
    nn21[i] = qi < (unsigned)n1 ? (int)qi : 0;
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
            if ((unsigned)j < (unsigned)n2 && nn21[j] == i) {                 // :149 (the range check: arrays not from sslam_sim_argmax)
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
    // slots past the count are zeroed, so that the fixed-capacity arrays are a pure function of the inputs (whole-array
    // comparisons across chunkings / ranks hold; the reference's arrays simply end at the count)
    for (int sl = running + tid; sl < n1; sl += 256) {
        matches[2 * sl] = 0;
        matches[2 * sl + 1] = 0;
        quality[sl] = 0.f;
    }
}

}  // namespace

// scratch of the single-evaluation form: one 64-bit (value, ~index) key per (pair, candidate)
extern "C" long long sslam_sim_argmax_workspace_bytes(int n2, int n_pairs) {
    if (n2 <= 0 || n_pairs <= 0) return SSLAM_E_INVALID;
    const int forced = (int)sslam_knob(KNOB_M1_VARIANT, 0);
    if (forced == 1 || (forced == 0 && n_pairs < 16)) return 0;      // the two-pass form: no scratch
    return (long long)n_pairs * n2 * (long long)sizeof(unsigned long long);
}

extern "C" int sslam_sim_argmax_ws(const float *desc1, long long stride1, int n1, const float *desc2, long long stride2,
                                   int n2, int n_pairs, int32_t *nn12, float *s12, int32_t *nn21, float *s21, float *second12,
                                   void *workspace, long long workspace_bytes, void *stream) {
    if (!desc1 || !desc2 || !nn12 || !nn21 || n1 <= 0 || n2 <= 0 || n_pairs <= 0) return SSLAM_E_INVALID;
    if (((uintptr_t)desc1 | (uintptr_t)desc2) & 15 || (stride1 & 3) || (stride2 & 3) || ((uintptr_t)workspace & 7)) return SSLAM_E_INVALID;
    if ((long long)n_pairs * ((((n1 > n2 ? n1 : n2) + QB - 1) / QB)) > 0x7ffffff0LL / 8) return SSLAM_E_UNSUPPORTED;
    // test-only knob (common.h): 0 = by batch size, 1 = S per direction (no workspace), 2 = S once + 64-bit key reduction
    const int forced = (int)sslam_knob(KNOB_M1_VARIANT, 0);
    hipStream_t st = (hipStream_t)stream;
    const long long need = (long long)n_pairs * n2 * (long long)sizeof(unsigned long long);
    const bool have_ws = workspace && workspace_bytes >= need;
    if (forced == 2 && !have_ws) return SSLAM_E_INVALID;
    if (have_ws && (forced == 2 || (forced == 0 && n_pairs >= 16))) {
        unsigned long long *keys = (unsigned long long *)workspace;      // CALLER-OWNED scratch: the library allocates nothing
        if (hipMemsetAsync(keys, 0, (size_t)need, st) != hipSuccess) return SSLAM_E_LAUNCH;
        const int qb1 = (n1 + QB - 1) / QB;
        const dim3 grid((unsigned)((n_pairs + 7) / 8 * 8 * qb1), 1, 1);
        if (second12)
            hipLaunchKernelGGL((sim_argmax_kernel<true, true>), grid, dim3(NTM), 0, st, desc1, stride1, n1, desc2, stride2, n2, nn12, s12,
                               nn21, s21, second12, keys, n_pairs, qb1);
        else
            hipLaunchKernelGGL((sim_argmax_kernel<true, false>), grid, dim3(NTM), 0, st, desc1, stride1, n1, desc2, stride2, n2, nn12, s12,
                               nn21, s21, second12, keys, n_pairs, qb1);
        SSLAM_CHECK_LAUNCH();
        const long long n = (long long)n_pairs * n2;
        hipLaunchKernelGGL(keys_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, n, n1, nn21, s21);
        SSLAM_CHECK_LAUNCH();
        return SSLAM_OK;
    }
    const int nmax = n1 > n2 ? n1 : n2;
    const int qbm = (nmax + QB - 1) / QB;
    const dim3 grid((unsigned)((n_pairs + 7) / 8 * 8 * qbm), 2, 1);
    if (second12)
        hipLaunchKernelGGL((sim_argmax_kernel<false, true>), grid, dim3(NTM), 0, st, desc1, stride1, n1, desc2, stride2, n2, nn12, s12, nn21,
                           s21, second12, nullptr, n_pairs, qbm);
    else
        hipLaunchKernelGGL((sim_argmax_kernel<false, false>), grid, dim3(NTM), 0, st, desc1, stride1, n1, desc2, stride2, n2, nn12, s12, nn21,
                           s21, second12, nullptr, n_pairs, qbm);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

// the form without a workspace: never allocates - the similarity matrix is evaluated once per direction
extern "C" int sslam_sim_argmax(const float *desc1, long long stride1, int n1, const float *desc2, long long stride2,
                                int n2, int n_pairs, int32_t *nn12, float *s12, int32_t *nn21, float *s21, float *second12,
                                void *stream) {
    return sslam_sim_argmax_ws(desc1, stride1, n1, desc2, stride2, n2, n_pairs, nn12, s12, nn21, s21, second12, nullptr, 0, stream);
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

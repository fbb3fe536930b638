// refine.hip - A6 + A7: bilinear feature gather and the descriptor MLP, fused per 32-row tile.
// Replaces DinoBackbone.extract_at_keypoints (reference semantic-slam/models/dino_backbone.py:114-152) and
// DescriptorRefiner.forward / ResidualBlock.forward (semantic-slam/models/descriptor_refiner.py:58-126).
//
// One workgroup (4 waves; three are co-resident per CU) owns 32 keypoint rows for the WHOLE chain: gather ->
// input_proj+ReLU -> n_blocks x {LN, fc1, ReLU, LN, fc2, +identity, ReLU} -> output_proj -> L2 normalise.  The 32x384
// activation tile lives in LDS (KP8 order, 388-float rows).  The products are evaluated TRANSPOSED: the weight
// fragment is the MFMA A operand and the activation rows the B operand, so a lane's accumulators hold ONE activation row
// (96 of its columns per wave).  ReLU, the residual identity and LayerNorm then work on registers: row statistics are
// lane-local sums, one xor-32 shuffle and a 4-wave exchange through LDS; the tile is written once per layer in 8-byte
// pairs.  (The earlier row-per-wave LayerNorm on the LDS tile cost ~1 000 instructions per call, and every instruction
// a wave issues besides its MFMAs takes issue time from the matrix pipe it shares - DESIGN_HISTORY.md section 9.)
// The pre-packed weights (3.17 MB, L2-resident, shared by all workgroups) are NOT staged through LDS: they are stored in
// MFMA-fragment order ([k/8][n][8 floats KP8]) and fetched by buffer loads with scalar (layer, k-group, tile) offsets,
// 1 KB coalesced per wave-instruction, two k-groups ahead.  A layer's GEMM has no barrier.
// All contractions are v_mfma_f32_32x32x2_f32 chains in increasing k from the bias: bit-identical to the oracle.
//
// Roofline: MFMA-bound, 1 572 864 FLOP per row (786.4 MFLOP per 500-keypoint frame) against 1.5 KB in / 0.5 KB out.
#include "common.h"
#include "gather_taps.h"

namespace {

#ifndef SSLAM_REFINE_WMR
#define SSLAM_REFINE_WMR 1
#endif
constexpr int WMR = SSLAM_REFINE_WMR;   // waves along M: 1 -> 32-row workgroups of 4 waves (3 co-resident per CU, so the
                                        // gather / LayerNorm phases of one overlap the GEMMs of the others); 2 -> 64 rows
constexpr int RM = 32 * WMR;        // rows per workgroup
constexpr int NTHR = 256 * WMR;
constexpr int HID = SSLAM_HID;      // 384
constexpr int LDH = HID + 4;        // activation row stride (floats)
constexpr int NKG = HID / 8;        // 48 k-groups of 8 per layer
constexpr int H_FLOATS = RM * LDH;
constexpr int SCRATCH_FLOATS = 256;     // two [4][32] row-statistic exchange buffers
constexpr int SMEM_FLOATS = H_FLOATS + SCRATCH_FLOATS;   // 100 352 B

struct RefArgs {
    const float *packed;
    sslam_refiner_layout_t lay;
};

// acc[t] (32 rows x 32 cols each, t-th N tile of this wave) = bias + H(64 x 384) . W^T, one fma chain per output in
// increasing k.  B fragments come straight from global memory (fragment-ordered packed weights), ring of 3 k-groups.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void gemm_lds(const float *H, __amdgpu_buffer_rsrc_t wrs, int w_off, const float *__restrict__ bias,
                                         int tid, f32x16 (&acc)[NT]) {
    constexpr int N = 128 * NT;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
    // TRANSPOSED product: the weight fragment is the MFMA A operand and the activation row the B operand, so the
    // accumulator of lane (r, h) holds activation row m = r and output columns n = slab + 32t + crow(e, h) (runs of
    // four consecutive n).  The fma chain of every output is unchanged (a*b commutes, k ascending from the bias).
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + wn * 32 * NT + t * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[t][4 * q + i] = bv[i];
        }
    // lane's B fragment of k-group g, tile t: 16 B at ((g*N + n)*8 + 4h) floats, n = wn*32*NT + t*32 + r.  Buffer loads:
    // address = descriptor base + 32-bit lane offset (constant) + SCALAR offset of (layer, k-group, tile), so the k loop
    // carries no vector address arithmetic at all - every non-MFMA instruction costs matrix-pipe issue time (DESIGN_HISTORY.md section 9)
    const int loff = ((wn * 32 * NT + r) * 2 + h) * 16;
    const float *A = H + (wm * 32 + r) * LDH + 4 * h;
    f32x4 b0[NT], b1[NT], b2[NT];
#define LOAD_B(dst, g)                                                               \
    _Pragma("unroll") for (int t = 0; t < NT; t++)                                   \
        dst[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, w_off + ((g) * N + t * 32) * 32, 0));
#define STEP(cur, g)                                                                  \
    {                                                                                 \
        const f32x4 a = an;                                                           \
        an = *reinterpret_cast<const f32x4 *>(A + 8 * (((g) + 1) < NKG ? (g) + 1 : (g))); /* next k-group's A */ \
        _Pragma("unroll") for (int st = 0; st < 4; st++)                              \
            _Pragma("unroll") for (int t = 0; t < NT; t++) acc[t] = mfma32(cur[t][st], a[st], acc[t]); \
    }
    LOAD_B(b0, 0);
    LOAD_B(b1, 1);
    f32x4 an = *reinterpret_cast<const f32x4 *>(A);
#pragma unroll 1
    for (int g = 0; g < NKG; g += 3) {
        // the refills are UNCONDITIONAL (clamped index, a redundant reload in the last trip): with a branch around them the
        // compiler cannot count the loads in flight and falls back to s_waitcnt vmcnt(0) before the third step of every
        // trip, which drains the whole prefetch ring
        // (sched_barrier: program order = issue order, or the scheduler sinks the loads next to their uses)
        LOAD_B(b2, g + 2);
        __builtin_amdgcn_sched_barrier(0);
        STEP(b0, g);
        __builtin_amdgcn_sched_barrier(0);
        LOAD_B(b0, min(g + 3, NKG - 1));
        __builtin_amdgcn_sched_barrier(0);
        STEP(b1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        LOAD_B(b1, min(g + 4, NKG - 1));
        __builtin_amdgcn_sched_barrier(0);
        STEP(b2, g + 2);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef LOAD_B
#undef STEP
}

// this wave's 32 rows x 96 columns (transposed accumulators) -> activation tile (KP8 positions): the four consecutive
// columns c .. c + 3 of an accumulator quad land at floats 0, 4, 1, 5 of their group of 8.  Two ds_write2_b32 per quad, written
// as such: from C++ stores hipcc builds 8-byte writes of the pairs (c, c + 2), (c + 1, c + 3), which need those pairs in adjacent
// registers - six v_mov per quad, and every vector instruction of a hand-over is issue time the matrix pipe does not get.
// (The compiler does not count these LDS operations: lds_quads_done() waits for them before the barrier that publishes the tile.)
template <int V>
struct IC {
    static constexpr int value = V;
};
template <int DW>
__device__ __forceinline__ void store_quad(unsigned row_lds, float y0, float y1, float y2, float y3) {
    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(row_lds), "v"(y0), "v"(y1), "n"(DW), "n"(DW + 4) : "memory");
    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(row_lds), "v"(y2), "v"(y3), "n"(DW + 1), "n"(DW + 5) : "memory");
}
__device__ __forceinline__ void lds_quads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const float *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float *)p;
}
template <int I>
__device__ __forceinline__ void store_rows_from(unsigned row_lds, const f32x16 (&v)[3]) {
    if constexpr (I < 12) {
        constexpr int t = I / 4, q = I % 4;
        store_quad<32 * t + 8 * q>(row_lds, v[t][4 * q], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]);
        store_rows_from<I + 1>(row_lds, v);
    }
}
__device__ __forceinline__ void store_rows(float *H, int tid, const f32x16 (&v)[3]) {
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wn = wave & 3;
    store_rows_from<0>(lds_addr(H + r * LDH + wn * 96 + 2 * h), v);
    lds_quads_done();
}

// lane-local partial p of one row statistic -> its total over the row's 384 (or 128) columns, in the canonical tree of the
// oracle (slab_total): the two half-waves of a wave, then the four waves in order.  `part` is a [4][32] LDS array; the
// caller alternates between two of them so that consecutive reductions need no extra barrier.
__device__ __forceinline__ float row_total(float p, float *part, int tid) {
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wn = wave & 3;
    p = p + __shfl_xor(p, 32);
    if (h == 0) part[wn * 32 + r] = p;
    __syncthreads();
    return ((part[r] + part[32 + r]) + part[64 + r]) + part[96 + r];
}

// LayerNorm(384) of the wave's transposed accumulators (row m = r per lane), written to the activation tile as the next
// GEMM's operand; v itself (the residual identity, where it is one) stays untouched in registers.  Oracle: layernorm384.
// The affine parameters of the lane's 48 columns are 12 (gamma, beta) quads from the L2-resident packed buffer.  They are
// requested THREE QUADS AHEAD through a register ring, the first three before the row statistics: loaded where they are used,
// hipcc waits for each pair right behind its request - 12 serial L2 round trips per call, most of a hand-over's 40-50 k cycles.
__device__ __forceinline__ void layernorm_store(float *H, float *part0, float *part1, const float *__restrict__ gam,
                                                const float *__restrict__ bet, int tid, const f32x16 (&v)[3]) {
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wn = wave & 3;
    const float *gp = gam + wn * 96 + 4 * h, *bp = bet + wn * 96 + 4 * h;      // quad i = (t, qd) = (i / 4, i % 4) at + 32 t + 8 qd
    constexpr int AHEAD = 3;
    f32x4 g4[AHEAD], b4[AHEAD];
#define LN_REQ(i_)                                                                                       \
    {                                                                                                    \
        g4[(i_) % AHEAD] = *reinterpret_cast<const f32x4 *>(gp + 32 * ((i_) / 4) + 8 * ((i_) % 4));      \
        b4[(i_) % AHEAD] = *reinterpret_cast<const f32x4 *>(bp + 32 * ((i_) / 4) + 8 * ((i_) % 4));      \
    }
#pragma unroll
    for (int i = 0; i < AHEAD; i++) LN_REQ(i)
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) s = s + v[t][e];
    const float mean = row_total(s, part0, tid) / 384.0f;
    float q = 0.0f;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float d = v[t][e] - mean;
            q = __builtin_fmaf(d, d, q);
        }
    const float var = row_total(q, part1, tid) / 384.0f;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const unsigned row_lds = lds_addr(H + r * LDH + wn * 96 + 2 * h);
    auto quad = [&](auto ic) {
        constexpr int i = decltype(ic)::value, t = i / 4, qd = i % 4;
        const f32x4 gq = g4[i % AHEAD], bq = b4[i % AHEAD];
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) y[j] = __builtin_fmaf((v[t][4 * qd + j] - mean) * rstd, gq[j], bq[j]);
        if constexpr (i + AHEAD < 12) LN_REQ(i + AHEAD)
        __builtin_amdgcn_sched_barrier(0);      // keeps the request of quad i + 3 in front of the arithmetic of quad i + 1
        store_quad<32 * t + 8 * qd>(row_lds, y[0], y[1], y[2], y[3]);
    };
    quad(IC<0>{}); quad(IC<1>{}); quad(IC<2>{}); quad(IC<3>{}); quad(IC<4>{}); quad(IC<5>{});
    quad(IC<6>{}); quad(IC<7>{}); quad(IC<8>{}); quad(IC<9>{}); quad(IC<10>{}); quad(IC<11>{});
    lds_quads_done();
#undef LN_REQ
}

// Optional phase timers (build with -DSSLAM_CLOCK_PROBE, read with tools/clock_probe.py): wave 0 of the first 4096 workgroups
// records its lifetime and the shader-clock cycles it spent in the gather, the GEMM loops, the LayerNorm calls, the tile
// stores and waiting at barriers.  Compiled out of the product build.
#ifdef SSLAM_CLOCK_PROBE
__device__ unsigned long long g_probe_refine[8 * 4096];
#define PROBE_BEGIN() const unsigned long long pr_t0 = clock64(); unsigned long long pr_q = 0, pr_acc[6] = {0, 0, 0, 0, 0, 0}
#define PROBE(slot, stmt) { pr_q = clock64(); stmt; pr_acc[slot] += clock64() - pr_q; }
#define PROBE_END()                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                            \
        g_probe_refine[8 * blockIdx.x] = clock64() - pr_t0;                                                 \
        for (int i_ = 0; i_ < 6; i_++) g_probe_refine[8 * blockIdx.x + 1 + i_] = pr_acc[i_];                \
    }
#else
#define PROBE_BEGIN()
#define PROBE(slot, stmt) { stmt; }
#define PROBE_END()
#endif
// Issue priority: a wave outside its GEMM loops (gather, LayerNorm hand-over, tile stores, epilogue) runs at priority 1, inside
// them at 0: the arbiter otherwise serves the co-resident workgroups' back-to-back MFMAs first and a hand-over of ~1 000
// instructions takes 40-50 k cycles; with the hint it is through sooner and back to feeding the matrix pipe (-1.5 %).
#define PRIO_OTHER() __builtin_amdgcn_s_setprio(1)
#define PRIO_GEMM() __builtin_amdgcn_s_setprio(0)
enum { PR_GATHER = 0, PR_GEMM = 1, PR_LN = 2, PR_STORE = 3, PR_BARRIER = 4 };

__global__ __launch_bounds__(NTHR, WMR == 1 ? 3 : 2) void gather_refine_kernel(const float *__restrict__ feat, int G,
                                                             const float *__restrict__ kp_xy, int K,
                                                             const float *__restrict__ x_in, long long rows,
                                                             RefArgs args, float *__restrict__ desc) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float *H = smem, *scratch = smem + H_FLOATS;
    const int tid = threadIdx.x;
    PROBE_BEGIN();
    PRIO_OTHER();
    // XCD-aware order: workgroup b runs on XCD b % 8; give every XCD one contiguous range of row tiles so that the ~8
    // tiles gathering from one frame's feature map share that XCD's L2 instead of fetching the frame into all eight
    long long R0;
    {
        const int n_tiles = gridDim.x, b = blockIdx.x, q = n_tiles / 8, rem = n_tiles % 8, x = b % 8;
        R0 = (long long)((x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + b / 8) * RM;
    }
    const float *pk = args.packed;
    const sslam_refiner_layout_t &L = args.lay;
    // buffer descriptor over the packed weights (raw buffer, 32-bit element format word as on gfx90a/gfx94x/gfx950)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pk), 0, (int)(L.total * 4), 0x00020000);

    // ---- phase 0: fill the activation tile (gathered features, or rows of x_in) --------------------------------
    {
        // thread (row, part) moves the 8-channel chunks j*8 + part, j = 0..5: the 8 lanes of a row cover 64 CONTIGUOUS floats
        // per step - full 256-byte runs of every tap row, and 8 distinct 4-bank groups for the two ds_write_b128 (the earlier
        // part*6 + j order put the 8 lanes 48 floats apart: 4-way bank conflicts, PMC SQ_LDS_BANK_CONFLICT)
        const int row = tid >> 3, part = tid & 7;
        long long R = R0 + row;
        if (R > rows - 1) R = rows - 1;
        float *dst = H + row * LDH;
        if (feat) {
            const long long f = R / K;
            const Taps t = make_taps(feat + f * G * G * SSLAM_C, G, kp_xy[2 * R], kp_xy[2 * R + 1]);
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int c0 = 8 * (j * 8 + part);
                float4 ev, od;
                kp8_split(blend4(t, c0), blend4(t, c0 + 4), ev, od);
                *reinterpret_cast<float4 *>(dst + c0) = ev;
                *reinterpret_cast<float4 *>(dst + c0 + 4) = od;
            }
        } else {
            const float *src = x_in + R * SSLAM_C;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int c0 = 8 * (j * 8 + part);
                float4 ev, od;
                kp8_split(*reinterpret_cast<const float4 *>(src + c0), *reinterpret_cast<const float4 *>(src + c0 + 4), ev, od);
                *reinterpret_cast<float4 *>(dst + c0) = ev;
                *reinterpret_cast<float4 *>(dst + c0 + 4) = od;
            }
        }
    }
#ifdef SSLAM_CLOCK_PROBE
    pr_acc[PR_GATHER] = clock64() - pr_t0;
#endif
    PROBE(PR_BARRIER, __syncthreads();)

    // ---- input_proj + ReLU (descriptor_refiner.py:76) -----------------------------------------------------------
    // Hand-over between two GEMMs: every wave turns its accumulators into the next operand IN REGISTERS (ReLU, residual,
    // LayerNorm with lane-local row statistics) and writes the tile once; the barriers inside row_total also order the
    // tile accesses (all waves have left the GEMM that read the tile before anyone writes it).
    float *part0 = scratch, *part1 = scratch + 128;
    f32x16 X[3], acc[3];
    PRIO_GEMM(); PROBE(PR_GEMM, gemm_lds<3>(H, wrs, (int)L.in_w * 4, pk + L.in_b, tid, acc);) PRIO_OTHER();
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) X[t][e] = acc[t][e] > 0.0f ? acc[t][e] : 0.0f;

    // ---- residual blocks (descriptor_refiner.py:108-126) --------------------------------------------------------
    for (int b = 0; b < L.n_blocks; b++) {
        PROBE(PR_LN, layernorm_store(H, part0, part1, pk + L.blk[b][0], pk + L.blk[b][1], tid, X);)
        PROBE(PR_BARRIER, __syncthreads();)
        PRIO_GEMM(); PROBE(PR_GEMM, gemm_lds<3>(H, wrs, (int)L.blk[b][2] * 4, pk + L.blk[b][3], tid, acc);) PRIO_OTHER();
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[t][e] = acc[t][e] > 0.0f ? acc[t][e] : 0.0f;
        PROBE(PR_LN, layernorm_store(H, part0, part1, pk + L.blk[b][4], pk + L.blk[b][5], tid, acc);)
        PROBE(PR_BARRIER, __syncthreads();)
        PRIO_GEMM(); PROBE(PR_GEMM, gemm_lds<3>(H, wrs, (int)L.blk[b][6] * 4, pk + L.blk[b][7], tid, acc);) PRIO_OTHER();
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float v = acc[t][e] + X[t][e];
                X[t][e] = v > 0.0f ? v : 0.0f;
            }
    }
    PROBE(PR_BARRIER, __syncthreads();)            // every wave has finished reading the tile
    PROBE(PR_STORE, store_rows(H, tid, X);)
    PROBE(PR_BARRIER, __syncthreads();)

    // ---- output_proj + L2 normalise (:83-86; F.normalize eps 1e-12) --------------------------------------------
    f32x16 o[1];
    PRIO_GEMM(); PROBE(PR_GEMM, gemm_lds<1>(H, wrs, (int)L.out_w * 4, pk + L.out_b, tid, o);) PRIO_OTHER();
    {
        const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, wn = wave & 3;
        float ss = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; e++) ss = __builtin_fmaf(o[0][e], o[0][e], ss);
        const float den = fmaxf(sqrtf(row_total(ss, part0, tid)), 1e-12f);
        if (R0 + r < rows) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 w4;
#pragma unroll
                for (int i = 0; i < 4; i++) w4[i] = o[0][4 * q + i] / den;
                *reinterpret_cast<f32x4 *>(desc + (R0 + r) * SSLAM_D + wn * 32 + 8 * q + 4 * h) = w4;
            }
        }
    }
    PROBE_END();
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
    hipLaunchKernelGGL(gather_refine_kernel, dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, feat, G, kp_xy, K, x_in, rows,
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

// w (n_out, k_in) -> [k-group = k/8][n][8 floats in KP8 order]: the MFMA B-fragment order
extern "C" int sslam_pack_linear_host(const float *w, int n_out, int k_in, float *out) {
    if (!w || !out || n_out <= 0 || k_in <= 0 || (k_in % 8)) return SSLAM_E_INVALID;
    for (int n = 0; n < n_out; n++)
        for (int k = 0; k < k_in; k++)
            out[((long long)(k / 8) * n_out + n) * 8 + kp8(k % 8)] = w[(long long)n * k_in + k];
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

#ifdef SSLAM_CLOCK_PROBE
extern "C" int sslam_probe_refine(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe_refine), sizeof(unsigned long long) * 8 * 4096) == hipSuccess ? 0 : -3;
}
#endif

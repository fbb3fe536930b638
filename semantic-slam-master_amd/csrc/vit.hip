// vit.hip - A1: DINOv3 ViT-S/16 forward, the stage the reference delegates to the third-party `timm`
// package (call site semantic-slam/models/dino_backbone.py:85).  Architecture restated from the public DINOv3
// definition (see sslam_amd/vit.py, the torch-op reference this is checked against).
//
// Numerics: bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; LayerNorm, softmax, RoPE, GELU, LayerScale
// and the residual stream stay fp32.  This stage is tolerance-checked (it replaces an fp32 third-party model whose
// pretrained weights are a remote fetch), unlike the bit-exact authored path that consumes its tokens.
//
// Per layer: LN1 -> [QKV GEMM + bias + RoPE + 1/sqrt(d) scale, scattered to (B, H, T, 64)] -> flash-style attention ->
// [o_proj GEMM + bias, LayerScale, residual add in place] -> LN2 -> [up GEMM + bias + exact GELU] ->
// [down GEMM + bias, LayerScale, residual add].  One "A-resident" bf16 GEMM (below) with the epilogue as a template
// functor; weights are pre-packed into MFMA B-fragment order.  Attention keeps the query on the lane (S^T = K.Q^T), so the
// running max / sum are lane-local and the probability tile is fed to the P.V MFMA straight from the accumulator
// registers (no LDS round trip for P); K is rotated and V transposed once per layer by kv_prep_kernel.
#include "common.h"

#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's uint4 struct did not)

namespace {

constexpr int VD = 384, VH = 6, VHD = 64, VMLP = 1536, VPATCH = 16, VPREFIX = 5, VLAYERS = 12;

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------- row-tile GEMM: A in registers, B through LDS
// C (M x N) = A (M x K, K = 384 * KC) . W^T with W pre-packed by sslam_vit_pack_linear_host in the order the kernel
// consumes it: [N/192][K/16][6][64 lanes][8] bf16 - one 1 KB MFMA fragment per (192-column tile, k-step, 32-column
// slice), so a "group" of 4 k-steps is 24 KB of CONTIGUOUS global memory.
//
// Workgroup = 4 waves = 128 rows x (nt_per_part tiles of 192 columns); wave w owns rows 32w..32w+31 and keeps their A
// fragments for the current 384-wide K chunk in REGISTERS (24 fragments = 96 VGPRs), so the matrix pipe needs ONE
// ds_read_b128 per MFMA.  The weight stream goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: a linear copy, 6 KB per
// wave and group, no registers, no address arithmetic) into a 2 x 24 KB ring: group g+1 is in flight while group g is
// multiplied; one barrier per group.  Two workgroups per CU (2 x 66 KB LDS, 8 waves) so that one's prologue / epilogue
// overlaps the other's MFMAs.
//
// Every global access of activations is a FULL 128-byte line per 8 lanes: rows are loaded / stored in row-major pieces
// (lane = (row q of 8, 16-byte piece p of 8)) and transposed to / from the MFMA layouts through a 4.5 KB wave-private LDS
// tile (144-byte rows: conflict-free fragment reads) - fragment-shaped global accesses (32 rows x 32 B per instruction)
// cost 2-4x the time of the whole MFMA loop (measured: o_proj 37 us with them, 10 us of MFMA loop).
// The product is evaluated TRANSPOSED (weights = MFMA A operand, activations = B operand): a lane owns ONE token and 4
// consecutive output features per accumulator quad, so the epilogue stages 8- / 16-byte pieces.
//
// LayerNorm is folded in: the residual epilogues (o_proj, down_proj, patch embedding) emit per-row partial (sum, sum of
// squares) of the NEW residual stream for their 192-column half, and the prologue of the consuming GEMM (QKV, up_proj)
// normalises the fp32 rows while it converts them to bf16 fragments - no LayerNorm kernel, no bf16 copy of the stream.
constexpr int RT_BM = 128, RT_NT = 192, RT_SL = RT_NT / 32, RT_GK = 4, RT_KS = 24, RT_NG = RT_KS / RT_GK;
constexpr int RT_STEP_ELEMS = RT_SL * 512;                 // bf16 elements per k-step slab (6 fragments of 1 KB)
constexpr int RT_GROUP_BYTES = RT_GK * RT_SL * 1024;        // 24 576
constexpr int RT_PIECES = RT_GK * RT_SL / 4;                // 1 KB DMA pieces per wave and group (6)
constexpr int RT_STG_ROW = 144, RT_STG_BYTES = 32 * RT_STG_ROW;            // wave-private transposition tile
constexpr int RT_LDS_BYTES = 2 * RT_GROUP_BYTES + 4 * RT_STG_BYTES;         // 67 584 -> two workgroups per CU

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void rt_dma_group(const bf16 *src, char *dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < RT_PIECES; i++) {
        const int piece = wave * RT_PIECES + i;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 512 + lane * 8), (lptr_t)(dst + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (bf16)lo;
    v[1] = (bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// Activation loads that hipcc must not schedule or wait for: beside LDS-DMA traffic it serialises every ordinary global
// load (load, s_waitcnt vmcnt(0), use - one memory round trip EACH; measured 37k cycles for the 52 loads of the LayerNorm
// prologue).  These asm loads are invisible to its counters; rt_wait<N>() is the hand-placed s_waitcnt vmcnt(N) that also
// names the registers it guards (so that no use is moved above it).  Vector-memory operations retire in issue order.
__device__ __forceinline__ void rt_gload(f32x4 &dst, const float *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void rt_gload(bf16x8 &dst, const bf16 *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N, class T>
__device__ __forceinline__ void rt_wait4(T &a, T &b, T &c, T &d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

// ---- prologues: fill the wave's 24 A fragments (rows row0 .. row0+31, k chunk kc) -------------------------------------
// fragment ks of lane (r = lane & 31, h = lane >> 5) = the 8 bf16 at [row r][16 ks + 8 h ..]

struct ProBf16 {          // A is a bf16 row-major matrix with row length K
    const bf16 *A;
    int K;
    static constexpr int VEC = 0;
    __device__ __forceinline__ void fill(float *vec, int tid) const {}
    __device__ __forceinline__ void load(bf16x8 (&a)[RT_KS], int row0, int kc, char *stg, const float *vec, int lane, int M) const {
        const int q = lane >> 3, p = lane & 7, r = lane & 31, h = lane >> 5;
        const bf16 *ar[4];
#pragma unroll
        for (int i = 0; i < 4; i++) ar[i] = A + (long long)min(row0 + 8 * i + q, M - 1) * K + kc * 384 + 8 * p;
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) rt_gload(a[4 * c + i], ar[i] + 64 * c);
#pragma unroll
        for (int c = 0; c < 6; c++) {
            if (c == 0) rt_wait4<20>(a[0], a[1], a[2], a[3]);          // batch c has landed: 4 (5 - c) younger loads may fly on
            if (c == 1) rt_wait4<16>(a[4], a[5], a[6], a[7]);
            if (c == 2) rt_wait4<12>(a[8], a[9], a[10], a[11]);
            if (c == 3) rt_wait4<8>(a[12], a[13], a[14], a[15]);
            if (c == 4) rt_wait4<4>(a[16], a[17], a[18], a[19]);
            if (c == 5) rt_wait4<0>(a[20], a[21], a[22], a[23]);
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<bf16x8 *>(stg + (8 * i + q) * RT_STG_ROW + 16 * p) = a[4 * c + i];
#pragma unroll
            for (int j = 0; j < 4; j++) a[4 * c + j] = *reinterpret_cast<const bf16x8 *>(stg + r * RT_STG_ROW + 32 * j + 16 * h);
        }
    }
};

struct ProLN {            // A = LayerNorm(x) of the fp32 residual stream, statistics from the producing epilogue
    const float *x;
    const float4 *stats;                  // per row: (sum, sumsq) of columns 0..191, (sum, sumsq) of columns 192..383
    const float *gamma, *beta;
    float eps;
    static constexpr int VEC = 2 * VD;    // gamma, beta cached in LDS: no global-load latency between the row batches
    __device__ __forceinline__ void fill(float *vec, int tid) const {
        for (int i = tid; i < VD; i += 256) {
            vec[i] = gamma[i];
            vec[VD + i] = beta[i];
        }
    }
    __device__ __forceinline__ void load(bf16x8 (&a)[RT_KS], int row0, int kc, char *stg, const float *vec, int lane, int M) const {
        const int q = lane >> 3, p = lane & 7, r = lane & 31, h = lane >> 5;
        // The transposition tile of THIS prologue: rows of exactly 128 B, the 16-byte chunk c of row r at chunk c ^ tau(r),
        // tau = (r1, r3, r4 ^ r0) (bits of r, low to high).  Writes are row-major 8-byte units - a ds_write_b64 group is 16
        // contiguous lanes = 2 rows x 8 units of one 64-byte half: tau's top bit differs between rows r and r + 1, so the two
        // rows take opposite halves of the 128-byte bank window (with the 144-byte rows of the other users they overlapped:
        // 2-way on every store).  Reads are the fragment side - lane (r, h) takes chunk 2 j + h of row r - and in each 16-lane
        // group of a ds_read_b128 the eight even rows have eight different (r1, r3, r4), the odd rows likewise, and even / odd rows
        // sit in opposite halves of the 256-byte bank row: conflict-free both ways.
        const int tq = ((q >> 1) & 1) | ((q & 1) << 2);                                   // tau of row 8 i + q, up to the i bits
        const int tr = ((r >> 1) & 1) | (((r >> 3) & 1) << 1) | ((((r >> 4) ^ r) & 1) << 2);
        char *wbase = stg + q * 128 + 8 * p;
        const char *rbase = stg + r * 128;
        const int hr = h ^ tr;
        float mean[4], rstd[4], nmr[4];
        const float *xr[4];
        f32x4 st4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) rt_gload(st4[i], reinterpret_cast<const float *>(stats + min(row0 + 8 * i + q, M - 1)));
        rt_wait4<0>(st4[0], st4[1], st4[2], st4[3]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = min(row0 + 8 * i + q, M - 1);
            const f32x4 st = st4[i];
            mean[i] = (st.x + st.z) * (1.0f / VD);
            const float var = fmaxf((st.y + st.w) * (1.0f / VD) - mean[i] * mean[i], 0.0f);
            rstd[i] = rsqrtf(var + eps);
            nmr[i] = -mean[i] * rstd[i];
            xr[i] = x + (long long)row * VD + 4 * p;
        }
        // six batches of 64 features (8 x 16 bytes per lane each).  ALL 48 loads are requested up front - 192 registers, free at
        // this point: no accumulator is live and the fragments are produced batch by batch - so the rows cost ONE memory latency
        // instead of one per batch (measured: the two-deep pipeline spent 11-15 k cycles here, three exposed round trips)
        f32x4 xq[6][2][4];
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int i = 0; i < 4; i++) rt_gload(xq[c][hf][i], xr[i] + 64 * c + 32 * hf);
#define LN_WAIT(src, n_)                                                                                             \
    rt_wait4<n_>(src[0][0], src[0][1], src[0][2], src[0][3]);                                                         \
    rt_wait4<n_>(src[1][0], src[1][1], src[1][2], src[1][3]);
#define LN_EMIT(src, c_)                                                                                             \
    _Pragma("unroll") for (int hf = 0; hf < 2; hf++) {                                                                \
        const float4 gm = *reinterpret_cast<const float4 *>(vec + 64 * (c_) + 32 * hf + 4 * p);                       \
        const float4 bt = *reinterpret_cast<const float4 *>(vec + VD + 64 * (c_) + 32 * hf + 4 * p);                  \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                               \
            const f32x4 v = src[hf][i];                                                                               \
            uint2 o;                                                                                                  \
            /* (v - mean) rstd gamma + beta as two fused multiply-adds per value (nmr = -mean rstd): half the vector */   \
            /* instructions of the sub / mul / mul / add form - this prologue is a quarter of the QKV wave's instruction stream */ \
            o.x = pack_bf16x2(__builtin_fmaf(__builtin_fmaf(v.x, rstd[i], nmr[i]), gm.x, bt.x),                       \
                              __builtin_fmaf(__builtin_fmaf(v.y, rstd[i], nmr[i]), gm.y, bt.y));                      \
            o.y = pack_bf16x2(__builtin_fmaf(__builtin_fmaf(v.z, rstd[i], nmr[i]), gm.z, bt.z),                       \
                              __builtin_fmaf(__builtin_fmaf(v.w, rstd[i], nmr[i]), gm.w, bt.w));                      \
            /* unit 8 hf + p of row 8 i + q, chunk index xor tau: only bits 1..3 of the unit move, p's low bit stays */      \
            const int ti = tq ^ ((i & 1) << 1) ^ (((i >> 1) & 1) << 2);                                               \
            *reinterpret_cast<uint2 *>(wbase + 8 * i * 128 + (((8 * hf + (p & 6)) ^ (2 * ti)) - (p & 6)) * 8) = o;    \
        }                                                                                                             \
    }                                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                                                     \
        a[4 * (c_) + j] = *reinterpret_cast<const bf16x8 *>(rbase + 16 * ((2 * j) ^ hr));
        LN_WAIT(xq[0], 40)             // batch c has landed once at most 8 (5 - c) younger loads are outstanding
        LN_EMIT(xq[0], 0)
        LN_WAIT(xq[1], 32)
        LN_EMIT(xq[1], 1)
        LN_WAIT(xq[2], 24)
        LN_EMIT(xq[2], 2)
        LN_WAIT(xq[3], 16)
        LN_EMIT(xq[3], 3)
        LN_WAIT(xq[4], 8)
        LN_EMIT(xq[4], 4)
        LN_WAIT(xq[5], 0)
        LN_EMIT(xq[5], 5)
#undef LN_WAIT
#undef LN_EMIT
    }
};

// The bias of this workgroup's tiles is cached in LDS (the accumulators start from it); LayerScale and the 1/sqrt(d) of q are
// folded into the packed weights and biases on the host (sslam_amd/vit_hip.py), so the epilogues only touch activations.
__device__ __forceinline__ void rt_fill_vec(float *dst, const float *src, int nt0, int nt_per_part, int tid) {
    for (int i = tid; i < RT_NT * nt_per_part; i += 256) dst[i] = src[RT_NT * nt0 + i];
}

#ifdef SSLAM_RT_PROBE
// diagnostic build only (tools/rt_probe.py): shader-clock stamps of wave 0 per workgroup; never in the product build
__device__ unsigned long long g_rt_probe[8 * 2048];
#define RT_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define RT_WALL(var) const unsigned long long var = wall_clock64()          /* 100 MHz real-time counter: cycles / ticks = the shader clock */
#define RT_ACC(dst, t1, t0) dst += (t1) - (t0)
#else
#define RT_STAMP(var)
#define RT_WALL(var)
#define RT_ACC(dst, t1, t0)
#endif

template <int KC, int NTP, class Pro, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_rt_kernel(Pro pro, const bf16 *__restrict__ Wp, int M, int parts, Epi epi, int dbg) {
    constexpr int nt_per_part = NTP;         // 192-column tiles per workgroup (compile time: with 1 the A fragments die before the epilogue)
    extern __shared__ __attribute__((aligned(16))) char rt_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks b and b + 8 share an XCD (round-robin dispatch): give them the same row tile, so that the second part's A rows
    // come from that XCD's L2 (a speed choice only)
    const int b16 = blockIdx.x / (8 * parts), bl = blockIdx.x % (8 * parts);
    const int tile = min(b16 * 8 + (bl & 7), (M + RT_BM - 1) / RT_BM - 1), part = bl >> 3;
    const bool dup = b16 * 8 + (bl & 7) >= (M + RT_BM - 1) / RT_BM;      // padding block of the last group of 8 tiles
    const int nt0 = part * nt_per_part;
    constexpr long long NT_ELEMS = (long long)KC * RT_KS * RT_STEP_ELEMS;       // packed elements per 192-column tile
    if (dup) return;
    const int row0 = tile * RT_BM + wave * 32;
    char *stg = rt_smem + 2 * RT_GROUP_BYTES + wave * RT_STG_BYTES;
    float *vec_pro = reinterpret_cast<float *>(rt_smem + 2 * RT_GROUP_BYTES + 4 * RT_STG_BYTES), *vec_epi = vec_pro + Pro::VEC;

    // the weight stream is one linear walk: (n-tile, k chunk, k group) in loop order = memory order
    const bf16 *wnext = Wp + (long long)nt0 * NT_ELEMS;
    const int n_groups = nt_per_part * KC * RT_NG;
    int issued = 1, buf = 0;
    rt_dma_group(wnext, rt_smem, wave, lane);
    wnext += RT_GK * RT_STEP_ELEMS;
    pro.fill(vec_pro, tid);                      // small per-feature vectors (LayerNorm affine, bias) -> LDS
    rt_fill_vec(vec_epi, epi.bias, nt0, nt_per_part, tid);
    epi.fill_extra(vec_epi + RT_NT * nt_per_part, tid);
    __syncthreads();

#ifdef SSLAM_RT_PROBE
    unsigned long long p_pro = 0, p_wait = 0, p_mma = 0, p_epi = 0;
#endif
    RT_STAMP(t_begin);
    RT_WALL(w_begin);
    typename Epi::State est;
    bf16x8 a[RT_KS];
    if (KC == 1) {            // one K chunk: the fragments are loaded once, before any accumulator is live
        RT_STAMP(t_p0);
        pro.load(a, row0, 0, stg, vec_pro, lane, M);
        RT_STAMP(t_p1);
        RT_ACC(p_pro, t_p1, t_p0);
    }
#pragma unroll 1
    for (int nt = 0; nt < nt_per_part; nt++) {
        f32x16 acc[RT_SL];            // starts from the bias: feature 32 s + 8 g + 4 h + i of this tile sits in acc[s][4 g + i]
#pragma unroll
        for (int s = 0; s < RT_SL; s++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 b = *reinterpret_cast<const float4 *>(vec_epi + RT_NT * nt + 32 * s + 8 * g + 4 * (lane >> 5));
                acc[s][4 * g + 0] = b.x;
                acc[s][4 * g + 1] = b.y;
                acc[s][4 * g + 2] = b.z;
                acc[s][4 * g + 3] = b.w;
            }
        for (int kc = 0; kc < KC; kc++) {
            if (KC > 1) {
                RT_STAMP(t_p0);
                pro.load(a, row0, kc, stg, vec_pro, lane, M);
                RT_STAMP(t_p1);
                RT_ACC(p_pro, t_p1, t_p0);
            }
#pragma unroll
            for (int kg = 0; kg < RT_NG; kg++) {
                RT_STAMP(t_w0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the current group have landed
                __syncthreads();                                   // ... everyone's; and everyone left the other buffer
                RT_STAMP(t_w1);
                RT_ACC(p_wait, t_w1, t_w0);
                if (issued < n_groups) {
                    rt_dma_group(wnext, rt_smem + (buf ^ 1) * RT_GROUP_BYTES, wave, lane);
                    wnext += RT_GK * RT_STEP_ELEMS;
                    issued++;
                }
                // the epilogue's own loads (the residual rows of x) go out one group early: all but the last four A fragments
                // are dead by now, and 24 MFMAs per wave cover most of the memory latency
                if (Epi::PREFETCH && nt_per_part == 1 && kg == RT_NG - 1 && kc == KC - 1) epi.prefetch(est, row0, nt0 + nt, lane, M);
                const char *bb = rt_smem + buf * RT_GROUP_BYTES + lane * 16;
                // fragment triples, software-pipelined by hand: the three ds_read_b128 of triple i + 1 are issued BEFORE the
                // three MFMAs of triple i (96 cycles of matrix work cover the LDS latency); left alone hipcc reads two
                // fragments, waits lgkmcnt(0), issues two MFMAs - the pipe idles through every LDS round trip
                bf16x8 wa[3], wb[3];
#define RT_LD3(dst, i_) _Pragma("unroll") for (int t = 0; t < 3; t++) dst[t] = *reinterpret_cast<const bf16x8 *>(bb + ((i_) * 3 + t) * 1024);
#define RT_MM3(src, i_)                                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 3; t++) acc[((i_) & 1) * 3 + t] = mfma_bf16(src[t], a[kg * RT_GK + (i_) / 2], acc[((i_) & 1) * 3 + t]); \
    __builtin_amdgcn_sched_barrier(0);
                RT_LD3(wa, 0)
                RT_LD3(wb, 1) RT_MM3(wa, 0)
                RT_LD3(wa, 2) RT_MM3(wb, 1)
                RT_LD3(wb, 3) RT_MM3(wa, 2)
                RT_LD3(wa, 4) RT_MM3(wb, 3)
                RT_LD3(wb, 5) RT_MM3(wa, 4)
                RT_LD3(wa, 6) RT_MM3(wb, 5)
                RT_LD3(wb, 7) RT_MM3(wa, 6)
                RT_MM3(wb, 7)
#undef RT_LD3
#undef RT_MM3
                buf ^= 1;
                RT_STAMP(t_w2);
                RT_ACC(p_mma, t_w2, t_w1);
            }
        }
        RT_STAMP(t_e0);
#ifdef SSLAM_RT_PROBE
        if (dbg & 2) {
#pragma unroll
            for (int s = 0; s < RT_SL; s++) asm volatile("" ::"v"(acc[s]));
        } else
#endif
        {
            // several column tiles per workgroup: the A fragments stay live, so the epilogue's own loads cannot go out a group early
            if (Epi::PREFETCH && nt_per_part != 1) epi.prefetch(est, row0, nt0 + nt, lane, M);
            epi.tile(acc, est, row0, nt0 + nt, stg, vec_epi + RT_NT * nt_per_part, lane, M);
        }
        RT_STAMP(t_e1);
        RT_ACC(p_epi, t_e1, t_e0);
    }
#ifdef SSLAM_RT_PROBE
    if (tid == 0 && blockIdx.x < 2048) {
        unsigned long long *o = g_rt_probe + 8 * blockIdx.x;
        o[0] = __builtin_readcyclecounter() - t_begin;
        o[1] = p_pro; o[2] = p_wait; o[3] = p_mma; o[4] = p_epi;
        o[6] = wall_clock64() - w_begin; o[7] = w_begin;
    }
#endif
}

// ---- epilogues -----------------------------------------------------------------------------------------------------
// They see the transposed tile: lane (r, h) holds token row0 + r and, for slice s / quad g, the 4 consecutive output
// features 192 nt + 32 s + 8 g + 4 h + {0..3} in acc[s][4g .. 4g+3].  Pieces are staged in the wave's LDS tile and leave
// in row-major order: lane (q = lane >> 3, p = lane & 7) handles rows row0 + 8 i + q, i = 0..3, 16-byte piece p.

// x[row, f] += acc on the fp32 residual stream (o_proj / down_proj; bias and LayerScale are already inside acc), plus the
// LayerNorm partial sums of the new rows over this 192-column half (N = 384: nt is 0 or 1).  The 24 row pieces of x are
// requested up front (the A fragments are dead by now: one memory latency for the whole tile instead of one per slice).
struct EpiResidual {
    const float *bias;                    // LayerScale-folded
    float *x;
    float2 *stats;
    int extra_floats() const { return 0; }
    __device__ __forceinline__ void fill_extra(float *, int) const {}
    struct State {
        f32x4 xv[RT_SL][4];
    };
    static constexpr bool PREFETCH = true;
    __device__ __forceinline__ void prefetch(State &st, int row0, int nt, int lane, int M) const {
        const int q = lane >> 3, p = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float *px = x + (long long)min(row0 + 8 * i + q, M - 1) * VD + RT_NT * nt + 4 * p;
#pragma unroll
            for (int s = 0; s < RT_SL; s++) rt_gload(st.xv[s][i], px + 32 * s);
        }
    }
    __device__ __forceinline__ void tile(f32x16 (&acc)[RT_SL], State &st, int row0, int nt, char *stg, const float *, int lane, int M) const {
        const int q = lane >> 3, p = lane & 7, r = lane & 31, h = lane >> 5;
        float *px[4];
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = x + (long long)min(row0 + 8 * i + q, M - 1) * VD + RT_NT * nt + 4 * p;
        f32x4 (&xv)[RT_SL][4] = st.xv;
        // all 24 loads are older than anything issued since (the last group's LDS-DMA pieces included): one wait covers them
        rt_wait4<0>(xv[0][0], xv[0][1], xv[0][2], xv[0][3]);
        float sum[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < RT_SL; s++) {
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(stg + r * RT_STG_ROW + 32 * g + 16 * h) =
                    make_float4(acc[s][4 * g + 0], acc[s][4 * g + 1], acc[s][4 * g + 2], acc[s][4 * g + 3]);
            if (s > 0) rt_wait4<0>(xv[s][0], xv[s][1], xv[s][2], xv[s][3]);      // names the registers only: everything has landed
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 d = *reinterpret_cast<const float4 *>(stg + (8 * i + q) * RT_STG_ROW + 16 * p);
                f32x4 v = xv[s][i];
                v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
                if (row0 + 8 * i + q < M) *reinterpret_cast<f32x4 *>(px[i] + 32 * s) = v;
                sum[i] += (v.x + v.y) + (v.z + v.w);
                sq[i] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                sum[i] += __shfl_xor(sum[i], o);
                sq[i] += __shfl_xor(sq[i], o);
            }
            const int row = row0 + 8 * i + q;
            if (p == 0 && row < M) stats[2 * (long long)row + nt] = make_float2(sum[i], sq[i]);
        }
    }
};

// GELU(v) = v * Phi(v) (torch's default, the erf form) with Phi(v) - 1/2 = y * P(y^2), y = clamp(v, -4, 4), P the degree-6
// minimax polynomial (fitted by linear programming with Phi(4) pinned to 1, so the tails are exactly v and 0):
// |error| <= 1.9e-4 in fp32 - below the bf16 rounding of the result (and below the tanh form's 4.7e-4).  10 plain
// multiply-adds per value, no transcendental, all packable two-at-a-time (v_pk_fma_f32): the up_proj epilogue is
// VALU-bound (96 activations per lane and tile against 144 MFMAs), the erf / exp / rcp form took 14 instructions per value.
__device__ __forceinline__ float gelu_poly(float v) {
    const float y = __builtin_amdgcn_fmed3f(v, -4.0f, 4.0f);
    const float t = y * y;
    float p = 2.258823990e-08f;
    p = __builtin_fmaf(p, t, -1.588827486e-06f);
    p = __builtin_fmaf(p, t, 4.776385402e-05f);
    p = __builtin_fmaf(p, t, -8.121865301e-04f);
    p = __builtin_fmaf(p, t, 8.763681258e-03f);
    p = __builtin_fmaf(p, t, -6.455437055e-02f);
    p = __builtin_fmaf(p, t, 3.978702073e-01f);
    return v * __builtin_fmaf(y, p, 0.5f);
}

// rows of 64 bf16 (two slices = 128 B per token) leave the staging tile as full lines: dst(i) = address of the 128-byte
// row segment of token row0 + 8 i + q
//
// The bf16 pair tile (32 tokens x 128 B) is SWIZZLED at 8-byte granularity, rows of exactly 128 B: unit u (0..15) of row r sits at
// r * 128 + 8 * (u ^ sw(r)), sw(r) = 2 (r & 7) + ((r >> 3) & 1).  Writes come from the transposed side - lane (r, h) stores unit
// 2 g + h: a ds_write_b64 is served in groups of 16 CONTIGUOUS lanes (r .. r + 15, one h) against 32 banks, and with any padded
// row stride that keeps rows 16-byte aligned the lanes r and r + 8 meet on one bank (the 144-byte rows of the other epilogues:
// 2-way on every store, a quarter of this kernel's LDS cycles by SQ_LDS_BANK_CONFLICT); sw sends the 16 lanes to 16 different
// units.  Reads are row-major 16-byte pieces - lane (q, p) takes chunk p of row 8 i + q, found at chunk p ^ q, its two 8-byte
// halves exchanged in the rows with (r >> 3) & 1 = i & 1 = 1 (known at compile time: free) - and the four 16-lane groups of a
// ds_read_b128 land on 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int rt_pair_sw(int r) { return 2 * (r & 7) + ((r >> 3) & 1); }
template <class RowPtr>
__device__ __forceinline__ void rt_store_pair(const char *stg, int row0, int lane, int M, RowPtr dst) {
    const int q = lane >> 3, p = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u32x4 v = *reinterpret_cast<const u32x4 *>(stg + (8 * i + q) * 128 + 16 * (p ^ q));
        if (i & 1) v = u32x4{v[2], v[3], v[0], v[1]};
        if (row0 + 8 * i + q < M) *reinterpret_cast<u32x4 *>(dst(i) + 8 * p) = v;
    }
}

// out[row, f] = gelu(acc) as bf16       (up_proj; acc holds the bias already)
struct EpiGelu {
    const float *bias;
    bf16 *out;
    int ldo;
    int extra_floats() const { return 0; }
    __device__ __forceinline__ void fill_extra(float *, int) const {}
    struct State {};
    static constexpr bool PREFETCH = false;
    __device__ __forceinline__ void prefetch(State &, int, int, int, int) const {}
    __device__ __forceinline__ void tile(f32x16 (&acc)[RT_SL], State &, int row0, int nt, char *stg, const float *, int lane, int M) const {
        const int q = lane >> 3, r = lane & 31, h = lane >> 5;
        char *wrow = stg + r * 128;
        const int sw = rt_pair_sw(r);
#pragma unroll
        for (int pp = 0; pp < 3; pp++) {
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int s = 2 * pp + s2;
                    uint2 o;
                    o.x = pack_bf16x2(gelu_poly(acc[s][4 * g + 0]), gelu_poly(acc[s][4 * g + 1]));
                    o.y = pack_bf16x2(gelu_poly(acc[s][4 * g + 2]), gelu_poly(acc[s][4 * g + 3]));
                    *reinterpret_cast<uint2 *>(wrow + 8 * ((8 * s2 + 2 * g + h) ^ sw)) = o;
                }
            rt_store_pair(stg, row0, lane, M, [&](int i) { return out + (long long)(row0 + 8 * i + q) * ldo + RT_NT * nt + 64 * pp; });
        }
    }
};

// patch embedding: row = frame*cells + patch -> x[frame*T + 5 + patch, f] = acc; also the LayerNorm partial sums
struct EpiPatch {
    const float *bias;
    float *x;
    float2 *stats;
    int cells, T;
    int extra_floats() const { return 0; }
    __device__ __forceinline__ void fill_extra(float *, int) const {}
    struct State {};
    static constexpr bool PREFETCH = false;
    __device__ __forceinline__ void prefetch(State &, int, int, int, int) const {}
    __device__ __forceinline__ void tile(f32x16 (&acc)[RT_SL], State &, int row0, int nt, char *stg, const float *, int lane, int M) const {
        const int q = lane >> 3, p = lane & 7, r = lane & 31, h = lane >> 5;
        float sum[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
        long long orow[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = min(row0 + 8 * i + q, M - 1), f = row / cells;
            orow[i] = (long long)f * T + VPREFIX + (row - f * cells);
        }
#pragma unroll
        for (int s = 0; s < RT_SL; s++) {
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(stg + r * RT_STG_ROW + 32 * g + 16 * h) =
                    make_float4(acc[s][4 * g + 0], acc[s][4 * g + 1], acc[s][4 * g + 2], acc[s][4 * g + 3]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 v = *reinterpret_cast<const float4 *>(stg + (8 * i + q) * RT_STG_ROW + 16 * p);
                if (row0 + 8 * i + q < M) *reinterpret_cast<float4 *>(x + orow[i] * VD + RT_NT * nt + 32 * s + 4 * p) = v;
                sum[i] += (v.x + v.y) + (v.z + v.w);
                sq[i] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                sum[i] += __shfl_xor(sum[i], o);
                sq[i] += __shfl_xor(sq[i], o);
            }
            if (p == 0 && row0 + 8 * i + q < M) stats[2 * orow[i] + nt] = make_float2(sum[i], sq[i]);
        }
    }
};

// QKV: RoPE on the patch tokens of q and k, then scatter to (B, H, T, 64) bf16 (bias inside acc; the q rows of the weight /
// bias carry log2(e)/sqrt(64), folded on the host).  A 192-column tile is three heads of one of q / k / v (tiles 0-1: q,
// 2-3: k, 4-5: v), a slice pair is one head (128 B per token).  RoPE partners (d, d + 32) are the same accumulator slot of
// the two slices of a pair, on the same lane.  The (G*G, 64) cos / sin tables are AXIAL (DINOv3: d % 32 < 16 depends on the
// patch row only, the rest on the patch column only, tiled twice), so 4 x G x 16 floats describe them: cached in LDS
// (7 KB at G = 28) - per-token global loads of cos / sin either stall every quad or, prefetched, spill the A fragments.
struct EpiQKV {
    const float *bias;
    const float *cosb, *sinb;
    bf16 *q, *k, *v;
    int T, G;
    // rows of the x tables are ROPE_RS = 20 floats apart (16 used): the 16 lanes of a ds_read_b128 group hold tokens of up to 16
    // different patch columns, and with rows of 64 B the columns px and px + 4 share their four banks (4-way on every x-table
    // read); 80-byte rows put 16 consecutive columns on 16 different 16-byte slots of the 256-byte bank row.  The y tables keep
    // 64-byte rows (a group sees one or two patch rows) - at G = 28 the kernel's LDS then stays below half a CU's 160 KB.
    static constexpr int ROPE_RS = 20;
    int extra_floats() const { return 2 * 16 * G + 2 * ROPE_RS * G; }
    __device__ __forceinline__ void fill_extra(float *rope, int tid) const {
        // [cos_y | sin_y] each (G, 16), then [cos_x | sin_x] each (G, ROPE_RS): row py of the y tables from token (py, 0), row px
        // of the x tables from token (0, px)
        float *rx = rope + 32 * G;
        for (int i = tid; i < 16 * G; i += 256) {
            const int p = i >> 4, dd = i & 15, j = p * ROPE_RS + dd;
            rope[i] = cosb[(long long)p * G * VHD + dd];
            rope[16 * G + i] = sinb[(long long)p * G * VHD + dd];
            rx[j] = cosb[(long long)p * VHD + 16 + dd];
            rx[ROPE_RS * G + j] = sinb[(long long)p * VHD + 16 + dd];
        }
    }
    struct State {};
    static constexpr bool PREFETCH = false;
    __device__ __forceinline__ void prefetch(State &, int, int, int, int) const {}
    __device__ __forceinline__ void tile(f32x16 (&acc)[RT_SL], State &, int row0, int nt, char *stg, const float *rope, int lane, int M) const {
        const int q8 = lane >> 3, r = lane & 31, h = lane >> 5;
        char *wrow = stg + r * 128;
        const int sw = rt_pair_sw(r);
        const int which = nt >> 1;
        bf16 *dst = which == 0 ? q : (which == 1 ? k : v);
        // rotation coefficients of this lane's token: v tiles, [CLS] and register tokens use the identity (cos 1, sin 0)
        const int row_r = min(row0 + r, M - 1), t = row_r % T;
        const int tp = max(t - VPREFIX, 0), py = tp / G, px = tp - py * G;
        const float rot = (which < 2 && t >= VPREFIX) ? 1.0f : 0.0f, keep = 1.0f - rot;
        long long orow[4];                 // (frame * 6 * T + t) * 64 of the four rows this lane stores
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = min(row0 + 8 * i + q8, M - 1), f = row / T;
            orow[i] = ((long long)f * VH * T + (row - f * T)) * VHD;
        }
#pragma unroll
        for (int pp = 0; pp < 3; pp++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                // the accumulators are only READ (rotating them in place made hipcc keep two copies and spill the A fragments)
                const float *tc = (g < 2 ? rope + py * 16 : rope + 32 * G + px * ROPE_RS) + 8 * (g & 1) + 4 * h;
                const float4 cs = *reinterpret_cast<const float4 *>(tc), sn = *reinterpret_cast<const float4 *>(tc + (g < 2 ? 16 : ROPE_RS) * G);
                const float cv[4] = {cs.x * rot + keep, cs.y * rot + keep, cs.z * rot + keep, cs.w * rot + keep};
                const float sv[4] = {sn.x * rot, sn.y * rot, sn.z * rot, sn.w * rot};
                float o0[4], o1[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float x0 = acc[2 * pp][4 * g + i], x1 = acc[2 * pp + 1][4 * g + i];
                    o0[i] = x0 * cv[i] - x1 * sv[i];
                    o1[i] = x1 * cv[i] + x0 * sv[i];
                }
                *reinterpret_cast<uint2 *>(wrow + 8 * ((2 * g + h) ^ sw)) = make_uint2(pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3]));
                *reinterpret_cast<uint2 *>(wrow + 8 * ((8 + 2 * g + h) ^ sw)) = make_uint2(pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3]));
            }
            const int head = 3 * (nt & 1) + pp;
            rt_store_pair(stg, row0, lane, M, [&](int i) { return dst + orow[i] + (long long)head * T * VHD; });
        }
    }
};

// ------------------------------------------------------------------------- fused MLP: LN2 -> up -> GELU -> down -> residual
// One launch for the whole MLP half of a block: the 1536-wide hidden activation never leaves the chip (it was 155 MB
// written + 263 MB read per layer and 64 frames, a quarter of the layer's HBM traffic, and two of its five launches).
// Workgroup = 8 waves = 4 wave PAIRS x 32 tokens, one per CU.  Both waves of a pair keep LayerNorm(x) of their 32 tokens as
// bf16 fragments in registers (ProLN, 96 VGPRs).  The hidden dimension is walked in 24 chunks of 64; per chunk
//   U: wave `hf` of the pair multiplies hidden slice hf (32 units) over K = 384: H^T = W1 . Xn^T, hidden on accumulator rows,
//      the token on the lane (24 MFMAs), adds the bias (initial accumulator), applies GELU in registers;
//   X: the 32 x 32 bf16 tile IS the B operand of the next product for its two k-steps (accumulator registers 8 s .. 8 s + 7 of a
//      tile are the fragment of k-step s, with the k order 16 s + 8 (j >> 2) + 4 h + (j & 3) - W2 is packed in that order), so the
//      only exchange is the partner's tile, 2 KB per wave through LDS;
//   D: out^T += W2[:, chunk] . H^T for the wave's half of the 384 outputs (6 slices x 4 k-steps = 24 MFMAs, 96 accumulators).
// The weights of a chunk are 2 x 48 KB of contiguous memory (sslam_vit_pack_mlp_host) streamed by LDS-DMA into a 2 x 48 KB
// ring shared by all eight waves: group g + 1 is in flight while group g is multiplied, one barrier per group.  The
// epilogue is EpiResidual (x += out, LayerNorm partial sums of the new rows).
constexpr int MF_GROUP_BYTES = 48 * 1024, MF_CHUNKS = VMLP / 64;
static_assert(8 * 2 * 1024 <= 8 * RT_STG_BYTES, "the exchange tiles (two 1 KB fragments per wave) overlay the staging tiles");
constexpr int MF_LDS_BYTES = 2 * MF_GROUP_BYTES + 8 * RT_STG_BYTES + 4 * (2 * VD + VMLP + VD);   // ring + staging (overlaid by the exchange) + vectors

__device__ __forceinline__ void mf_dma_group(const bf16 *src, char *dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 6; i++) {                                            // 48 pieces of 1 KB over 8 waves
        const int piece = wave * 6 + i;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 512 + lane * 8), (lptr_t)(dst + piece * 1024), 16, 0, 0);
    }
}

__global__ __launch_bounds__(512, 2) void mlp_fused_kernel(ProLN pro, const bf16 *__restrict__ Wp, const float *__restrict__ b_up, int M,
                                                           EpiResidual epi) {
    extern __shared__ __attribute__((aligned(16))) char mf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), pair = wave >> 1, hf = wave & 1;
    const int row0 = blockIdx.x * RT_BM + pair * 32;
    char *stg = mf_smem + 2 * MF_GROUP_BYTES + wave * RT_STG_BYTES;
    char *xch = mf_smem + 2 * MF_GROUP_BYTES;                                 // exchange tiles overlay the staging tiles (main loop only)
    float *vec_pro = reinterpret_cast<float *>(mf_smem + 2 * MF_GROUP_BYTES + 8 * RT_STG_BYTES);
    float *vec_b1 = vec_pro + 2 * VD, *vec_b2 = vec_b1 + VMLP;

    const bf16 *wnext = Wp;
    mf_dma_group(wnext, mf_smem, wave, lane);
    wnext += MF_GROUP_BYTES / 2;
    for (int i = tid; i < VD; i += 512) {
        vec_pro[i] = pro.gamma[i];
        vec_pro[VD + i] = pro.beta[i];
        vec_b2[i] = epi.bias[i];
    }
    for (int i = tid; i < VMLP; i += 512) vec_b1[i] = b_up[i];
    __syncthreads();

#ifdef SSLAM_RT_PROBE
    unsigned long long p_pro = 0, p_wait = 0, p_mma = 0, p_epi = 0, p_gelu = 0;
#endif
    RT_STAMP(t_begin);
    RT_WALL(w_begin);
    bf16x8 a[RT_KS];
    pro.load(a, row0, 0, stg, vec_pro, lane, M);
    RT_STAMP(t_pro1);
    RT_ACC(p_pro, t_pro1, t_begin);
    typename EpiResidual::State est;
    f32x16 acc[RT_SL];            // this wave's 192 outputs of its 32 tokens; starts from the (LayerScale-folded) bias
#pragma unroll
    for (int s = 0; s < RT_SL; s++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 b = *reinterpret_cast<const float4 *>(vec_b2 + RT_NT * hf + 32 * s + 8 * g + 4 * h);
            acc[s][4 * g + 0] = b.x;
            acc[s][4 * g + 1] = b.y;
            acc[s][4 * g + 2] = b.z;
            acc[s][4 * g + 3] = b.w;
        }
    __syncthreads();             // every wave is done with its staging tile: the exchange tiles may overlay them

    int buf = 0;
    for (int c = 0; c < MF_CHUNKS; c++) {
        // ---- U: hidden slice hf of chunk c ------------------------------------------------------------------------------
        RT_STAMP(t_u0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // group U_c has landed; everyone has left the other buffer (D_{c-1})
        RT_STAMP(t_u1);
        RT_ACC(p_wait, t_u1, t_u0);
        mf_dma_group(wnext, mf_smem + (buf ^ 1) * MF_GROUP_BYTES, wave, lane);          // D_c
        wnext += MF_GROUP_BYTES / 2;
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 b = *reinterpret_cast<const float4 *>(vec_b1 + 64 * c + 32 * hf + 8 * g + 4 * h);
            hacc[4 * g + 0] = b.x;
            hacc[4 * g + 1] = b.y;
            hacc[4 * g + 2] = b.z;
            hacc[4 * g + 3] = b.w;
        }
        {
            const char *bb = mf_smem + buf * MF_GROUP_BYTES + hf * 1024 + lane * 16;
            bf16x8 wa[2], wb[2];
#define MF_LDU(dst, i_) _Pragma("unroll") for (int t = 0; t < 2; t++) dst[t] = *reinterpret_cast<const bf16x8 *>(bb + (2 * (i_) + t) * 2048);
#define MF_MMU(src, i_)                                                                                 \
    _Pragma("unroll") for (int t = 0; t < 2; t++) hacc = mfma_bf16(src[t], a[2 * (i_) + t], hacc);        \
    __builtin_amdgcn_sched_barrier(0);
            MF_LDU(wa, 0)
            MF_LDU(wb, 1) MF_MMU(wa, 0)
            MF_LDU(wa, 2) MF_MMU(wb, 1)
            MF_LDU(wb, 3) MF_MMU(wa, 2)
            MF_LDU(wa, 4) MF_MMU(wb, 3)
            MF_LDU(wb, 5) MF_MMU(wa, 4)
            MF_LDU(wa, 6) MF_MMU(wb, 5)
            MF_LDU(wb, 7) MF_MMU(wa, 6)
            MF_LDU(wa, 8) MF_MMU(wb, 7)
            MF_LDU(wb, 9) MF_MMU(wa, 8)
            MF_LDU(wa, 10) MF_MMU(wb, 9)
            MF_LDU(wb, 11) MF_MMU(wa, 10)
            MF_MMU(wb, 11)
#undef MF_LDU
#undef MF_MMU
        }
        buf ^= 1;
        RT_STAMP(t_u2);
        RT_ACC(p_mma, t_u2, t_u1);
        // GELU in registers; registers 8 s .. 8 s + 7 of the tile are the B fragment of k-step 2 hf + s of the chunk
        u32x4 own[2];
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) own[s][j] = pack_bf16x2(gelu_poly(hacc[8 * s + 2 * j]), gelu_poly(hacc[8 * s + 2 * j + 1]));
#pragma unroll
        for (int s = 0; s < 2; s++) *reinterpret_cast<u32x4 *>(xch + (wave * 2 + s) * 1024 + lane * 16) = own[s];
        // ---- D: the wave's 192 outputs over the 64 hidden units of the chunk ---------------------------------------------
        RT_STAMP(t_d0);
        RT_ACC(p_gelu, t_d0, t_u2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // group D_c has landed, U_c is consumed, both tiles of every pair are written
        RT_STAMP(t_d1);
        RT_ACC(p_wait, t_d1, t_d0);
        if (c + 1 < MF_CHUNKS) {
            mf_dma_group(wnext, mf_smem + (buf ^ 1) * MF_GROUP_BYTES, wave, lane);      // U_{c+1}
            wnext += MF_GROUP_BYTES / 2;
        }
        u32x4 other[2];
#pragma unroll
        for (int s = 0; s < 2; s++) other[s] = *reinterpret_cast<const u32x4 *>(xch + ((wave ^ 1) * 2 + s) * 1024 + lane * 16);
        {
            const char *bb = mf_smem + buf * MF_GROUP_BYTES + (6 * hf) * 1024 + lane * 16;
            bf16x8 wa[3], wb[3];
            // k-step s4 (16 hidden units) of the chunk belongs to the wave with hf == s4 >> 1
#define MF_PB(s4_) __builtin_bit_cast(bf16x8, ((s4_) >> 1) == hf ? own[(s4_) & 1] : other[(s4_) & 1])
#define MF_LDD(dst, i_) _Pragma("unroll") for (int t = 0; t < 3; t++) dst[t] = *reinterpret_cast<const bf16x8 *>(bb + (((i_) >> 1) * 12 + ((i_) & 1) * 3 + t) * 1024);
#define MF_MMD(src, i_)                                                                                              \
    _Pragma("unroll") for (int t = 0; t < 3; t++) acc[((i_) & 1) * 3 + t] = mfma_bf16(src[t], MF_PB((i_) >> 1), acc[((i_) & 1) * 3 + t]); \
    __builtin_amdgcn_sched_barrier(0);
            MF_LDD(wa, 0)
            MF_LDD(wb, 1) MF_MMD(wa, 0)
            MF_LDD(wa, 2) MF_MMD(wb, 1)
            MF_LDD(wb, 3) MF_MMD(wa, 2)
            MF_LDD(wa, 4) MF_MMD(wb, 3)
            MF_LDD(wb, 5) MF_MMD(wa, 4)
            MF_LDD(wa, 6) MF_MMD(wb, 5)
            MF_LDD(wb, 7) MF_MMD(wa, 6)
            MF_MMD(wb, 7)
#undef MF_LDD
#undef MF_MMD
#undef MF_PB
        }
        buf ^= 1;
        RT_STAMP(t_d2);
        RT_ACC(p_mma, t_d2, t_d1);
    }
    __syncthreads();             // the exchange tiles are dead: the staging tiles take the region back
    RT_STAMP(t_e0);
    epi.prefetch(est, row0, hf, lane, M);
    epi.tile(acc, est, row0, hf, stg, nullptr, lane, M);
#ifdef SSLAM_RT_PROBE
    if (tid == 0 && blockIdx.x < 2048) {
        unsigned long long *o = g_rt_probe + 8 * blockIdx.x;
        const unsigned long long t_end = __builtin_readcyclecounter();
        o[0] = t_end - t_begin;
        o[1] = p_pro; o[2] = p_wait; o[3] = p_mma; o[4] = t_end - t_e0; o[5] = p_gelu;
        o[6] = wall_clock64() - w_begin; o[7] = w_begin;
    }
#endif
}

// --------------------------------------------------------------------------------------------- LayerNorm
// one wave per row of 384: lane l holds elements l*2 + 128*j .. (float2 x 3); fp32 two-pass statistics
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                       const float *__restrict__ b, float eps, long long rows,
                                                       void *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = x + row * VD;
    float2 v[3];
#pragma unroll
    for (int j = 0; j < 3; j++) v[j] = *reinterpret_cast<const float2 *>(p + 128 * j + 2 * lane);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) s += v[j].x + v[j].y;
    const float mean = bfly64(s) * (1.0f / VD);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float dx = v[j].x - mean, dy = v[j].y - mean;
        s2 += dx * dx + dy * dy;
    }
    const float rstd = rsqrtf(bfly64(s2) * (1.0f / VD) + eps);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int c = 128 * j + 2 * lane;
        const float y0 = (v[j].x - mean) * rstd * g[c] + b[c], y1 = (v[j].y - mean) * rstd * g[c + 1] + b[c + 1];
        if (OUT_BF16) {
            bf16 *o = reinterpret_cast<bf16 *>(out) + row * VD + c;
            o[0] = (bf16)y0;
            o[1] = (bf16)y1;
        } else {
            *reinterpret_cast<float2 *>(reinterpret_cast<float *>(out) + row * VD + c) = make_float2(y0, y1);
        }
    }
}

// ------------------------------------------------------------------------------------- patches / prefix rows
// (B, 3, S, S) fp32 -> (B*G*G, 768) bf16 with k = c*256 + ky*16 + kx   (nn.Conv2d(3, 384, 16, 16) as a GEMM)
__global__ __launch_bounds__(256) void im2patch_kernel(const float *__restrict__ img, int S, long long items, bf16 *__restrict__ out) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;   // one 8-pixel run
    if (it >= items) return;
    const int G = S / VPATCH;
    const int run = (int)(it % 96);                 // 768 / 8 runs per patch
    const long long patch = it / 96;
    const int cells = G * G;
    const long long f = patch / cells;
    const int pc = (int)(patch % cells), py = pc / G, px = pc % G;
    const int c = run / 32, ky = (run % 32) / 2, kx0 = (run & 1) * 8;
    const float *src = img + ((f * 3 + c) * S + (py * VPATCH + ky)) * (long long)S + px * VPATCH + kx0;
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    bf16x8 o;
    o[0] = (bf16)a.x; o[1] = (bf16)a.y; o[2] = (bf16)a.z; o[3] = (bf16)a.w;
    o[4] = (bf16)b.x; o[5] = (bf16)b.y; o[6] = (bf16)b.z; o[7] = (bf16)b.w;
    *reinterpret_cast<bf16x8 *>(out + patch * 768 + run * 8) = o;
}

// [CLS] + register rows of every frame, and their LayerNorm sums (both halves' worth in the first slot): one wave per row
__global__ __launch_bounds__(64) void prefix_rows_kernel(const float *__restrict__ prefix, int T, float *__restrict__ x,
                                                         float2 *__restrict__ stats) {
    const int lane = threadIdx.x, i = blockIdx.x % VPREFIX;
    const long long row = (long long)(blockIdx.x / VPREFIX) * T + i;
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float2 v = *reinterpret_cast<const float2 *>(prefix + i * VD + 128 * j + 2 * lane);
        *reinterpret_cast<float2 *>(x + row * VD + 128 * j + 2 * lane) = v;
        s += v.x + v.y;
        s2 += v.x * v.x + v.y * v.y;
    }
    s = bfly64(s);
    s2 = bfly64(s2);
    if (lane == 0) {
        stats[2 * row] = make_float2(s, s2);
        stats[2 * row + 1] = make_float2(0.f, 0.f);
    }
}

// --------------------------------------------------------------------------------------------- attention
// Flash-style attention, one workgroup = 128 queries of one (frame, head), 4 waves x 32 queries, keys in tiles of 64.
// q, k arrive rotated (RoPE) and q pre-scaled by log2(e)/sqrt(64) from the QKV epilogue; v is in its natural (B, H, T, 64)
// layout.  Everything per query lives on ONE lane pair (r, r + 32):
//   S^T = K . Q^T   (keys on accumulator rows, the query on the lane)  -> max / exp2 / row sums are lane-local;
//   O^T = V^T . P^T (d on accumulator rows, the query on the lane): P^T is fed to the MFMA straight from the S^T
//   accumulator registers (B operand), V^T fragments come from the natural-layout V tile by ds_read_b64_tr_b16 (the
//   hardware transposing LDS read; tile rows of 128 B with the 16-byte chunk index XOR-ed by 4 ((key >> 1) & 1) - conflict
//   free), and the rescaling of O is lane-local too - no LDS broadcast of row statistics.
// Softmax runs with a FIXED per-query shift m0 (the first tile's maximum, folded into the MFMA chain as its initial
// accumulator: no subtraction per score): exp2(s - m0) cannot overflow unless a later score exceeds m0 by > 64 - a guard
// (one max per two scores, lane-local) then rescales O, the row sum and the shift; that branch is taken ~never for
// LayerNorm-ed ViT activations, but it keeps the kernel exact for any input.  VALU work per score: exp2, add, half a
// convert, half a max - the kernel is VALU-bound at head dim 64 (one exp2 per 256 FLOP), so this count IS its speed.
constexpr int AQ = 128, AKT = 64, KLD = 72;

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 ld_tr2(const char *p0, const char *p1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, c);
}

__global__ __launch_bounds__(256, 3) void attn_kernel(const bf16 *__restrict__ q, const bf16 *__restrict__ k, const bf16 *__restrict__ v,
                                                      bf16 *__restrict__ o, int T, int n_groups) {
    __shared__ __attribute__((aligned(16))) char att_smem[2 * AKT * KLD * 2 + 2 * AKT * VHD * 2];     // K ring 18 KB + V ring 16 KB
    bf16 *Ks = reinterpret_cast<bf16 *>(att_smem);
    char *Vs = att_smem + 2 * AKT * KLD * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    // XCD-aware order (a speed choice only): the query tiles of one (frame, head) stream the SAME K / V - 202 KB that the 7
    // tiles of a 789-token frame re-read.  Blocks b and b + 8 share an XCD (round-robin dispatch), so (frame, head) group g
    // goes to XCD g % 8 and its tiles take consecutive slots there: K / V come from that XCD's L2 after the first tile
    // (measured before: 582 MB fetched per layer and 64 frames for 116 MB of q, k, v).
    const int qtiles = (T + AQ - 1) / AQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = (slot / qtiles) * 8 + xcd;
    if (grp >= n_groups) return;
    const int qt = slot - (slot / qtiles) * qtiles;
    const int head = grp % VH;
    const long long f = grp / VH;
    const long long bh = (f * VH + head) * (long long)T;
    const int i0 = qt * AQ + wave * 32;
    const int qi = min(i0 + r, T - 1);

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) qf[ks] = *reinterpret_cast<const bf16x8 *>(q + (bh + qi) * VHD + ks * 16 + 8 * h);

    // staging: thread (row = tid >> 3, c8 = tid & 7) moves the 16-byte piece c8 of key rows `row` and `row + 32` of K and V.  A
    // ds_write_b128 is served in groups of 8 CONTIGUOUS lanes against 32 banks: here a group stores the 8 pieces of ONE row - 128
    // contiguous bytes, every bank once (four lanes on each of two rows, as before: the rows' 144- / 128-byte strides put them on
    // the same banks - 2-way on every store, a quarter of the kernel's LDS cycles by SQ_LDS_BANK_CONFLICT)
    const int srow = tid >> 3, c8 = tid & 7;
    const int vswz = 4 * ((srow >> 1) & 1);                  // = that of row srow + 32
    u32x4 rk[2], rv[2];
#define A_LOAD(kt)                                                                                   \
    {                                                                                                \
        /* keys beyond T re-read row T - 1 (finite data): their scores are set to -inf below, so P = 0 */   \
        const long long off0 = (bh + min((kt) * AKT + srow, T - 1)) * VHD + c8 * 8;                  \
        const long long off1 = (bh + min((kt) * AKT + 32 + srow, T - 1)) * VHD + c8 * 8;             \
        rk[0] = *reinterpret_cast<const u32x4 *>(k + off0);                                          \
        rk[1] = *reinterpret_cast<const u32x4 *>(k + off1);                                          \
        rv[0] = *reinterpret_cast<const u32x4 *>(v + off0);                                          \
        rv[1] = *reinterpret_cast<const u32x4 *>(v + off1);                                          \
    }
#define A_STORE(buf)                                                                                 \
    {                                                                                                \
        *reinterpret_cast<u32x4 *>(Ks + (buf) * AKT * KLD + srow * KLD + c8 * 8) = rk[0];            \
        *reinterpret_cast<u32x4 *>(Ks + (buf) * AKT * KLD + (32 + srow) * KLD + c8 * 8) = rk[1];     \
        *reinterpret_cast<u32x4 *>(Vs + (buf) * AKT * 128 + srow * 128 + ((c8 ^ vswz) << 4)) = rv[0];        \
        *reinterpret_cast<u32x4 *>(Vs + (buf) * AKT * 128 + (32 + srow) * 128 + ((c8 ^ vswz) << 4)) = rv[1]; \
    }
    // transposing V reads: lane (group G = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3) addresses key row (.. + q4),
    // d = 32 dt + 16 (G & 1) + 4 p4; it receives d = 32 dt + (lane & 31), keys (.. + 0..3)
    int voff[2];
    {
        const int q4 = (lane >> 2) & 3, p4 = lane & 3, dl = 16 * ((lane >> 4) & 1) + 4 * p4, sw = 4 * ((q4 >> 1) & 1);
#pragma unroll
        for (int dt = 0; dt < 2; dt++) voff[dt] = (4 * h + q4) * 128 + ((((dl + 32 * dt) >> 3) ^ sw) << 4) + (dl & 7) * 2;
    }

    f32x16 oacc[2], negm;
#pragma unroll
    for (int e = 0; e < 16; e++) oacc[0][e] = oacc[1][e] = negm[e] = 0.0f;
    float l = 0.0f;

    const int ntile = (T + AKT - 1) / AKT;
    A_LOAD(0);
    A_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < ntile; kt++) {
        if (kt + 1 < ntile) A_LOAD(kt + 1);
        const bf16 *Kb = Ks + (kt & 1) * AKT * KLD;
        const char *Vb = Vs + (kt & 1) * AKT * 128;
        // S^T tiles: rows = keys, cols = queries, in the log2 domain, minus the query's shift (initial accumulator)
        f32x16 st[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            st[j] = negm;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const bf16x8 ka = *reinterpret_cast<const bf16x8 *>(Kb + (j * 32 + r) * KLD + ks * 16 + 8 * h);
                st[j] = mfma_bf16(ka, qf[ks], st[j]);
            }
        }
        if (kt + 1 == ntile) {       // only the last tile has keys beyond T
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (kt * AKT + j * 32 + crow(e, h) >= T) st[j][e] = -INFINITY;
        }
        float mt = st[0][0];           // v_max3 by hand: fmaxf() on MFMA outputs costs two canonicalising v_max each
#pragma unroll
        for (int e = 0; e < 16; e++) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mt) : "v"(st[0][e]), "v"(st[1][e]));
        if (__any(mt > 64.0f) || kt == 0) {
            // (re)centre this query: lanes r and r + 32 hold the two key halves of the same query and must agree
            const float d = fmaxf(mt, __shfl_xor(mt, 32));
            const float a = __builtin_amdgcn_exp2f(-d);
            l *= a;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                oacc[0][e] *= a;
                oacc[1][e] *= a;
                negm[e] -= d;
                st[0][e] -= d;
                st[1][e] -= d;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f(st[j][e]);
                st[j][e] = p;
                l += p;
            }
        // O^T += V^T . P^T: P^T registers 8 s2 .. 8 s2 + 7 of tile j are the B fragment of k-step s2; its element e is key
        // 32 j + 16 s2 + 8 (e >> 2) + 4 h + (e & 3), which is exactly what two transposing reads of 4 key rows deliver
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                bf16x8 pb;
#pragma unroll
                for (int e = 0; e < 8; e++) pb[e] = (bf16)st[j][8 * s2 + e];
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const char *vp = Vb + (32 * j + 16 * s2) * 128 + voff[dt];
                    oacc[dt] = mfma_bf16(ld_tr2(vp, vp + 8 * 128), pb, oacc[dt]);
                }
            }
        if (kt + 1 < ntile) A_STORE((kt + 1) & 1);
        __syncthreads();
    }
#undef A_LOAD
#undef A_STORE
    // normalise, transpose through LDS (wave-private 32 x 144-byte tile in the K ring: every wave is past its last read) and
    // write (B, T, 384) bf16 as full 128-byte lines
    const float inv = 1.0f / (l + __shfl_xor(l, 32));
    char *stg = att_smem + wave * 32 * 144;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            uint2 ov;
            ov.x = pack_bf16x2(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv);
            ov.y = pack_bf16x2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
            *reinterpret_cast<uint2 *>(stg + r * 144 + 64 * dt + 16 * g + 8 * h) = ov;
        }
    {
        const int q8 = lane >> 3, p8 = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const u32x4 val = *reinterpret_cast<const u32x4 *>(stg + (8 * i + q8) * 144 + 16 * p8);
            const int row = i0 + 8 * i + q8;
            if (row < T) *reinterpret_cast<u32x4 *>(o + ((long long)f * T + row) * VD + head * VHD + 8 * p8) = val;
        }
    }
}

int g_rt_dbg = 0;
// row-tile GEMM launch: one workgroup per (128-row tile, half of the 192-column tiles)
template <int KC, int NTP, class Pro, class Epi>
void launch_rt(Pro pro, const bf16 *Wp, long long M, int N, Epi epi, hipStream_t st) {
    const int parts = N / RT_NT / NTP, n_tiles = (int)((M + RT_BM - 1) / RT_BM);
    hipLaunchKernelGGL((gemm_rt_kernel<KC, NTP, Pro, Epi>), dim3((n_tiles + 7) / 8 * 8 * parts), dim3(256),
                       RT_LDS_BYTES + 4 * (Pro::VEC + RT_NT * NTP + epi.extra_floats()), st, pro, Wp, (int)M, parts, epi, g_rt_dbg);
}

size_t ws_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int sslam_vit_pack_linear_host(const float *w, int n_out, int k_in, uint16_t *out) {
    if (!w || !out || n_out <= 0 || k_in <= 0) return SSLAM_E_INVALID;
    if (n_out % RT_NT || k_in % 384) return SSLAM_E_UNSUPPORTED;
    const int ksteps = k_in / 16;
    for (int n = 0; n < n_out; n++)
        for (int k = 0; k < k_in; k++) {
            const float f = w[(size_t)n * k_in + k];
            uint32_t u;
            memcpy(&u, &f, 4);
            const uint16_t b = (u & 0x7fffffffu) > 0x7f800000u ? (uint16_t)((u >> 16) | 0x40) : (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            const size_t frag = ((size_t)(n / RT_NT) * ksteps + k / 16) * RT_SL + (n % RT_NT) / 32;
            out[(frag * 64 + ((k % 16) / 8) * 32 + n % 32) * 8 + k % 8] = b;      // round-to-nearest-even bf16, NaN kept
        }
    return SSLAM_OK;
}

// fused-MLP weight stream: per 64-wide hidden chunk c the 48 KB of up_proj rows [64 c, 64 c + 64) as [k-step 24][slice 2][lane][8]
// followed by the 48 KB of down_proj columns [64 c, 64 c + 64) as [k-step 4][out slice 12][lane][8] with the k order of an
// accumulator tile used as B operand: element j of lane half h of k-step s is hidden unit 16 s + 8 (j >> 2) + 4 h + (j & 3).
// row_scale (384 floats or NULL) multiplies the down_proj rows (LayerScale).
extern "C" int sslam_vit_pack_mlp_host(const float *w_up, const float *w_down, const float *row_scale, uint16_t *out) {
    if (!w_up || !w_down || !out) return SSLAM_E_INVALID;
    auto bf = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        return (u & 0x7fffffffu) > 0x7f800000u ? (uint16_t)((u >> 16) | 0x40) : (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    for (int c = 0; c < MF_CHUNKS; c++) {
        uint16_t *up = out + (size_t)c * 2 * (MF_GROUP_BYTES / 2), *dn = up + MF_GROUP_BYTES / 2;
        for (int ks = 0; ks < 24; ks++)
            for (int sl = 0; sl < 2; sl++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        const int hid = 64 * c + 32 * sl + (l & 31), k = 16 * ks + 8 * (l >> 5) + j;
                        up[((ks * 2 + sl) * 64 + l) * 8 + j] = bf(w_up[(size_t)hid * VD + k]);
                    }
        for (int s4 = 0; s4 < 4; s4++)
            for (int sl = 0; sl < 12; sl++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        const int n = 32 * sl + (l & 31), hid = 64 * c + 16 * s4 + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                        dn[((s4 * 12 + sl) * 64 + l) * 8 + j] = bf(w_down[(size_t)n * VMLP + hid] * (row_scale ? row_scale[n] : 1.0f));
                    }
    }
    return SSLAM_OK;
}

#ifdef SSLAM_RT_PROBE
extern "C" int sslam_probe_vit(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rt_probe), sizeof(unsigned long long) * 8 * 2048) == hipSuccess ? 0 : -3;
}
#endif

extern "C" long long sslam_vit_workspace_bytes(int n_frames, int size) {
    if (n_frames <= 0 || size <= 0 || size % VPATCH) return SSLAM_E_INVALID;
    const long long G = size / VPATCH, T = G * G + VPREFIX, rows = (long long)n_frames * T;
    size_t b = 0;
    b += ws_align(rows * VD * 4);                 // x   fp32 residual stream
    b += ws_align(rows * VD * 2);                 // y   bf16 LN output / attention output
    b += ws_align(rows * VD * 2 * 3);             // q, k, v bf16
    b += ws_align(rows * VMLP * 2);               // h   bf16 MLP hidden (also the patch matrix)
    b += ws_align(rows * 16);                     // LayerNorm partial sums: (sum, sumsq) x 2 column halves per row
    return (long long)b;
}

// images_chw (fp32 planar; patches == nullptr) or patches (bf16 rows of 768 per patch, from sslam_preprocess_u8_patches)
static int vit_forward_impl(const float *images_chw, const bf16 *patches, int n_frames, int size, const sslam_vit_weights_t *w,
                            void *workspace, long long workspace_bytes, float *tokens_out, void *stream) {
    if ((!images_chw && !patches) || !w || !workspace || !tokens_out || n_frames <= 0 || size <= 0 || size % VPATCH) return SSLAM_E_INVALID;
    if (workspace_bytes < sslam_vit_workspace_bytes(n_frames, size)) return SSLAM_E_INVALID;
    if (((uintptr_t)images_chw | (uintptr_t)patches | (uintptr_t)workspace | (uintptr_t)tokens_out) & 15) return SSLAM_E_INVALID;
    const int G = size / VPATCH, cells = G * G, T = cells + VPREFIX;
    const long long rows = (long long)n_frames * T, prow = (long long)n_frames * cells;
    if (rows * VMLP > 0x7fffffffLL * 64) return SSLAM_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#ifdef SSLAM_RT_PROBE
    { const char *e = getenv("SSLAM_RT_DBG"); g_rt_dbg = e ? atoi(e) : 0; }     // probe builds only (tools/rt_probe.py)
#endif
    char *p = (char *)workspace;
    float *x = (float *)p;            p += ws_align(rows * VD * 4);
    bf16 *y = (bf16 *)p;              p += ws_align(rows * VD * 2);
    bf16 *q = (bf16 *)p;              bf16 *k = q + rows * VD, *v = k + rows * VD;   p += ws_align(rows * VD * 2 * 3);
    bf16 *hbuf = (bf16 *)p;          p += ws_align(rows * VMLP * 2);
    float2 *stats = (float2 *)p;

    // patch embedding + prefix tokens
    {
        if (!patches) {
            const long long items = prow * 96;
            hipLaunchKernelGGL(im2patch_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, images_chw, size, items, hbuf);
            sslam_count_launches(1);
        }
        launch_rt<2, 1>(ProBf16{patches ? patches : hbuf, 768}, (const bf16 *)w->patch_w, prow, VD, EpiPatch{w->patch_b, x, stats, cells, T}, st);
        sslam_count_launches(1);
        hipLaunchKernelGGL(prefix_rows_kernel, dim3(n_frames * VPREFIX), dim3(64), 0, st, w->prefix, T, x, stats);
        sslam_count_launches(1);
    }
    const float4 *st4 = (const float4 *)stats;
    const bool small = (rows + RT_BM - 1) / RT_BM * 4 <= 256;
    const bool no_fused = sslam_knob(KNOB_VIT_NO_FUSED_MLP, 0) != 0;      // test-only A/B knob: the two-launch MLP     // <= 8 frames at 448 x 448: latency-shaped launches
    int rt_stop = 0;
#ifdef SSLAM_RT_PROBE
    rt_stop = (int)sslam_knob(KNOB_RT_STOP, 0);
#endif
    for (int L = 0; L < VLAYERS; L++) {
        const sslam_vit_layer_t &ly = w->layer[L];
        // few frames: one 192-column tile per workgroup (6 / 8 column parts instead of 2), so that a frame's 7 row tiles
        // spread over 42 / 56 CUs and a workgroup's serial chain is 6 groups instead of 18 / 24
        if (small)
            launch_rt<1, 1>(ProLN{x, st4, ly.ln1_g, ly.ln1_b, 1e-5f}, (const bf16 *)ly.wqkv, rows, 3 * VD, EpiQKV{ly.bqkv, w->rope_cos, w->rope_sin, q, k, v, T, G}, st);
        else
            launch_rt<1, 3>(ProLN{x, st4, ly.ln1_g, ly.ln1_b, 1e-5f}, (const bf16 *)ly.wqkv, rows, 3 * VD, EpiQKV{ly.bqkv, w->rope_cos, w->rope_sin, q, k, v, T, G}, st);
        if (rt_stop == 1) break;
        hipLaunchKernelGGL(attn_kernel, dim3((unsigned)((n_frames * VH + 7) / 8 * 8 * ((T + AQ - 1) / AQ))), dim3(256), 0, st, q, k, v, y, T, n_frames * VH);
        launch_rt<1, 1>(ProBf16{y, VD}, (const bf16 *)ly.wo, rows, VD, EpiResidual{ly.bo, x, stats}, st);
        if (rt_stop == 2) break;
        if (ly.wmlp && !small && !no_fused) {
            // LN2 -> up -> GELU -> down -> residual in one launch: the hidden activation stays on the chip
            hipLaunchKernelGGL(mlp_fused_kernel, dim3((unsigned)((rows + RT_BM - 1) / RT_BM)), dim3(512), MF_LDS_BYTES, st,
                               ProLN{x, st4, ly.ln2_g, ly.ln2_b, 1e-5f}, (const bf16 *)ly.wmlp, ly.bup, (int)rows, EpiResidual{ly.bdown, x, stats});
            if (rt_stop == 3 || rt_stop == 4) break;
            sslam_count_launches(4);
            continue;
        }
        if (small)
            launch_rt<1, 1>(ProLN{x, st4, ly.ln2_g, ly.ln2_b, 1e-5f}, (const bf16 *)ly.wup, rows, VMLP, EpiGelu{ly.bup, hbuf, VMLP}, st);
        else
            launch_rt<1, 4>(ProLN{x, st4, ly.ln2_g, ly.ln2_b, 1e-5f}, (const bf16 *)ly.wup, rows, VMLP, EpiGelu{ly.bup, hbuf, VMLP}, st);
        if (rt_stop == 3) break;
        launch_rt<4, 1>(ProBf16{hbuf, VMLP}, (const bf16 *)ly.wdown, rows, VD, EpiResidual{ly.bdown, x, stats}, st);
        if (rt_stop == 4) break;
        sslam_count_launches(5);
    }
    const unsigned ln_grid = (unsigned)((rows + 3) / 4);
    hipLaunchKernelGGL(ln_rows_kernel<false>, dim3(ln_grid), dim3(256), 0, st, x, w->norm_g, w->norm_b, 1e-5f, rows, (void *)tokens_out);
    SSLAM_CHECK_LAUNCH();
    return SSLAM_OK;
}

extern "C" int sslam_vit_forward(const float *images_chw, int n_frames, int size, const sslam_vit_weights_t *w, void *workspace,
                                 long long workspace_bytes, float *tokens_out, void *stream) {
    if (!images_chw) return SSLAM_E_INVALID;
    return vit_forward_impl(images_chw, nullptr, n_frames, size, w, workspace, workspace_bytes, tokens_out, stream);
}

extern "C" int sslam_vit_forward_patches(const void *patches_bf16, int n_frames, int size, const sslam_vit_weights_t *w, void *workspace,
                                         long long workspace_bytes, float *tokens_out, void *stream) {
    if (!patches_bf16) return SSLAM_E_INVALID;
    return vit_forward_impl(nullptr, (const bf16 *)patches_bf16, n_frames, size, w, workspace, workspace_bytes, tokens_out, stream);
}

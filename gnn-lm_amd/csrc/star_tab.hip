// Star attention over PQ codes with the centroid table RESIDENT IN LDS, chunk by chunk
// (('ntgt','inter','tgt') edges of layer 1: fn.v_dot_u + edge_softmax + u_mul_e/sum, fairseq/models/hgt.py:354-356,383-385,
//  fused with quant_neighbor_feats[offset] and the look-up half of TorchPQCodec.decode, knn/pq_wrapper.py:189-196).
//
// Why this formulation.  Round 1's swept kernel (attn.hip) decoded every (neighbour, sub-quantizer) pair with a
// 16/32-B gather from the L1: 2 x 16,384 line look-ups per token at ~0.37 lines per clock per CU = 1.1 ms per 8192
// tokens, while HBM (one 128-B code row per neighbour), the matrix pipe (40 %) and the LDS (8 %) idled.  An LDS
// serves the same random 32-B row ~7x faster than the L1's tag pipeline, so here the TABLE goes to LDS instead of
// the decoded features:
//   * the 1024 feature dims are swept in 32-dim chunks; a chunk's sub-tables ((32 / dsub) x 256 rows = 32 KiB,
//     contiguous in the [M][256][dsub] table) arrive by LDS-DMA (global_load_lds_dwordx4, coalesced, L2-resident
//     source, no VGPRs), double-buffered, one barrier per chunk;
//   * FOUR tokens share a workgroup (8 waves: two per token), i.e. one table sweep per pass serves 4 x 128
//     neighbours -- 2 x 1 MiB of L2 -> LDS traffic per 4 tokens instead of the 2 x 16,384 L1 gathers per token;
//   * pass 1, S = X U^T on v_mfma_f32_16x16x4_f32: a lane's A operand IS its look-up.  Lane (r, g) of a 16-neighbour
//     tile reads the 32-B centroid row of (neighbour r, sub-quantizer 4c + g) with two ds_read_b128 and feeds the 8
//     floats to 8 MFMAs (k index g <-> that sub-quantizer's dims).  No decoded slab is ever written or re-read.
//     Odd lane groups read the two halves of the row in the other order so that a ds_read_b128 group spreads over
//     all 16 slots of the bank window; the k <-> dim map is permuted accordingly on the U side (free);
//   * softmax over the neighbours per (token, head);
//   * pass 2, Z = alpha^T X: the B operand of each MFMA is ONE ds_read_b32 straight out of the table,
//     X[j][d] = tab[(m(d), code[j][m(d)], d % dsub)]; the code bytes of a k-step's 4 neighbours come with one
//     ds_read_b32 per (k-step, chunk).  The two waves of a token split the neighbours and meet through 1 KiB of LDS.
// Per token and pass: 2048 MFMAs (the heads fill 8 of the 16 MFMA columns -- 4.2 MFLOP issued for 2.1 useful, the
// floor of this shape on the f32 matrix cores: ~0.44 ms per 8192 tokens for both passes), 16 KiB of random LDS
// reads, 0 L1 gathers.  HBM traffic is unchanged: the code rows (read once per token, staged in LDS), U and Z.
#include "kernels.h"

namespace gnnlm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((address_space(3))) const float lds_cfloat_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x2 lds_cfloat2_t;

namespace {

#ifndef GNNLM_STAB_OFF
#define GNNLM_STAB_OFF 0     // timing-only ablations (wrong results), a bit mask of what to switch OFF:
#endif                       //   1 pass-1 MFMAs, 2 pass-2 MFMAs, 4 bank conflicts (every look-up reads row 0), 8 the table DMA,
                             //   16 code staging, 32 U loads in the sweep, 64 Z stores, 128 the table look-ups themselves
#define STAB_OFF(bit_) ((GNNLM_STAB_OFF & (bit_)) != 0)

constexpr int TPW = 4;                  // tokens per workgroup
constexpr int KGM = 128;                // neighbours per token (padded)
constexpr int HB = 8;                   // heads per launch
constexpr int CD = 32;                  // feature dims per chunk
constexpr int TABF = CD * 256;          // floats per table chunk (32 KiB)
constexpr int SCS = KGM + 4;            // score row stride: the heads of a token land on different banks
constexpr int NTHREADS = 768;           // 8 compute waves (two per token) + 4 loader waves (table DMA only)

struct Carve {
    int tab, sc, zpart, okf, lcodes, total;     // byte offsets
};
__host__ __device__ inline Carve carve(int M) {
    Carve c;
    c.tab = 0;
    c.sc = c.tab + 2 * TABF * 4;
    c.zpart = c.sc + TPW * HB * SCS * 4;
    c.okf = c.zpart + TPW * 2 * 256 * 4;
    c.lcodes = c.okf + TPW * KGM;
    c.total = c.lcodes + ((TPW * KGM * (M + 4) + 15) & ~15);
    return c;
}

template <int DSUB>
__global__ __launch_bounds__(NTHREADS) void star_attn_tab_kernel(StarAttnParams p, int h0) {
    constexpr int MPC = CD / DSUB;              // sub-quantizers per chunk: 4 (dsub 8) or 8 (dsub 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, D = p.D, H = p.H, kg = p.kg, NCH = D / CD;
    const int MS = M + 4;                       // code row stride (bytes): rows of a tile sit on different banks
    const Carve cv = carve(M);
    float* tab = reinterpret_cast<float*>(smem + cv.tab);              // [2][TABF]
    float* sc = reinterpret_cast<float*>(smem + cv.sc);                // [TPW][HB][SCS] scores -> alphas
    // 8 KiB that serve both sweeps: pass 1 [2 buffers][TPW][64 slots of 16 B] the chunk's U rows in MFMA operand order,
    // pass 2 [2 neighbour halves][TPW][HB][32] the chunk's partial sums on their way out
    float* zpart = reinterpret_cast<float*>(smem + cv.zpart);
    unsigned char* okf = smem + cv.okf;                                // [TPW][KGM]
    unsigned char* lcodes = smem + cv.lcodes;                          // [TPW][KGM][MS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = wave >> 1, half = wave & 1;                          // token of this wave, neighbour half
    const int grp = wave >> 2;                                         // ping-pong group (see the sweeps)
    const int n16 = lane & 15, g = lane >> 4;
    const int tok0 = blockIdx.x * TPW;
    const int i_tok = min(tok0 + t, p.T - 1);                          // tail workgroups recompute the last token
    const bool live = tok0 + t < p.T;

    // ---------------------------------------------------------------- table chunk c -> LDS buffer b (32 KiB, linear)
    // The DMA is issued from inline asm on purpose: hipcc treats a __builtin_amdgcn_global_load_lds in flight as a
    // possible writer of EVERY LDS address and puts s_waitcnt vmcnt(0) in front of the table reads of the OTHER
    // buffer (seen in the ISA of a first version of this kernel: the DMA of chunk c + 1 was drained before the first
    // look-up of chunk c).  The asm loads are invisible to its counters; the kernel waits for them itself
    // (STAB_LAND before the chunk barrier).  M0 = LDS byte address of the wave's 1-KiB piece.
    const float* cen = p.centroids;
    const unsigned tab_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)tab);
    // pieces [p0_, p0_ + n_) of chunk c_ (32 pieces of 1 KiB) -> buffer b_
#define STAB_DMA_PIECES(c_, b_, p0_, n_)                                                             \
    {                                                                                                \
        const float* src_ = cen + (int64_t)(c_) * TABF + (p0_) * 256 + lane * 4;                    \
        const unsigned dst_ = tab_lds + (b_) * (TABF * 4) + (p0_) * 1024;                           \
        _Pragma("unroll") for (int q = 0; q < (n_); ++q) {                                          \
            unsigned keep_;                                                                          \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                         : "=&s"(keep_) : "v"(src_ + q * 256), "s"(dst_ + q * 1024) : "memory");    \
        }                                                                                            \
    }
#define STAB_DMA(c_, b_) STAB_DMA_PIECES(c_, b_, (wave - 8) * 8, 8)    /* the four loader waves, 8 pieces each */
    // one DMA instruction: 16 B per lane from src_ (a per-lane pointer) to LDS bytes [dst_, dst_ + 1024) in lane order
#define STAB_DMA_ONE(src_, dst_)                                                                     \
    {                                                                                                \
        unsigned keep_;                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src_), "s"(dst_) : "memory");                              \
    }
    // Loader waves 8..11 own the table DMA.  Issuing a 1-KiB LDS-DMA costs the issuing wave ~100-150 cycles here; with
    // the compute waves issuing their own (4 or 8 per chunk) that was the longest item of their look-up phase (1.76k
    // cycles per phase measured against 1.0k of MFMAs).  Global phase P: A's L(c) = 2c, B's L(c) = 2c + 1; chunk c + 1
    // replaces chunk c - 1 (last read in phase 2c - 1): issued in phase 2c, landed by the end of phase 2c + 1.
#define STAB_LOADER_SWEEP(extra_even_, extra_any_)                                                   \
    for (int P = 0; P <= 2 * NCH; ++P) {                                                             \
        const int c1 = (P >> 1) + 1;                                                                 \
        if (!(P & 1) && c1 < NCH && !STAB_OFF(8)) {                                                  \
            STAB_DMA(c1, c1 & 1)                                                                     \
            extra_even_                                                                              \
        }                                                                                            \
        if (P & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  \
        extra_any_                                                                                   \
        STAB_PHASE()                                                                                 \
    }
// -DGNNLM_STAB_CLK=1: cycle stamps of wave 0 of the first 256 workgroups, written over has_nb (a timing build, wrong has_nb):
// [0] staging, [1] pass 1, [2] softmax, [3] pass 2
#ifndef GNNLM_STAB_CLK
#define GNNLM_STAB_CLK 0
#endif
#ifndef GNNLM_STAB_CLKW
#define GNNLM_STAB_CLKW 0   // the compute wave whose stamps are written
#endif
#if GNNLM_STAB_CLK
#define STAB_CLK() clock64()
#else
#define STAB_CLK() 0ll
#endif
// the sched_barrier keeps the chunk's MFMAs (register-only, free to move for the compiler) ABOVE the wait: the DMA
// of the next chunk then flies under them instead of being waited for first
#define STAB_LAND()                                   \
    {                                                 \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    }
// Phase barrier of the ping-pong: LDS traffic of this wave done, then s_barrier -- and NOTHING else: __syncthreads()
// would also drain the vector-memory counter, i.e. wait in every phase for the U loads / Z stores / DMA that are
// meant to fly across phases (a first ping-pong version did, and every phase cost a memory round trip: 1.42 ms).
// The "memory" clobber keeps the compiler's loads and stores on their side; MFMAs are pinned by the sched_barriers.
#define STAB_PHASE()                                                  \
    {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                            \
    }
// the same with the time of the phase's body (from t_), of the LDS drain and of the barrier accounted (CLK builds)
#if GNNLM_STAB_CLK
#define STAB_PHASE_T(t_, body_, wait_, bar_)                          \
    {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
        const long long a_ = STAB_CLK();                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            \
        const long long b_ = STAB_CLK();                              \
        asm volatile("s_barrier" ::: "memory");                       \
        const long long c_ = STAB_CLK();                              \
        body_ += a_ - (t_); wait_ += b_ - a_; bar_ += c_ - b_; tend = c_; \
        __builtin_amdgcn_sched_barrier(0);                            \
    }
#else
#define STAB_PHASE_T(t_, body_, wait_, bar_) STAB_PHASE()
#endif
    [[maybe_unused]] long long tl[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // pass 1: L body/wait/barrier, M body/wait/barrier; pass 2 likewise
    [[maybe_unused]] long long tend = 0, gap[2] = {0, 0};
    [[maybe_unused]] const long long clk0 = STAB_CLK();

    // ---------------------------------------------------------------- phase 0: validity, code rows (zeros when invalid)
    for (int e = tid; e < TPW * KGM; e += NTHREADS) {
        const int tt = e >> 7, j = e & (KGM - 1);
        const int i = min(tok0 + tt, p.T - 1);
        okf[e] = (j < kg && star_nb_ok(p, i, j, p.ids[(int64_t)i * kg + j])) ? 1 : 0;
    }
    __syncthreads();
    {
        const int per_row = M >> 4;
        for (int e = tid; e < TPW * KGM * per_row; e += NTHREADS) {
            const int row = e / per_row, part = e - row * per_row;
            const int tt = row >> 7, j = row & (KGM - 1);
            const int i = min(tok0 + tt, p.T - 1);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (okf[row] && !STAB_OFF(16)) {
                const int64_t lrow = star_code_row(p, i, j, p.ids[(int64_t)i * kg + j]);
                v = *reinterpret_cast<const uint4*>(p.codes + lrow * M + 16 * part);
            }
            uint32_t* dst = reinterpret_cast<uint32_t*>(lcodes + row * MS + 16 * part);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    if (wave >= 8) {
        // ============================================================ loader waves: the table DMA of both sweeps, and
        // every barrier of the compute path below, in the same order
        // No compute wave issues a vector-memory instruction inside the sweeps: its U operands arrive in LDS and its sums
        // leave through LDS, both moved by these waves.  (With the U loads and Z stores in the compute waves each of
        // them queued behind the table DMA at the texture unit, 64 B per clock per CU, and stalled its wave at issue:
        // switching off the U loads / the Z stores / the DMA saved 102 / 81 / 96 us of 967 per 8192 tokens.)
        const int lw = wave - 8;                                        // U: loader wave lw moves token lw's rows
        const float* usrc;
        {
            const int hh = lane >> 5, ab = (lane >> 4) & 1, dq_ = (lane >> 2) & 3, li_ = lane & 3;
            const int lo_ = DSUB == 8 ? 4 * (dq_ & 1) : 0;
            usrc = p.U + ((int64_t)min(tok0 + lw, p.T - 1) * H + h0 + min(4 * hh + li_, H - 1 - h0)) * D + 8 * dq_ + (ab ? 4 - lo_ : lo_);
        }
        const unsigned u_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)zpart) + lw * 1024;
#define STAB_DMA_U(c_) if (!STAB_OFF(32)) STAB_DMA_ONE(usrc + (c_) * CD, u_lds + ((c_) & 1) * 4096)
        STAB_DMA(0, 0)
        STAB_DMA_U(0)
        STAB_LAND();
        __syncthreads();                // codes staged, chunk 0 landed
        STAB_LOADER_SWEEP(STAB_DMA_U(c1), )
#undef STAB_DMA_U
        __syncthreads();                // end of pass 1
        STAB_DMA(0, 0)
        __syncthreads();                // scores written
        STAB_LAND();
        __syncthreads();                // alphas written, chunk 0 landed
        // Z: in phase P the sums of chunk (P - 2) / 2 of tokens 0, 1 (P even) or of chunk (P - 3) / 2 of tokens 2, 3 (P odd)
        // are complete (written one phase earlier): 128 lanes add the two neighbour halves and store 16 B each
        const int ze = lw * 32 + (lane & 31);                           // (token of the pair, head, 4-dim group)
        const int zt = ze >> 6, zh = (ze >> 3) & 7, zq = ze & 7;
#define STAB_ZOUT(tt_, c_)                                                                           \
    if (lane < 32 && tok0 + (tt_) < p.T && h0 + zh < H && !STAB_OFF(64)) {                           \
        const float* z_ = zpart + ((tt_) * HB + zh) * CD + 4 * zq;                                   \
        const float4 a_ = *reinterpret_cast<const float4*>(z_);                                      \
        const float4 b_ = *reinterpret_cast<const float4*>(z_ + TPW * HB * CD);                      \
        *reinterpret_cast<float4*>(p.Z + ((int64_t)(tok0 + (tt_)) * H + h0 + zh) * D + (c_) * CD + 4 * zq) = \
            make_float4(a_.x + b_.x, a_.y + b_.y, a_.z + b_.z, a_.w + b_.w);                         \
    }
        STAB_LOADER_SWEEP(, if (P >= 2) STAB_ZOUT(2 * (P & 1) + zt, (P - 2 - (P & 1)) >> 1))
        STAB_ZOUT(2 + zt, NCH - 1)
#undef STAB_ZOUT
        return;
    }

    // Both sweeps run as a PING-PONG between the two halves of the workgroup.  Waves w and w + 4 share a SIMD (the
    // dispatcher deals a workgroup's waves round the SIMDs; measured with tools/probes/mfma_16x16x4.hip), group A =
    // waves 0..3 (tokens 0, 1), group B = waves 4..7 (tokens 2, 3).  Every chunk c is two phases per group,
    //     L(c): the look-ups of chunk c into registers, barrier
    //     M(c): the 32 MFMAs of chunk c, barrier
    // and B runs one phase behind A, so on every SIMD one wave is on the matrix pipe while its partner is on the LDS /
    // DMA path.  Why: with all eight waves in the same phase (first versions of this kernel, 1.0 ms per 8192 tokens)
    // the burst of look-ups and DMA issues of a step and its MFMAs never overlapped -- in-kernel cycle stamps
    // (-DGNNLM_STAB_CLK=1) showed ~1000 cycles of look-up latency + ~1000 cycles of MFMAs at full rate + ~1400 cycles
    // waiting for the slower waves per step, against 2048 cycles of MFMA issue per SIMD; software pipelining inside a
    // wave cannot fix that (in-order issue, 15 LDS operations in flight per wave).
    // Table buffer c & 1 is read in L(c): by A in phase 2c, by B in phase 2c + 1; the loader waves refill it with chunk
    // c + 2 in phases 2c + 2 / 2c + 3 (STAB_LOADER_SWEEP).
    [[maybe_unused]] long long clk1 = 0, clk2 = 0, clk3 = 0;

    // ================================================================ pass 1: S[128 nb x 16 (8 real) heads] = X U^T
    {
        // lane (n16, g) carries U[head n16 & 7][chunk dims of k slot g]: the 8 dims of sub-quantizer 4c + g (dsub 8; odd
        // g with the two halves swapped, see the header) or of the pair 8c + 2g, 8c + 2g + 1 (dsub 4)
        const int li = lane & 3, dq = (lane >> 2) & 3, ng = lane >> 4;
        // U[head 4 hh + li][the 8 dims of sub-quantizer dq] as two float4 (halves in this lane's look-up order), from LDS
        const float* ul = zpart + t * 256 + (dq * 4 + li) * 4;
        const int lo = DSUB == 8 ? 4 * (dq & 1) : 0, hi = 4 - lo;
        f32x4 acc[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][0] = acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // code bytes of this lane: neighbour 64 half + 16 q + 4 ng + li, sub-quantizer(s) dq of the chunk (padding rows are zeros)
        const unsigned char* crow = lcodes + (t * KGM + 64 * half + 4 * ng + li) * MS + (DSUB == 8 ? dq : 2 * dq);
        unsigned code[4];
        float4 xa[4], xb[4];
        float4 ua[2], ub[2];
        STAB_LAND();
        __syncthreads();                                               // codes staged, chunk 0 landed
        clk1 = STAB_CLK();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            code[q] = DSUB == 8 ? (unsigned)crow[q * 16 * MS] : (unsigned)*reinterpret_cast<const unsigned short*>(crow + q * 16 * MS);
        if (grp) STAB_PHASE()                                          // B starts one phase late
        for (int c = 0; c < NCH; ++c) {
            // ---------------- L(c)
            {
                [[maybe_unused]] const long long tL = STAB_CLK();
                if (GNNLM_STAB_CLK && c > 0) gap[0] += tL - tend;
                // The look-up phase runs at raised priority: its VALU / LDS instructions compete for issue slots with the
                // partner wave's MFMA stream, and the younger group (waves 4..7) loses that arbitration at equal priority
                // (its look-up phases measured 1.6-2.0k cycles against 0.5-1.1k for the older group's).
                __builtin_amdgcn_s_setprio(1);
                const float* tb = tab + (c & 1) * TABF;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                          // the look-ups ARE the A operands
                    if (STAB_OFF(128)) {
                        xa[q] = make_float4(1.f, 2.f, 3.f, (float)code[q]);
                        xb[q] = xa[q];
                    } else if constexpr (DSUB == 8) {
                        const float* r_ = tb + (dq * 256 + (STAB_OFF(4) ? 0u : code[q])) * 8;
                        xa[q] = *reinterpret_cast<const float4*>(r_ + lo);
                        xb[q] = *reinterpret_cast<const float4*>(r_ + hi);
                    } else {
                        xa[q] = *reinterpret_cast<const float4*>(tb + ((2 * dq) * 256 + (STAB_OFF(4) ? 0u : (code[q] & 255u))) * 4);
                        xb[q] = *reinterpret_cast<const float4*>(tb + ((2 * dq + 1) * 256 + (STAB_OFF(4) ? 0u : (code[q] >> 8))) * 4);
                    }
                }
                {
                    const float* u_ = ul + (c & 1) * 1024;
                    ua[0] = *reinterpret_cast<const float4*>(u_);       ub[0] = *reinterpret_cast<const float4*>(u_ + 64);
                    ua[1] = *reinterpret_cast<const float4*>(u_ + 128); ub[1] = *reinterpret_cast<const float4*>(u_ + 192);
                }
                if (c + 1 < NCH) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        code[q] = DSUB == 8 ? (unsigned)crow[q * 16 * MS + MPC * (c + 1)]
                                            : (unsigned)*reinterpret_cast<const unsigned short*>(crow + q * 16 * MS + MPC * (c + 1));
                }
                __builtin_amdgcn_s_setprio(0);
                STAB_PHASE_T(tL, tl[0], tl[1], tl[2])
            }
            [[maybe_unused]] const long long tM = STAB_CLK();
            if (GNNLM_STAB_CLK) gap[1] += tM - tend;
            // ---------------- M(c)
            if (!STAB_OFF(1)) {
#define STAB_P1_STEP(xv_, uv_, e_)                                                                                   \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                  \
        acc[q][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv_[q].e_, uv_[0].e_, acc[q][0], 0, 0, 0);                    \
        acc[q][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv_[q].e_, uv_[1].e_, acc[q][1], 0, 0, 0);                    \
    }
                STAB_P1_STEP(xa, ua, x) STAB_P1_STEP(xa, ua, y) STAB_P1_STEP(xa, ua, z) STAB_P1_STEP(xa, ua, w)
                STAB_P1_STEP(xb, ub, x) STAB_P1_STEP(xb, ub, y) STAB_P1_STEP(xb, ub, z) STAB_P1_STEP(xb, ub, w)
#undef STAB_P1_STEP
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    asm volatile("" :: "v"(xa[q].x), "v"(xa[q].w), "v"(xb[q].x), "v"(xb[q].w), "v"(ua[0].x), "v"(ub[1].w));
            }
            STAB_PHASE_T(tM, tl[3], tl[4], tl[5])
        }
        if (!grp) STAB_PHASE()                                         // A finished one phase early
        __syncthreads();
        clk2 = STAB_CLK();
        // C layout of the 16 blocks: acc[q][hh][rr] of lane (ng, dq, li) = the part of S[neighbour 64 half + 16 q + 4 ng + rr]
        // [head 4 hh + li] that comes from the dims of sub-quantizer dq: the four dq lanes meet, dq = 0 writes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t okw = *reinterpret_cast<const uint32_t*>(okf + t * KGM + 64 * half + 16 * q + 4 * ng);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float v = acc[q][hh][rr];
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 8);
                    const int j = 64 * half + 16 * q + 4 * ng + rr;
                    if (dq == 0) sc[(t * HB + 4 * hh + li) * SCS + j] = ((okw >> (8 * rr)) & 1u) ? v : -INFINITY;
                }
            }
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- softmax over the neighbours: wave (t, half) -> heads 4 half ..
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float* row = sc + (t * HB + 4 * half + q) * SCS;
        const float v0 = row[lane], v1 = row[64 + lane];
        const float mx = wave_max(fmaxf(v0, v1));
        const float e0 = v0 == -INFINITY ? 0.f : expf(v0 - mx), e1 = v1 == -INFINITY ? 0.f : expf(v1 - mx);
        const float sum = wave_sum(e0 + e1);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        row[lane] = e0 * inv;
        row[64 + lane] = e1 * inv;
        if (q == 0 && half == 0 && h0 == 0 && lane == 0 && p.has_nb && live && !GNNLM_STAB_CLK) p.has_nb[i_tok] = sum > 0.f ? 1.f : 0.f;
    }
    STAB_LAND();
    __syncthreads();                    // alphas written, chunk 0 landed
    clk3 = STAB_CLK();

    // ================================================================ pass 2: Z[16 (8 real) heads x 32 dims] = alpha^T X per chunk
    {
        const int kh = half;
        // k step ks, lane group g  <->  neighbour j = 64 kh + 4 ks + g (alpha = 0 and code row = zeros for padding)
        float a_reg[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) a_reg[ks] = n16 < HB ? sc[(t * HB + n16) * SCS + 64 * kh + 4 * ks + g] : 0.f;
        // The two 16-column MFMA tiles of a chunk take the EVEN and the ODD dims: column n of tile ct is dim
        // DSUB * mloc + 2 * dp + ct with mloc = n / (DSUB / 2), dp = n % (DSUB / 2).  A lane's two B operands are then
        // neighbours in one centroid row: ONE ds_read_b64 and one code byte per k step feed both MFMAs (a first version
        // gave tile ct the dims 16 ct .. 16 ct + 15: two ds_read_b32 from two rows, twice the look-ups and twice the
        // address arithmetic; the LDS, not the matrix pipe, set the pace of this pass).
        constexpr unsigned ROWSH = DSUB == 8 ? 5 : 4;      // log2 of a centroid row in bytes
        const int mloc = n16 / (DSUB / 2), dp = n16 % (DSUB / 2);
        const unsigned lbase = (unsigned)(uintptr_t)(lds_void_t*)tab + (mloc * 256 * DSUB + 2 * dp) * 4;   // (row 0, dim pair) in buffer 0
        const unsigned shift = 8 * (mloc & 3);
        const int widx = mloc >> 2;
        const unsigned char* cbase = lcodes + (t * KGM + 64 * kh + g) * MS;
        float* zb = zpart + ((kh * TPW + t) * HB + 4 * g) * CD + 2 * n16;   // this lane's dim pair of heads 4 g .. 4 g + 3
        uint32_t w[16];                 // code words of this lane's 16 neighbours for the chunk whose look-ups come next
        f32x2 b[16];
        f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) w[ks] = reinterpret_cast<const uint32_t*>(cbase + 4 * ks * MS)[widx];
        // (parking the chunk's 8 x 32 sums in LDS to leave with one 16-B store per lane instead of four 8-B stores
        //  measured slower: 982 vs 962 us per 8192 tokens)
        if (grp) STAB_PHASE()                                          // B starts one phase late
        for (int c = 0; c < NCH; ++c) {
            // ---------------- L(c): look-ups of chunk c; the partial sums of chunk c - 1 meet and leave
            [[maybe_unused]] const long long tL = STAB_CLK();
            __builtin_amdgcn_s_setprio(1);
            {
                const unsigned sb = lbase + (c & 1) * (TABF * 4);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {                      // the look-ups ARE the B operands
                    const unsigned cc = STAB_OFF(4) ? 0u : __builtin_amdgcn_ubfe(w[ks], shift, 8u);
                    if (STAB_OFF(128)) b[ks] = f32x2{(float)cc, 1.f};
                    else b[ks] = *(lds_cfloat2_t*)(uintptr_t)((cc << ROWSH) + sb);
                }
                if (c + 1 < NCH) {
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks)
                        w[ks] = reinterpret_cast<const uint32_t*>(cbase + 4 * ks * MS + MPC * (c + 1))[widx];
                }
                __builtin_amdgcn_s_setprio(0);
                STAB_PHASE_T(tL, tl[6], tl[7], tl[8])
            }
            [[maybe_unused]] const long long tM = STAB_CLK();
            // ---------------- M(c): four accumulator chains -- in its MFMA phase a wave has the matrix pipe to itself, and a
            // dependent 16x16x4 MFMA can only issue 40 cycles after its predecessor (32 for an independent one): with two
            // chains every MFMA waited 8 cycles (1.6k cycles per phase measured against 1.2k in pass 1)
            z0 = f32x4{0.f, 0.f, 0.f, 0.f};
            z1 = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 16; ks += 2) {
                    if (!STAB_OFF(2)) {
                        z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[ks], b[ks].x, z0, 0, 0, 0);
                        z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[ks], b[ks].y, z1, 0, 0, 0);
                        y0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[ks + 1], b[ks + 1].x, y0, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[ks + 1], b[ks + 1].y, y1, 0, 0, 0);
                    } else {
                        asm volatile("" :: "v"(b[ks].x), "v"(b[ks].y), "v"(b[ks + 1].x), "v"(b[ks + 1].y));
                    }
                }
                z0 += y0;
                z1 += y1;
            }
            // C layout: z<ct>[rr] = Z[head 4 g + rr][dim 2 n16 + ct]; heads 8..15 (g >= 2) are padding.  The loader waves add
            // the two neighbour halves and store them in the next phase (see STAB_ZOUT).
            if (g < 2) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) *reinterpret_cast<f32x2*>(zb + CD * rr) = f32x2{z0[rr], z1[rr]};
            }
            STAB_PHASE_T(tM, tl[9], tl[10], tl[11])
        }
        if (!grp) STAB_PHASE()                                         // A finished one phase early
    }
#if GNNLM_STAB_CLK
    if (blockIdx.x < 256 && tid == 64 * GNNLM_STAB_CLKW && p.has_nb) {   // T >= 4096
        const long long clk4 = STAB_CLK();
        unsigned* o = reinterpret_cast<unsigned*>(p.has_nb) + blockIdx.x * 16;
        o[0] = (unsigned)(clk1 - clk0); o[1] = (unsigned)(clk2 - clk1); o[2] = (unsigned)(clk3 - clk2); o[3] = (unsigned)(clk4 - clk3);
        for (int e = 0; e < 10; ++e) o[4 + e] = (unsigned)tl[e];
        o[14] = (unsigned)gap[0]; o[15] = (unsigned)gap[1];
    }
#endif
#undef STAB_DMA
#undef STAB_LOADER_SWEEP
#undef STAB_DMA_PIECES
#undef STAB_PHASE
#undef STAB_PHASE_T
#undef STAB_LAND
#undef STAB_CLK
}

}  // namespace

bool star_attn_tab_eligible(const StarAttnParams& p) {
    return p.codes && p.centroids && p.kg <= KGM && (p.dsub == 4 || p.dsub == 8) && p.M % 16 == 0 && p.D % CD == 0 &&
           p.M * p.dsub == p.D && (uintptr_t)p.codes % 16 == 0 && (uintptr_t)p.centroids % 16 == 0 &&
           (uintptr_t)p.U % 16 == 0 && carve(p.M).total <= 160 * 1024;
}

int star_attn_tab(const StarAttnParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(star_attn_tab_eligible(p), "star_attn_tab: shape not supported by the table-resident kernel");
    const int lds_bytes = carve(p.M).total;
    static bool attr_set = false;
    if (!attr_set) {
        GNNLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&star_attn_tab_kernel<8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GNNLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&star_attn_tab_kernel<4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const dim3 grid((unsigned)cdiv(p.T, TPW)), block(NTHREADS);
    for (int h0 = 0; h0 < p.H; h0 += HB) {
        if (p.dsub == 8) hipLaunchKernelGGL((star_attn_tab_kernel<8>), grid, block, lds_bytes, stream, p, h0);
        else hipLaunchKernelGGL((star_attn_tab_kernel<4>), grid, block, lds_bytes, stream, p, h0);
    }
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm

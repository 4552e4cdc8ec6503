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
//     source, no VGPRs), double-buffered;
//   * FOUR tokens share a workgroup (8 compute waves: two per token, 64 neighbours each), i.e. one table sweep per
//     pass serves 4 x 128 neighbours -- 2 x 1 MiB of L2 -> LDS traffic per 4 tokens instead of 2 x 16,384 L1 gathers
//     per token; four more waves only move data (table and U in, Z out);
//   * both contractions run on v_mfma_f32_4x4x1_16B_f32, whose 16 independent 4 x 4 blocks take the 8 heads without
//     padding (on the 16x16x4 shape the heads fill half of a 16-wide operand: twice the matrix time).  A lane's MFMA
//     operands ARE its look-ups -- pass 1 (S = X U^T): the 32-B centroid row of (neighbour, sub-quantizer), two
//     ds_read_b128, eight k steps; pass 2 (Z = alpha^T X): 8 B of the row, one ds_read_b64, two dims x two head
//     groups.  No decoded slab is ever written or re-read.  Blocks that split a sum (sub-quantizers in pass 1,
//     neighbour slots in pass 2) meet by two DPP row rotations per accumulator register;
//   * softmax over the neighbours per (token, head) between the sweeps.
// What sets the time (tools/probes/mfma_mix.hip, tools/star_clk.py): on this part nothing issues in the shadow of an
// f32 MFMA -- a VALU or DS instruction next to a 4x4x1 / 16x16x4 stream costs its own 4-6 cycles of the SIMD, from
// the same wave or from the other one -- so a sweep costs (MFMA cycles) + ~4.4 x (every other instruction of the
// three waves of the SIMD), and the work went into both terms: 128 MFMAs of ~9 cycles per SIMD and chunk, 3 (pass 2)
// or 5 (pass 1) instructions per look-up, no vector-memory instruction in the compute waves.  Per 8192 tokens:
// 962 us (round 2 start, 16x16x4 ping-pong) -> 685 us; MFMA time alone is ~300 us at the ~1.9 GHz the part holds here.
// HBM traffic is unchanged: the code rows (read once per token, staged in LDS), U and Z.
#include "kernels.h"

namespace gnnlm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((address_space(3))) const float lds_cfloat_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x2 lds_cfloat2_t;
typedef __attribute__((address_space(3))) const f32x4 lds_cfloat4_t;

namespace {

#ifndef GNNLM_STAB_OFF
#define GNNLM_STAB_OFF 0     // timing-only ablations (wrong results), a bit mask of what to switch OFF:
#endif                       //   1 pass-1 MFMAs, 2 pass-2 MFMAs, 4 bank conflicts (every look-up reads row 0), 8 the table DMA,
                             //   16 code staging, 32 U loads in the sweep, 64 Z stores, 128 the table look-ups themselves
#define STAB_OFF(bit_) ((GNNLM_STAB_OFF & (bit_)) != 0)
#ifndef GNNLM_STAB_CLK
#define GNNLM_STAB_CLK 0     // 1: cycle stamps of waves 0 and 4 of the first 256 workgroups written over has_nb (tools/star_clk.py)
#endif

constexpr int TPW = 4;                  // tokens per workgroup
constexpr int KGM = 128;                // neighbours per token (padded)
constexpr int HB = 8;                   // heads per launch
constexpr int CD = 32;                  // feature dims per chunk
constexpr int TABF = CD * 256;          // floats per table chunk (32 KiB)
constexpr int SCS = KGM + 4;            // score row stride: the heads of a token land on different banks
constexpr int NTHREADS = 768;           // 8 compute waves (two per token) + 4 loader waves (table and U in, Z out)

struct Carve {
    int tab, sc, ubuf, okf, lcodes, total;      // byte offsets
};
__host__ __device__ inline Carve carve(int M) {
    Carve c;
    c.tab = 0;
    c.sc = c.tab + 2 * TABF * 4;
    c.ubuf = c.sc + TPW * HB * SCS * 4;
    c.okf = c.ubuf + 2 * TPW * 1024;
    c.lcodes = c.okf + TPW * KGM;
    c.total = c.lcodes + TPW * KGM * M;
    return c;
}

// MT: the number of sub-quantizers when known at compile time (piece counts and loop bounds of the staging fold), else 0
template <int DSUB, int MT>
__global__ __launch_bounds__(NTHREADS) void star_attn_tab_kernel(StarAttnParams p, int h0) {
    constexpr int MPC = CD / DSUB;              // sub-quantizers per chunk: 4 (dsub 8) or 8 (dsub 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = MT ? MT : p.M, D = p.D, H = p.H, kg = p.kg, NCH = D / CD;
    const Carve cv = carve(M);
    float* tab = reinterpret_cast<float*>(smem + cv.tab);              // [2][TABF]
    float* sc = reinterpret_cast<float*>(smem + cv.sc);                // pass 1 -> softmax: [TPW][HB][SCS] scores -> alphas
    float* zbuf = sc;                                                  // pass 2: [2 buffers][2 neighbour halves][TPW][HB][CD] sums on their way out
    float* ubuf = reinterpret_cast<float*>(smem + cv.ubuf);            // pass 1: [2 buffers][TPW][64 slots of 16 B] U rows in MFMA operand order
    unsigned char* okf = smem + cv.okf;                                // [TPW][KGM]
    // code rows, PIECE-MAJOR: [M / 16 pieces][TPW * KGM rows][16 B] -- a 16-B piece of 64 consecutive rows is 1 KiB of LDS in
    // row order, i.e. what one LDS-DMA instruction writes (lane = row); word reads of 8 consecutive rows hit 8 different banks
    unsigned char* lcodes = smem + cv.lcodes;
    constexpr int NR = TPW * KGM;
    // byte offset of the 4-B word that holds code column MPC * c_ (+ a lane's column base < 16 - (MPC * c_ & 15)) of row 0
#define STAB_COL(c_) (((MPC * (c_)) >> 4) * (NR * 16) + ((MPC * (c_)) & 15))
    static_assert(2 * 2 * TPW * HB * CD <= TPW * HB * SCS, "the outgoing sums reuse the score rows");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = wave >> 1, half = wave & 1;                          // token of this wave, neighbour half
    const int tok0 = blockIdx.x * TPW;
    const int i_tok = min(tok0 + t, p.T - 1);                          // tail workgroups recompute the last token
    const bool live = tok0 + t < p.T;

    // ---------------------------------------------------------------- LDS-DMA (global_load_lds_dwordx4: 16 B per lane, 1 KiB per
    // instruction, lane order).  Issued from inline asm on purpose: hipcc treats a __builtin_amdgcn_global_load_lds in
    // flight as a possible writer of EVERY LDS address and puts s_waitcnt vmcnt(0) in front of the table reads of the
    // OTHER buffer (seen in the ISA of a first version of this kernel).  The asm loads are invisible to its counters;
    // the loader waves wait for them themselves (vmcnt(0) before the chunk barrier).  M0 = LDS byte address.
#define STAB_DMA_ONE(src_, dst_)                                                                     \
    {                                                                                                \
        unsigned keep_;                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src_), "s"(dst_) : "memory");                              \
    }
#if GNNLM_STAB_CLK
#define STAB_CLK() clock64()
#else
#define STAB_CLK() 0ll
#endif
    // Chunk barrier: LDS traffic of this wave done, then s_barrier -- and nothing else (__syncthreads() would also drain
    // the vector-memory counter).  The "memory" clobber keeps the compiler's loads and stores on their side, the
    // sched_barriers pin the MFMAs (register-only, otherwise free to move).
#define STAB_SYNC()                                                   \
    {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                            \
    }
#define STAB_PIN() __builtin_amdgcn_sched_barrier(0)
    // x += x of the lane 4 / 8 further round its row of 16 lanes (DPP); the operand must not have been written by one of the
    // two preceding instructions
#define STAB_ROR_ADD(x_, n_) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:" #n_ " row_mask:0xf bank_mask:0xf" : "+v"(x_));
    [[maybe_unused]] const long long clk0 = STAB_CLK();

    // ---------------------------------------------------------------- staging: validity, code rows (zeros when invalid)
    // (all twelve waves; the loader waves put the first two table chunks on their way before they join)
    [[maybe_unused]] long long st[4] = {0, 0, 0, 0};
    auto stage_codes = [&]() {
        // One pass: every 16-B piece of a code row re-derives the neighbour's validity from its id (the eight pieces of a row sit
        // in neighbouring lanes: one request), piece 0 records it.  Six pieces per thread and round, the loads of a round in
        // flight together.  (Validity first, a barrier, then the rows one piece after the other: 20k cycles per workgroup.)
        constexpr int RB = 6;
        const int per_row = M >> 4, n_items = TPW * KGM * per_row;
        for (int e0 = 0; e0 < n_items; e0 += RB * NTHREADS) {
            int64_t id[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int e = e0 + r * NTHREADS + tid;
                const int row = e / per_row, tt = row >> 7, j = row & (KGM - 1);
                id[r] = (e < n_items && j < kg) ? p.ids[(int64_t)min(tok0 + tt, p.T - 1) * kg + j] : -1;
            }
            if (GNNLM_STAB_CLK) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[0] = STAB_CLK(); }
            const uint8_t* src[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int e = e0 + r * NTHREADS + tid;
                const int row = e / per_row, part = e - row * per_row, tt = row >> 7, j = row & (KGM - 1);
                const int i = min(tok0 + tt, p.T - 1);
                const bool ok = e < n_items && j < kg && star_nb_ok(p, i, j, id[r]);
                if (e < n_items && part == 0) okf[row] = ok ? 1 : 0;
                src[r] = ok && !STAB_OFF(16) ? star_code_ptr(p, i, j, id[r]) + 16 * part : nullptr;     // (a mapped shard of a peer: the load crosses xGMI)
            }
            uint4 v[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) v[r] = src[r] ? *reinterpret_cast<const uint4*>(src[r]) : make_uint4(0, 0, 0, 0);
            if (GNNLM_STAB_CLK) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[1] = STAB_CLK(); }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int e = e0 + r * NTHREADS + tid;
                if (e < n_items) {
                    const int row = e / per_row, part = e - row * per_row;
                    *reinterpret_cast<uint4*>(lcodes + part * (NR * 16) + row * 16) = v[r];
                }
            }
        }
    };

    // Both sweeps have the same skeleton.  Chunk k of the table lives in buffer k & 1.  Iteration c of a compute wave issues
    // the MFMAs of chunk c from registers, INTERLEAVED with the look-ups of chunk c + 1 into a second register set, then the
    // chunk barrier E(c).  After E(k - 1) nobody reads buffer k & 1 any more (its look-ups were waited for before the
    // barrier): the loader waves refill it with chunk k + 2 during iteration k and wait for the DMA before E(k).
    //   Why not a ping-pong of two wave groups (one on the matrix pipe while its SIMD partner does its look-ups; an earlier
    // version of this file): cycle stamps showed that a wave makes NO progress while its partner on the SIMD streams
    // MFMAs back to back -- the look-up phase of group B started when the last MFMA of group A had issued and vice versa,
    // the per-chunk period was the SUM of both groups' MFMA and look-up times (850-960 us per 8192 tokens).  Within one
    // wave an LDS read or a VALU instruction issues in the shadow of the wave's own previous MFMA, so the overlap has to
    // be built inside each wave.
    //   No compute wave issues a vector-memory instruction inside the sweeps: U arrives in LDS and Z leaves through LDS,
    // both moved by the loader waves (with U loads and Z stores in the compute waves each of them queued behind the table
    // DMA at the texture unit and stalled its wave at issue: 102 + 81 us of 967).
    if (wave >= 8) {
        // ============================================================ loader waves 8..11
        const int lw = wave - 8;
        __builtin_amdgcn_s_setprio(3);     // few instructions, all on the critical path of the next chunk: never queue behind the MFMA streams
        const unsigned tab_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)tab) + lw * 8192;
        const float* tsrc = p.centroids + lw * 2048 + lane * 4;          // wave lw moves pieces 8 lw .. 8 lw + 7 of a chunk
#define STAB_DMA_TAB(c_)                                                                             \
    if (!STAB_OFF(8)) {                                                                              \
        const float* s_ = tsrc + (int64_t)(c_) * TABF;                                               \
        const unsigned d_ = __builtin_amdgcn_readfirstlane(tab_lds + ((c_) & 1) * (TABF * 4));       \
        /* the instruction offset counts for the global AND the LDS address: two M0 values and two address registers */ \
        unsigned keep_;                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                            \
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"              \
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"  \
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"                                                      \
                     "global_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"              \
                     "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %2, off offset:3072\n\t"  \
                     "s_mov_b32 m0, %0"                                                              \
                     : "=&s"(keep_) : "v"(s_), "v"(s_ + 1024), "s"(d_) : "memory", "scc");           \
    }
        // U: wave lw moves token lw's rows of the chunk; lane = slot (hh, ab, dq, li): 16 B of head 4 hh + li, sub-quantizer
        // dq, first (ab = 0) or second half in the order the lane group of dq reads its centroid rows (see pass 1)
        const float* usrc;
        {
            const int hh = lane >> 5, ab = (lane >> 4) & 1, dq_ = (lane >> 2) & 3, li_ = lane & 3;
            const int lo_ = DSUB == 8 ? 4 * (dq_ & 1) : 0;
            usrc = p.U + ((int64_t)min(tok0 + lw, p.T - 1) * H + h0 + min(4 * hh + li_, H - 1 - h0)) * D + 8 * dq_ + (ab ? 4 - lo_ : lo_);
        }
        const unsigned u_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)ubuf) + lw * 1024;
#define STAB_DMA_U(c_) if (!STAB_OFF(32)) STAB_DMA_ONE(usrc + (c_) * CD, __builtin_amdgcn_readfirstlane(u_lds + ((c_) & 1) * 4096))
#define STAB_LAND() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
        STAB_DMA_TAB(0) STAB_DMA_U(0)
        if (NCH > 1) { STAB_DMA_TAB(1) STAB_DMA_U(1) }
        stage_codes();
        STAB_LAND();
        __syncthreads();                // codes staged, chunks 0 and 1 landed
        STAB_SYNC()                     // E(-1)
        [[maybe_unused]] long long lt[6] = {0, 0, 0, 0, 0, 0};      // CLK builds: cycles issuing / landing / at the barrier, per sweep
        // Pass 1 has a second barrier H(k) inside iteration k, behind the look-ups of chunk k + 1: from there on buffer
        // (k + 1) & 1 is free, and chunk k + 3 has until E(k + 1) to arrive -- about 1.5 iterations instead of one.  (Issuing
        // a chunk's nine DMA instructions costs a loader wave 0.4-1.3k cycles next to the MFMA streams, its landing
        // another 1.1k: with issue and landing inside ONE iteration the sweep ran at 2.8k cycles per chunk, DMA-bound.)
        if (NCH > 2) { STAB_DMA_TAB(2) STAB_DMA_U(2) }
        for (int k = 0; k < NCH; ++k) {
            STAB_SYNC()                 // H(k)
            [[maybe_unused]] const long long a_ = STAB_CLK();
            if (k + 3 < NCH) {
                STAB_DMA_TAB(k + 3) STAB_DMA_U(k + 3)
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");    // chunk k + 2 has landed (loads return in order)
            } else {
                STAB_LAND();
            }
            [[maybe_unused]] const long long c_ = STAB_CLK();
            STAB_SYNC()                 // E(k)
            if (GNNLM_STAB_CLK) { lt[1] += c_ - a_; lt[2] += STAB_CLK() - c_; }
        }
        STAB_DMA_TAB(0)
        if (NCH > 1) STAB_DMA_TAB(1)
        __syncthreads();                // scores written
        STAB_LAND();
        __syncthreads();                // alphas written, chunks 0 and 1 landed
        // Z: wave lw adds the two neighbour halves of token lw's sums of a chunk and stores 16 B per lane
        const int zh = lane >> 3, zq = lane & 7;
        const bool zlive = tok0 + lw < p.T && h0 + zh < H && !STAB_OFF(64);
        const float* zsrc = zbuf + (lw * HB + zh) * CD + 4 * zq;
        float* zdst = p.Z + ((int64_t)min(tok0 + lw, p.T - 1) * H + h0 + min(zh, H - 1 - h0)) * D + 4 * zq;
#define STAB_ZOUT(c_)                                                                                \
    if (zlive) {                                                                                     \
        const float* z_ = zsrc + ((c_) & 1) * (2 * TPW * HB * CD);                                   \
        const float4 a_ = *reinterpret_cast<const float4*>(z_);                                      \
        const float4 b_ = *reinterpret_cast<const float4*>(z_ + TPW * HB * CD);                      \
        *reinterpret_cast<float4*>(zdst + (c_) * CD) = make_float4(a_.x + b_.x, a_.y + b_.y, a_.z + b_.z, a_.w + b_.w); \
    }
        STAB_SYNC()                     // E(-1)
        for (int k = 0; k < NCH; ++k) {
            [[maybe_unused]] const long long a_ = STAB_CLK();
            if (k > 0) STAB_ZOUT(k - 1)
            if (k + 2 < NCH) STAB_DMA_TAB(k + 2)
            [[maybe_unused]] const long long b_ = STAB_CLK();
            STAB_LAND();                // (also the Z store: issued first, so its round trip is under the DMA's)
            [[maybe_unused]] const long long c_ = STAB_CLK();
            STAB_SYNC()                 // E(k)
            if (GNNLM_STAB_CLK) { lt[3] += b_ - a_; lt[4] += c_ - b_; lt[5] += STAB_CLK() - c_; }
        }
        STAB_ZOUT(NCH - 1)
#if GNNLM_STAB_CLK
        if (blockIdx.x < 256 && tid == 512 && p.has_nb) {
            unsigned* o = reinterpret_cast<unsigned*>(p.has_nb) + blockIdx.x * 32 + 8;
            for (int e = 0; e < 6; ++e) o[e] = (unsigned)lt[e];
        }
#endif
#undef STAB_ZOUT
#undef STAB_DMA_U
#undef STAB_DMA_TAB
#undef STAB_LAND
        return;
    }

    stage_codes();
    if (GNNLM_STAB_CLK) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st[2] = STAB_CLK(); }
    [[maybe_unused]] long long clk1 = 0, clk2 = 0, clk3 = 0;
    // ================================================================ pass 1: S[128 nb x 8 heads] = X U^T on v_mfma_f32_4x4x1 (16 blocks)
    // Block b = lane / 4 of an MFMA is (neighbour group ng = b / 4, sub-quantizer dq = b % 4 of the chunk); lane li = lane % 4
    // of the block supplies, for k step e, A = X[neighbour 4 ng + li][dim e of sub-quantizer dq] and B = U[head li (+ 4)][same dim]:
    // the block accumulates the 4 x 4 tile S[4 neighbours][4 heads] restricted to its sub-quantizer's dims.  All 16 columns
    // of the operand are real work (on the 16x16x4 shape the 8 heads filled half of the 16 columns: twice the MFMA time).
    // The lane's A operands of a chunk ARE its look-up: the 32-B centroid row of (neighbour, sub-quantizer), two
    // ds_read_b128; odd dq read the two halves in the other order so that a 16-lane read group spreads over all 16 slots
    // of the bank window (the U side is stored by the loader in the same order).
    {
        const int li = lane & 3, dq = (lane >> 2) & 3, ng = lane >> 4;
        const float* ul = ubuf + t * 256 + (dq * 4 + li) * 4;
        const int lo = DSUB == 8 ? 4 * (dq & 1) : 0;
        // LDS byte address of (row 0 of this lane's sub-quantizer, the half it reads first) in buffer 0; the other half is ^ 16
        const unsigned lb1 = (unsigned)(uintptr_t)(lds_void_t*)tab + ((DSUB == 8 ? dq * 256 * 8 : 2 * dq * 256 * 4) + lo) * 4;
        f32x4 acc[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][0] = acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // code bytes of this lane: neighbour 64 half + 16 q + 4 ng + li, sub-quantizer(s) dq of the chunk (padding rows are zeros)
        const unsigned char* crow = lcodes + (t * KGM + 64 * half + 4 * ng + li) * 16 + (DSUB == 8 ? 0 : 4 * (dq >> 1));
        const unsigned cshift = DSUB == 8 ? 8 * dq : 16 * (dq & 1);     // this lane's byte (dsub 8) or byte pair (dsub 4) of the code word
        unsigned code[4];
        f32x4 xa[2][4], xb[2][4], ua[2][2], ub[2][2];
        __syncthreads();                                               // codes staged, chunks 0 and 1 landed
        clk1 = STAB_CLK();
#define STAB_CODES(c_)                                                                               \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                    \
        code[q] = *reinterpret_cast<const uint32_t*>(crow + q * 256 + STAB_COL(c_));
        // look-up of tile q_ of the chunk in table buffer tb_ into register set s_
#define STAB_LOOK(s_, q_, tb_)                                                                       \
    if (STAB_OFF(128)) {                                                                             \
        xa[s_][q_] = f32x4{1.f, 2.f, 3.f, (float)code[q_]}; xb[s_][q_] = xa[s_][q_];           \
    } else if constexpr (DSUB == 8) {                                                                \
        const unsigned a_ = ((STAB_OFF(4) ? 0u : __builtin_amdgcn_ubfe(code[q_], cshift, 8u)) << 5) + (tb_); \
        xa[s_][q_] = *(lds_cfloat4_t*)(uintptr_t)a_;                                                 \
        xb[s_][q_] = *(lds_cfloat4_t*)(uintptr_t)(a_ ^ 16u);                                         \
    } else {                                                                                         \
        xa[s_][q_] = *(lds_cfloat4_t*)(uintptr_t)(((STAB_OFF(4) ? 0u : __builtin_amdgcn_ubfe(code[q_], cshift, 8u)) << 4) + (tb_));                    \
        xb[s_][q_] = *(lds_cfloat4_t*)(uintptr_t)(((STAB_OFF(4) ? 0u : __builtin_amdgcn_ubfe(code[q_], cshift + 8u, 8u)) << 4) + (tb_) + 256 * 4 * 4);  \
    }
#define STAB_ULOOK(s_, u_)                                                                           \
    ua[s_][0] = *reinterpret_cast<const f32x4*>(u_);       ub[s_][0] = *reinterpret_cast<const f32x4*>((u_) + 64);  \
    ua[s_][1] = *reinterpret_cast<const f32x4*>((u_) + 128); ub[s_][1] = *reinterpret_cast<const f32x4*>((u_) + 192);
        // 8 MFMAs: k step e_ of the halves xv_ / uv_ of register set s_, all four tiles, both head groups
#define STAB_P1_STEP(s_, xv_, uv_, e_)                                                               \
    if (!STAB_OFF(1)) {                                                                              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                              \
            acc[q][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv_[s_][q].e_, uv_[s_][0].e_, acc[q][0], 0, 0, 0); \
            acc[q][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv_[s_][q].e_, uv_[s_][1].e_, acc[q][1], 0, 0, 0); \
        }                                                                                            \
    } else {                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(xv_[s_][q].e_), "v"(uv_[s_][0].e_), "v"(uv_[s_][1].e_)); \
    }
        // iteration c_: MFMAs of chunk c_ from set cur_, look-ups of chunk c_ + 1 into set nxt_ (the last iteration looks up a
        // buffer nobody refills: harmless), code bytes of chunk c_ + 2
#define STAB_P1_ITER(c_, cur_, nxt_)                                                                 \
    {                                                                                                \
        const unsigned tb_ = lb1 + (((c_) + 1) & 1) * (TABF * 4);                                    \
        const float* u_ = ul + (((c_) + 1) & 1) * 1024;                                              \
        const int c2_ = min((c_) + 2, NCH - 1);                                                      \
        STAB_P1_STEP(cur_, xa, ua, x) STAB_PIN(); STAB_LOOK(nxt_, 0, tb_) STAB_LOOK(nxt_, 1, tb_) STAB_PIN(); \
        STAB_P1_STEP(cur_, xa, ua, y) STAB_PIN(); STAB_LOOK(nxt_, 2, tb_) STAB_LOOK(nxt_, 3, tb_) STAB_PIN(); \
        STAB_P1_STEP(cur_, xa, ua, z) STAB_PIN(); STAB_ULOOK(nxt_, u_) STAB_PIN();                   \
        STAB_P1_STEP(cur_, xa, ua, w)                                                                \
        STAB_P1_STEP(cur_, xb, ub, x)                                                                \
        STAB_SYNC()                 /* H(c_): nobody reads buffer (c_ + 1) & 1 any more */           \
        STAB_P1_STEP(cur_, xb, ub, y) STAB_PIN(); STAB_CODES(c2_) STAB_PIN();                        \
        STAB_P1_STEP(cur_, xb, ub, z)                                                                \
        STAB_P1_STEP(cur_, xb, ub, w)                                                                \
        STAB_SYNC()                 /* E(c_) */                                                      \
    }
        STAB_CODES(0)
        {
            const unsigned tb_ = lb1;
#pragma unroll
            for (int q = 0; q < 4; ++q) { STAB_LOOK(0, q, tb_) }
            STAB_ULOOK(0, ul)
        }
        STAB_CODES(min(1, NCH - 1))
        STAB_SYNC()                                                    // E(-1)
        for (int c = 0; c < NCH; c += 2) {
            STAB_P1_ITER(c, 0, 1)
            if (c + 1 < NCH) STAB_P1_ITER(c + 1, 1, 0)
        }
#undef STAB_P1_ITER
#undef STAB_P1_STEP
#undef STAB_ULOOK
#undef STAB_LOOK
#undef STAB_CODES
        clk2 = STAB_CLK();
        // C layout of the 16 blocks: acc[q][hh][rr] of lane (ng, dq, li) = the part of S[neighbour 64 half + 16 q + 4 ng + rr]
        // [head 4 hh + li] that comes from the dims of sub-quantizer dq: the four dq lanes meet, dq = 0 writes
        asm volatile("s_nop 7\n\ts_nop 7");
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) STAB_ROR_ADD(acc[q][hh][rr], 4)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) STAB_ROR_ADD(acc[q][hh][rr], 8)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t okw = *reinterpret_cast<const uint32_t*>(okf + t * KGM + 64 * half + 16 * q + 4 * ng);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float v = acc[q][hh][rr];
                    const int j = 64 * half + 16 * q + 4 * ng + rr;
                    if (dq == 0) sc[(t * HB + 4 * hh + li) * SCS + j] = ((okw >> (8 * rr)) & 1u) ? v : -INFINITY;
                }
            }
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- softmax over the neighbours: wave (t, half) -> heads 4 half ..
    {
        float al[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* row = sc + (t * HB + 4 * half + q) * SCS;
            const float v0 = row[lane], v1 = row[64 + lane];
            const float mx = wave_max(fmaxf(v0, v1));
            const float e0 = v0 == -INFINITY ? 0.f : expf(v0 - mx), e1 = v1 == -INFINITY ? 0.f : expf(v1 - mx);
            const float sum = wave_sum(e0 + e1);
            const float inv = sum > 0.f ? 1.f / sum : 0.f;
            al[q][0] = e0 * inv;
            al[q][1] = e1 * inv;
            if (q == 0 && half == 0 && h0 == 0 && lane == 0 && p.has_nb && live && !GNNLM_STAB_CLK) p.has_nb[i_tok] = sum > 0.f ? 1.f : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* row = sc + (t * HB + 4 * half + q) * SCS;
            row[lane] = al[q][0];
            row[64 + lane] = al[q][1];
        }
    }
    __syncthreads();                    // alphas written, chunks 0 and 1 landed
    clk3 = STAB_CLK();

    // ================================================================ pass 2: Z[8 heads x 32 dims] = alpha^T X per chunk, on v_mfma_f32_4x4x1 too
    // Block (dg, ns) = lane / 4 is (sub-quantizer dg of the chunk, neighbour slot ns of a group of four); lane j of the block
    // supplies A = alpha[head j (+ 4)][neighbour 4 grp + ns] and B = X[that neighbour][dim 8 dg + 2 j + e], e = 0, 1: ONE
    // ds_read_b64 out of the centroid row feeds four MFMAs (two dims x two head groups), the B operands ARE the look-ups.
    // A block accumulates Z[4 heads][4 dims] over ITS neighbour slot; the four slots of a (dg, j) meet once per chunk by
    // two DPP row rotations per accumulator register.  All 16 operand columns are real work (on 16x16x4 the 8 heads
    // filled half of the rows: twice the MFMA time, and on this part nothing issues in the shadow of an f32 MFMA --
    // tools/probes/mfma_mix.hip -- so MFMA cycles and instruction count simply add).
    {
        const int kh = half;
        const int j4 = lane & 3, ns = (lane >> 2) & 3, dg = lane >> 4;
        // this lane's sub-quantizer slot in the chunk and byte offset in its centroid row
        const int msub = DSUB == 8 ? dg : 2 * dg + (j4 >> 1);
        const int boff = DSUB == 8 ? 8 * j4 : 8 * (j4 & 1);
        constexpr unsigned ROWSH = DSUB == 8 ? 5 : 4;      // log2 of a centroid row in bytes
        const unsigned lbase = (unsigned)(uintptr_t)(lds_void_t*)tab + msub * 256 * DSUB * 4 + boff;   // row 0 in buffer 0
        // group grp, slot ns  <->  neighbour j = 64 kh + 4 grp + ns (alpha = 0 and code row = zeros for padding)
        float a_reg[16][2];
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) {
            a_reg[gq][0] = sc[(t * HB + j4) * SCS + 64 * kh + 4 * gq + ns];
            a_reg[gq][1] = sc[(t * HB + 4 + j4) * SCS + 64 * kh + 4 * gq + ns];
        }
        const unsigned char* cbase = lcodes + (t * KGM + 64 * kh + ns) * 16 + (msub & ~3);
        const unsigned shift = 8 * (msub & 3);
        float* zb = zbuf + ((kh * TPW + t) * HB + j4) * CD + 8 * dg;   // (head j4, this lane group's 8 dims): see the write below
        uint32_t w[16];                 // code words of this lane's 16 neighbours for the chunk whose look-ups come next
        f32x2 b[2][16];
// (a ds_read_u8 of the lane's own byte would save the v_bfe, but measured 11 % slower on the whole sweep: four lanes reading
// different bytes of one dword do not broadcast)
#define STAB_W(gq_, c_) w[gq_] = *reinterpret_cast<const uint32_t*>(cbase + 64 * (gq_) + STAB_COL(c_));
#define STAB_BLOOK(s_, gq_, sb_)                                                                     \
    {                                                                                                \
        const unsigned cc_ = STAB_OFF(4) ? 0u : __builtin_amdgcn_ubfe(w[gq_], shift, 8u);            \
        if (STAB_OFF(128)) b[s_][gq_] = f32x2{(float)cc_, 1.f};                                      \
        else b[s_][gq_] = *(lds_cfloat2_t*)(uintptr_t)((cc_ << ROWSH) + (sb_));                      \
    }
        // iteration c_: per group the look-up of chunk c_ + 1, the code word of chunk c_ + 2 into the register just consumed,
        // four MFMAs of chunk c_
#define STAB_P2_ITER(c_, cur_, nxt_)                                                                 \
    {                                                                                                \
        const unsigned sb_ = lbase + (((c_) + 1) & 1) * (TABF * 4);                                  \
        const int c2_ = min((c_) + 2, NCH - 1);                                                      \
        f32x4 z[2][2];                                                                               \
        z[0][0] = z[0][1] = z[1][0] = z[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};                           \
        _Pragma("unroll") for (int gq = 0; gq < 16; ++gq) {                                         \
            STAB_BLOOK(nxt_, gq, sb_) STAB_W(gq, c2_)                                                \
            if (!STAB_OFF(2)) {                                                                      \
                z[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_reg[gq][0], b[cur_][gq].x, z[0][0], 0, 0, 0); \
                z[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_reg[gq][1], b[cur_][gq].x, z[0][1], 0, 0, 0); \
                z[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_reg[gq][0], b[cur_][gq].y, z[1][0], 0, 0, 0); \
                z[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_reg[gq][1], b[cur_][gq].y, z[1][1], 0, 0, 0); \
            } else asm volatile("" :: "v"(b[cur_][gq].x), "v"(b[cur_][gq].y));                        \
            STAB_PIN();                                                                              \
        }                                                                                            \
        asm volatile("s_nop 7\n\ts_nop 7");     /* the last MFMAs have written their registers before the DPP adds read them */ \
        /* C layout: z[e][hh][i] of lane (dg, ns, j) = the part of Z[head 4 hh + i][dim 8 dg + 2 j + e] of neighbour slot ns */ \
        /* (a DPP operand must have been written at least two instructions earlier: 16 registers per round) */ \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                \
            _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) STAB_ROR_ADD(z[e][hh][i], 4)           \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                \
            _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) STAB_ROR_ADD(z[e][hh][i], 8)           \
        /* The loader waves add the two neighbour halves and store them during the next iteration. */ \
        if (ns == 0) {                                                                               \
            float* zo_ = zb + ((c_) & 1) * (2 * TPW * HB * CD) + 2 * j4 - j4 * CD;                   \
            _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
                    *reinterpret_cast<f32x2*>(zo_ + (4 * hh + i) * CD) = f32x2{z[0][hh][i], z[1][hh][i]}; \
        }                                                                                            \
        STAB_SYNC()                                                                                  \
    }
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) STAB_W(gq, 0)
        {
            const unsigned sb_ = lbase;
#pragma unroll
            for (int gq = 0; gq < 16; ++gq) { STAB_BLOOK(0, gq, sb_) STAB_W(gq, min(1, NCH - 1)) }
        }
        STAB_SYNC()                                                    // E(-1): everybody has its alphas, the score rows may go
        for (int c = 0; c < NCH; c += 2) {
            STAB_P2_ITER(c, 0, 1)
            if (c + 1 < NCH) STAB_P2_ITER(c + 1, 1, 0)
        }
#undef STAB_P2_ITER
#undef STAB_ROR_ADD
#undef STAB_BLOOK
#undef STAB_W
    }
#if GNNLM_STAB_CLK
    if (blockIdx.x < 256 && (tid == 0 || tid == 256) && p.has_nb) {   // T >= 8192: waves 0 and 4
        const long long clk4 = STAB_CLK();
        unsigned* o = reinterpret_cast<unsigned*>(p.has_nb) + blockIdx.x * 32 + (tid ? 16 : 0);
        o[0] = (unsigned)clk0; o[1] = (unsigned)clk1; o[2] = (unsigned)clk2; o[3] = (unsigned)clk3; o[4] = (unsigned)clk4;
        o[5] = (unsigned)(st[0] - clk0); o[6] = (unsigned)(st[1] - clk0); o[7] = (unsigned)(st[2] - clk0);
    }
#endif
#undef STAB_DMA_ONE
#undef STAB_COL
#undef STAB_SYNC
#undef STAB_PIN
#undef STAB_CLK
}

}  // namespace

bool star_attn_tab_eligible(const StarAttnParams& p) {
    return (p.codes || p.shards) && p.centroids && p.kg <= KGM && (p.dsub == 4 || p.dsub == 8) && p.M % 16 == 0 && p.D % CD == 0 &&
           p.M * p.dsub == p.D && (uintptr_t)p.codes % 16 == 0 && (uintptr_t)p.centroids % 16 == 0 && !(p.shards && p.codes_direct) &&
           (uintptr_t)p.U % 16 == 0 && carve(p.M).total <= 160 * 1024;
}

int star_attn_tab(const StarAttnParams& p, hipStream_t stream) {
    GNNLM_REQUIRE(star_attn_tab_eligible(p), "star_attn_tab: shape not supported by the table-resident kernel");
    const int lds_bytes = carve(p.M).total;
    const auto k8m = &star_attn_tab_kernel<8, 128>, k8 = &star_attn_tab_kernel<8, 0>, k4 = &star_attn_tab_kernel<4, 0>;
    GNNLM_LDS_OPT_IN(k8m, 160 * 1024);
    GNNLM_LDS_OPT_IN(k8, 160 * 1024);
    GNNLM_LDS_OPT_IN(k4, 160 * 1024);
    const dim3 grid((unsigned)cdiv(p.T, TPW)), block(NTHREADS);
    for (int h0 = 0; h0 < p.H; h0 += HB) {
        if (p.dsub == 8 && p.M == 128) hipLaunchKernelGGL(k8m, grid, block, lds_bytes, stream, p, h0);
        else if (p.dsub == 8) hipLaunchKernelGGL(k8, grid, block, lds_bytes, stream, p, h0);
        else hipLaunchKernelGGL(k4, grid, block, lds_bytes, stream, p, h0);
    }
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm

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

namespace {

#ifndef GNNLM_STAB_EXP
#define GNNLM_STAB_EXP 0     // timing-only ablations (wrong results): 1 no pass-1 MFMAs, 2 no pass-2 MFMAs,
#endif                       // 3 every look-up reads row 0 (no bank conflicts), 4 no table DMA after the first chunk,
                             // 5 no code staging (phase 0 loads), 6 no U loads in the sweep, 7 no Z stores, 8 no MFMAs at all

constexpr int TPW = 4;                  // tokens per workgroup
constexpr int KGM = 128;                // neighbours per token (padded)
constexpr int HB = 8;                   // heads per launch
constexpr int CD = 32;                  // feature dims per chunk
constexpr int TABF = CD * 256;          // floats per table chunk (32 KiB)
constexpr int SCS = KGM + 4;            // score row stride: the heads of a token land on different banks
constexpr int NTHREADS = 512;

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
__global__ __launch_bounds__(NTHREADS, 2) void star_attn_tab_kernel(StarAttnParams p, int h0) {
    constexpr int MPC = CD / DSUB;              // sub-quantizers per chunk: 4 (dsub 8) or 8 (dsub 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, D = p.D, H = p.H, kg = p.kg, NCH = D / CD;
    const int MS = M + 4;                       // code row stride (bytes): rows of a tile sit on different banks
    const Carve cv = carve(M);
    float* tab = reinterpret_cast<float*>(smem + cv.tab);              // [2][TABF]
    float* sc = reinterpret_cast<float*>(smem + cv.sc);                // [TPW][HB][SCS] scores -> alphas
    float* zpart = reinterpret_cast<float*>(smem + cv.zpart);          // [TPW][2][2 tiles][4][32]
    unsigned char* okf = smem + cv.okf;                                // [TPW][KGM]
    unsigned char* lcodes = smem + cv.lcodes;                          // [TPW][KGM][MS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = wave >> 1, half = wave & 1;                          // token of this wave, neighbour half
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
    const unsigned tab_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)tab) + wave * 4096;
#define STAB_DMA(c_, b_)                                                                             \
    {                                                                                                \
        const float* src_ = cen + (int64_t)(c_) * TABF + wave * 1024 + lane * 4;                    \
        const unsigned dst_ = tab_lds + (b_) * (TABF * 4);                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
            unsigned keep_;                                                                          \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                         : "=&s"(keep_) : "v"(src_ + q * 256), "s"(dst_ + q * 1024) : "memory");    \
        }                                                                                            \
    }
// the sched_barrier keeps the chunk's MFMAs (register-only, free to move for the compiler) ABOVE the wait: the DMA
// of the next chunk then flies under them instead of being waited for first
#define STAB_LAND()                                   \
    {                                                 \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    }

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
            if (okf[row] && GNNLM_STAB_EXP != 5) {
                const int64_t lrow = star_code_row(p, i, j, p.ids[(int64_t)i * kg + j]);
                v = *reinterpret_cast<const uint4*>(p.codes + lrow * M + 16 * part);
            }
            uint32_t* dst = reinterpret_cast<uint32_t*>(lcodes + row * MS + 16 * part);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    STAB_DMA(0, 0)

    // Both sweeps are software-pipelined over the chunks.  Step c, between two barriers:
    //     DMA(c + 2) -> buffer c & 1          (the look-ups of chunk c completed before the barrier that opened the step)
    //     look-ups of chunk c + 1 -> register set (c + 1) & 1       (its table landed before that barrier)
    //     MFMAs of chunk c from register set c & 1
    //     wait for the DMA, barrier
    // so the LDS latency of a chunk's look-ups and the DMA of the one after hide under a chunk of MFMAs.  (A first
    // version read and multiplied the SAME chunk between two barriers: 1.01 ms per 8192 tokens, of which 0.67 ms
    // remained with every MFMA removed -- each wave may have 15 LDS operations in flight and there are only two
    // waves per SIMD, so the look-up phase was a latency chain with the matrix pipe idle.)  The loops are unrolled
    // by two so that the register sets have static names.

    // ================================================================ pass 1: S[128 nb x 16 (8 real) heads] = X U^T
    {
        // lane (n16, g) carries U[head n16 & 7][chunk dims of k slot g]: the 8 dims of sub-quantizer 4c + g (dsub 8; odd
        // g with the two halves swapped, see the header) or of the pair 8c + 2g, 8c + 2g + 1 (dsub 4)
        const float* Ur = p.U + ((int64_t)i_tok * H + h0 + min(n16 & 7, H - 1 - h0)) * D + 8 * g;
        const int lo = DSUB == 8 ? 4 * (g & 1) : 0, hi = 4 - lo;
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // code bytes of this lane: neighbour 64 half + 16 q + n16, sub-quantizer(s) of k slot g (padding rows are zeros)
        const unsigned char* crow = lcodes + (t * KGM + 64 * half + n16) * MS + (DSUB == 8 ? g : 2 * g);
        unsigned code[4];
        float4 xaA[4], xbA[4], xaB[4], xbB[4], uaA, ubA, uaB, ubB;
#define STAB_CODES(c_)                                                                               \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                   \
        code[q] = DSUB == 8 ? (unsigned)crow[q * 16 * MS + MPC * (c_)]                              \
                            : (unsigned)*reinterpret_cast<const unsigned short*>(crow + q * 16 * MS + MPC * (c_));
#define STAB_LOOK1(c_, S)      /* the look-ups ARE the A operands */                                 \
    {                                                                                                \
        const float* tb_ = tab + ((c_) & 1) * TABF;                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
            if constexpr (DSUB == 8) {                                                               \
                const float* r_ = tb_ + (g * 256 + (GNNLM_STAB_EXP == 3 ? 0u : code[q])) * 8;       \
                xa##S[q] = *reinterpret_cast<const float4*>(r_ + lo);                                \
                xb##S[q] = *reinterpret_cast<const float4*>(r_ + hi);                                \
            } else {                                                                                 \
                xa##S[q] = *reinterpret_cast<const float4*>(tb_ + ((2 * g) * 256 + (GNNLM_STAB_EXP == 3 ? 0u : (code[q] & 255u))) * 4); \
                xb##S[q] = *reinterpret_cast<const float4*>(tb_ + ((2 * g + 1) * 256 + (GNNLM_STAB_EXP == 3 ? 0u : (code[q] >> 8))) * 4); \
            }                                                                                        \
        }                                                                                            \
    }
#define STAB_ULOAD(c_, S)                                                                            \
    if (GNNLM_STAB_EXP != 6 || (c_) == 0) {                                                          \
        ua##S = *reinterpret_cast<const float4*>(Ur + (c_) * CD + lo);                               \
        ub##S = *reinterpret_cast<const float4*>(Ur + (c_) * CD + hi);                               \
    }
#define STAB_MMA1(S)                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
        if (GNNLM_STAB_EXP != 1 && GNNLM_STAB_EXP != 8) {                                            \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa##S[q].x, ua##S.x, acc[q], 0, 0, 0);     \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa##S[q].y, ua##S.y, acc[q], 0, 0, 0);     \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa##S[q].z, ua##S.z, acc[q], 0, 0, 0);     \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa##S[q].w, ua##S.w, acc[q], 0, 0, 0);     \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb##S[q].x, ub##S.x, acc[q], 0, 0, 0);     \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb##S[q].y, ub##S.y, acc[q], 0, 0, 0);     \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb##S[q].z, ub##S.z, acc[q], 0, 0, 0);     \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb##S[q].w, ub##S.w, acc[q], 0, 0, 0);     \
        } else {                                                                                     \
            asm volatile("" :: "v"(xa##S[q].x), "v"(xa##S[q].w), "v"(xb##S[q].x), "v"(xb##S[q].w), "v"(ua##S.x), "v"(ub##S.w)); \
        }                                                                                            \
    }
        // step c: CUR = register set of chunk c, NXT = set that receives chunk c + 1
#define STAB_STEP1(c_, CUR, NXT)                                                                     \
    {                                                                                                \
        if ((c_) + 2 < NCH && GNNLM_STAB_EXP != 4) STAB_DMA((c_) + 2, (c_) & 1)                      \
        if ((c_) + 1 < NCH) {                                                                        \
            STAB_LOOK1((c_) + 1, NXT)                                                                \
            STAB_ULOAD((c_) + 1, NXT)                                                                \
            if ((c_) + 2 < NCH) STAB_CODES((c_) + 2)                                                 \
        }                                                                                            \
        STAB_MMA1(CUR)                                                                               \
        STAB_LAND();                                                                                 \
        __syncthreads();                                                                             \
    }
        STAB_ULOAD(0, A)
        STAB_LAND();
        __syncthreads();                                               // codes staged, chunk 0 landed
        if (NCH > 1) STAB_DMA(1, 1)
        STAB_CODES(0)
        STAB_LOOK1(0, A)
        if (NCH > 1) STAB_CODES(1)
        STAB_LAND();
        __syncthreads();                                               // chunk 1 landed; set A holds chunk 0
        for (int c = 0; c < NCH; c += 2) {
            STAB_STEP1(c, A, B)
            if (c + 1 < NCH) STAB_STEP1(c + 1, B, A)
        }
#undef STAB_CODES
#undef STAB_LOOK1
#undef STAB_ULOAD
#undef STAB_MMA1
#undef STAB_STEP1
        STAB_DMA(0, 0)                  // pass 2's first chunks fly under the softmax
        if (NCH > 1) STAB_DMA(1, 1)
        // C layout: acc[q][rr] = S[neighbour 64 half + 16 q + 4 g + rr][head n16]
        if (n16 < HB) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t okw = *reinterpret_cast<const uint32_t*>(okf + t * KGM + 64 * half + 16 * q + 4 * g);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int j = 64 * half + 16 * q + 4 * g + rr;
                    sc[(t * HB + n16) * SCS + j] = ((okw >> (8 * rr)) & 1u) ? acc[q][rr] : -INFINITY;
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
        if (q == 0 && half == 0 && h0 == 0 && lane == 0 && p.has_nb && live) p.has_nb[i_tok] = sum > 0.f ? 1.f : 0.f;
    }
    STAB_LAND();
    __syncthreads();                    // alphas written, chunks 0 and 1 landed

    // ================================================================ pass 2: Z[16 (8 real) heads x 32 dims] = alpha^T X per chunk
    {
        const int kh = half;
        // k step ks, lane group g  <->  neighbour j = 64 kh + 4 ks + g (alpha = 0 and code row = zeros for padding)
        float a_reg[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) a_reg[ks] = n16 < HB ? sc[(t * HB + n16) * SCS + 64 * kh + 4 * ks + g] : 0.f;
        // B operand of column tile ct: dim 16 ct + n16 of the chunk = sub-quantizer mloc, component n16 % DSUB
        constexpr unsigned ROWSH = DSUB == 8 ? 5 : 4;      // log2 of a centroid row in bytes
        unsigned lbase[2], shift[2];                       // LDS byte address of (row 0, this lane's component) in buffer 0
        int widx[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int dl = 16 * ct + n16, mloc = dl / DSUB;
            lbase[ct] = (unsigned)(uintptr_t)(lds_void_t*)tab + (mloc * 256 * DSUB + (dl % DSUB)) * 4;
            shift[ct] = 8 * (mloc & 3);
            widx[ct] = mloc >> 2;
        }
        const unsigned char* cbase = lcodes + (t * KGM + 64 * kh + g) * MS;
        float* zp = zpart + t * 512 + lane;
        float* zo = p.Z + ((int64_t)i_tok * H + h0 + 4 * g) * D + n16;
        uint32_t w0[16], w1[16];        // code words of this lane's 16 neighbours for the chunk whose look-ups come next
        float b0A[16], b1A[16], b0B[16], b1B[16];
        f32x4 z0, z1;
#define STAB_WORDS(c_)                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 16; ++ks) {                                             \
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(cbase + 4 * ks * MS + MPC * (c_));   \
        w0[ks] = wp[widx[0]];                                                                        \
        w1[ks] = MPC == 4 ? w0[ks] : wp[widx[1]];                                                    \
    }
    // Two VALU operations per look-up (v_bfe_u32 + v_lshl_add_u32 on integer LDS addresses): the compiler's own
    // address arithmetic took five and made the VALU, not the matrix pipe, the longest chain of this pass.
#define STAB_LOOK2(c_, S)      /* the look-ups ARE the B operands */                                 \
    {                                                                                                \
        const unsigned sb0_ = lbase[0] + ((c_) & 1) * (TABF * 4), sb1_ = lbase[1] + ((c_) & 1) * (TABF * 4); \
        _Pragma("unroll") for (int ks = 0; ks < 16; ++ks) {                                         \
            const unsigned c0 = GNNLM_STAB_EXP == 3 ? 0u : __builtin_amdgcn_ubfe(w0[ks], shift[0], 8u); \
            const unsigned c1 = GNNLM_STAB_EXP == 3 ? 0u : __builtin_amdgcn_ubfe(w1[ks], shift[1], 8u); \
            b0##S[ks] = *(lds_cfloat_t*)(uintptr_t)((c0 << ROWSH) + sb0_);                           \
            b1##S[ks] = *(lds_cfloat_t*)(uintptr_t)((c1 << ROWSH) + sb1_);                           \
        }                                                                                            \
    }
    // A wave can have 15 LDS operations in flight (lgkmcnt is 4 bits) and issues in order: 48 look-ups placed in front of
    // the 32 MFMAs are a latency chain the matrix pipe waits behind, whatever chunk they belong to.  So the step
    // interleaves: per k step, the two look-ups of chunk c + 1 (+ the code word of chunk c + 2), then the two MFMAs
    // of chunk c; the sched_barrier pins that order.  Past the last chunk the look-ups run on a clamped index (unused).
#define STAB_STEP2(c_, CUR, NXT)                                                                     \
    {                                                                                                \
        if ((c_) + 2 < NCH && GNNLM_STAB_EXP != 4) STAB_DMA((c_) + 2, (c_) & 1)                      \
        const unsigned sb0_ = lbase[0] + (((c_) + 1) & 1) * (TABF * 4), sb1_ = lbase[1] + (((c_) + 1) & 1) * (TABF * 4); \
        const unsigned char* cw_ = cbase + MPC * min((c_) + 2, NCH - 1);                             \
        z0 = f32x4{0.f, 0.f, 0.f, 0.f};                                                              \
        z1 = f32x4{0.f, 0.f, 0.f, 0.f};                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 16; ++ks) {                                         \
            const unsigned c0 = GNNLM_STAB_EXP == 3 ? 0u : __builtin_amdgcn_ubfe(w0[ks], shift[0], 8u); \
            const unsigned c1 = GNNLM_STAB_EXP == 3 ? 0u : __builtin_amdgcn_ubfe(w1[ks], shift[1], 8u); \
            b0##NXT[ks] = *(lds_cfloat_t*)(uintptr_t)((c0 << ROWSH) + sb0_);                         \
            b1##NXT[ks] = *(lds_cfloat_t*)(uintptr_t)((c1 << ROWSH) + sb1_);                         \
            const uint32_t* wp = reinterpret_cast<const uint32_t*>(cw_ + 4 * ks * MS);               \
            w0[ks] = wp[widx[0]];                                                                    \
            w1[ks] = MPC == 4 ? w0[ks] : wp[widx[1]];                                                \
            if (GNNLM_STAB_EXP != 2 && GNNLM_STAB_EXP != 8) {                                        \
                z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[ks], b0##CUR[ks], z0, 0, 0, 0);      \
                z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[ks], b1##CUR[ks], z1, 0, 0, 0);      \
            } else {                                                                                 \
                asm volatile("" :: "v"(b0##CUR[ks]), "v"(b1##CUR[ks]));                              \
            }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                       \
        }                                                                                            \
        /* C layout: z[rr] = Z[head 4 g + rr][dim 32 c + 16 ct + n16]; heads 8..15 (g >= 2) are padding */ \
        float* zb = zp + ((c_) & 1) * 256;                                                           \
        if (kh == 1 && g < 2) {                                                                      \
            _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) { zb[32 * rr] = z0[rr]; zb[128 + 32 * rr] = z1[rr]; } \
        }                                                                                            \
        STAB_LAND();                                                                                 \
        __syncthreads();        /* partial sums of the other half written; chunk c + 2 landed; look-ups of c + 1 done */ \
        if (kh == 0 && g < 2 && live && GNNLM_STAB_EXP != 7) {                                       \
            _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                        \
                if (h0 + 4 * g + rr < H) {                                                           \
                    zo[(int64_t)rr * D + (c_) * CD] = z0[rr] + zb[32 * rr];                          \
                    zo[(int64_t)rr * D + (c_) * CD + 16] = z1[rr] + zb[128 + 32 * rr];               \
                }                                                                                    \
        }                                                                                            \
    }
        STAB_WORDS(0)
        STAB_LOOK2(0, A)
        if (NCH > 1) STAB_WORDS(1)
        __syncthreads();                // every wave's look-ups of chunk 0 are done before step 0 refills buffer 0
        for (int c = 0; c < NCH; c += 2) {
            STAB_STEP2(c, A, B)
            if (c + 1 < NCH) STAB_STEP2(c + 1, B, A)
        }
#undef STAB_WORDS
#undef STAB_LOOK2
#undef STAB_STEP2
    }
#undef STAB_DMA
#undef STAB_LAND
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

// f32 "NT" GEMM on the f32 matrix cores with a hand-placed main loop: the variant of gemm_f32.hip's kernel for
// 128x128 tiles whose k-loop is ONE inline-asm statement (gemm_sched_loop.inc, written by tools/gen_gemm_sched.py).
// Same contract, same epilogues (gemm_epilogue.inc), same numerics (v_mfma_f32_32x32x2_f32 fmaf chains; the k order
// inside a stage is permuted, which the MFMA sums over).
//
// Why: next to the f32 MFMA stream every other instruction costs the SIMD ~4-5 issue cycles (tools/probes/mfma_mix.hip)
// and an LDS-DMA piece 60-185 (MI355X_MICROARCH.md), so the loop carries the minimum -- per stage of 64 MFMAs and wave:
// 8 buffer loads of the NEXT stage into registers (soffset = k, no address arithmetic), 8 ds_write_b128 into the other
// LDS buffer behind counted vmcnt waits, 16 fragment reads placed one k-group ahead, one barrier -- at fixed places.
//
// LDS image of a stage (32 KiB, two of them): [256 rows (A 128 + W 128)][32 floats], slot s (16 B) of row r holds
// k-chunk s ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group (16 consecutive rows, one chunk) cover all banks.
#include "kernels.h"

namespace gnnlm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {
enum { EPI_STORE = 0, EPI_LSE = 1 };

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_sched_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 128, BK = 32, TM = 2, TN = 2, WROWS = 64, WCOLS = 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][256 rows][BK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;

    int M = p.M;
    if (p.m_dev) M = min(M, *p.m_dev);
    if (p.m_out && blockIdx.x == 0 && threadIdx.x == 0) *p.m_out = M;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = ((p.m_dev ? M : p.M) + BM - 1) / BM;
    const unsigned n_tiles = (unsigned)(tiles_m * tiles_n);
    const unsigned n_work = n_tiles * (unsigned)(p.batch1 * p.batch2);
    const int nk = p.K / BK;                                          // even (dispatch)

    // staging role of the thread: 16-B chunk `chunk` of rows srow + 32 q of the A and of the W tile
    const int srow = tid >> 3, chunk = tid & 7;
    const unsigned lw = (unsigned)(srow * 128 + ((chunk ^ ((srow >> 1) & 7)) << 4));
    // fragment reads: lane (l32, half) of k-group s reads chunk 2 s + half of rows l32 (+ 32 i) of its wave's slab
    const unsigned swz = (unsigned)((l32 >> 1) & 7);
    unsigned ra[4], rb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned slot = (((unsigned)(2 * s + half)) ^ swz) << 4;
        ra[s] = (unsigned)((wm * WROWS + l32) * 128) + slot;
        rb[s] = (unsigned)(BM * 128 + (wn * WCOLS + l32) * 128) + slot;
    }

    // (batch, tile) of list position v: tile origin, panel bases, byte offsets of the thread's staging rows inside the
    // panels (the dispatch checked that they fit 32 bits)
    struct Tile { int m0, n0, b1, b2; const float *A, *W; unsigned oa[4], ow[4]; };
    auto setup = [&](unsigned v_, Tile& t_) {
        const unsigned w_ = xcd_remap(v_, n_work);
        const unsigned by = w_ / n_tiles, t = w_ - by * n_tiles;
        t_.b1 = by / p.batch2; t_.b2 = by % p.batch2;
        int tm, tn;
        if (p.tile_order == 1) { tm = t / tiles_n; tn = t % tiles_n; }
        else if (p.tile_order == 2) { tn = t / tiles_m; tm = t % tiles_m; }
        else {
            const int GM = p.tile_order - 2;
            const int band = t / (GM * tiles_n);
            const int m_in = min(GM, tiles_m - band * GM);
            const int r = t - band * GM * tiles_n;
            tn = r / m_in;
            tm = band * GM + r % m_in;
        }
        t_.m0 = tm * BM; t_.n0 = tn * BN;
        t_.A = p.A + t_.b1 * p.sA1 + t_.b2 * p.sA2;
        t_.W = p.W + t_.b1 * p.sW1 + t_.b2 * p.sW2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gr = t_.m0 + srow + 32 * q;
            int64_t ar = gr < M ? (p.a_rows ? (int64_t)p.a_rows[gr] : (int64_t)gr) : 0;
            if (ar < 0) ar = 0;                                        // zero row, applied in the epilogue
            t_.oa[q] = (unsigned)(ar * p.lda * 4 + chunk * 16);
            const int gn = t_.n0 + srow + 32 * q;
            t_.ow[q] = (unsigned)((int64_t)(gn < p.N ? gn : 0) * p.ldw * 4 + chunk * 16);
        }
    };
    float4 sa0, sa1, sa2, sa3, sw0, sw1, sw2, sw3;                     // stage 0 of the tile about to start
#define GNNLM_LD16(base_, off_) (*reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base_) + (off_)))
#define GNNLM_LOAD_STAGE0(t_)                                                                                      \
    sa0 = GNNLM_LD16((t_).A, (t_).oa[0]); sa1 = GNNLM_LD16((t_).A, (t_).oa[1]);                                    \
    sa2 = GNNLM_LD16((t_).A, (t_).oa[2]); sa3 = GNNLM_LD16((t_).A, (t_).oa[3]);                                    \
    sw0 = GNNLM_LD16((t_).W, (t_).ow[0]); sw1 = GNNLM_LD16((t_).W, (t_).ow[1]);                                    \
    sw2 = GNNLM_LD16((t_).W, (t_).ow[2]); sw3 = GNNLM_LD16((t_).W, (t_).ow[3]);

    unsigned v = blockIdx.x;
    if (v >= n_work) return;
    Tile cur;
    setup(v, cur);
    GNNLM_LOAD_STAGE0(cur)
    while (true) {
        const int m0 = cur.m0, n0 = cur.n0, b1 = cur.b1, b2 = cur.b2;
        // the tile's row maps (gathered problems): one row per thread, loaded now, landed long before the epilogue reads them
        int map_c = 0, map_a = 0;
        float map_g = 1.f;
        if (EPI == EPI_STORE && tid < BM && m0 + tid < M) {
            if (p.c_rows) map_c = p.c_rows[m0 + tid];
            if (p.a_rows) map_a = p.a_rows[m0 + tid];
            if (p.gate) map_g = p.gate[m0 + tid];
        }
        // ... and the per-column bias of this lane's two columns: with both in hand the store epilogue issues no load of its
        // own, so it does not wait (in-order vmcnt) for the next tile's first stage that is requested just before it
        int pick_reg = -1;                                             // log-sum-exp problems: the tile's pick columns, likewise
        if (EPI == EPI_LSE && tid < BM && p.lse_pick && m0 + tid < M) pick_reg = p.lse_pick[m0 + tid];
        float bias_reg[TN] = {0.f, 0.f};
        if (EPI == EPI_STORE && p.bias && p.bias_mode == 1) {
            const float* bp = p.bias + b1 * p.sB1 + b2 * p.sB2;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WCOLS + j * 32 + l32;
                bias_reg[j] = col < p.N ? bp[col] : 0.f;
            }
        }
        // stage 0 -> LDS buffer 0
#define GNNLM_ST16(off_, v_) *reinterpret_cast<float4*>(reinterpret_cast<char*>(lds) + lw + (off_)) = v_;
        GNNLM_ST16(0, sa0) GNNLM_ST16(4096, sa1) GNNLM_ST16(8192, sa2) GNNLM_ST16(12288, sa3)
        GNNLM_ST16(16384, sw0) GNNLM_ST16(20480, sw1) GNNLM_ST16(24576, sw2) GNNLM_ST16(28672, sw3)
#undef GNNLM_ST16
        __syncthreads();
        // buffer descriptors (wave-uniform: kernel arguments and the tile's batch): base, stride 0, 4 GiB window
        const uint64_t pa = (uint64_t)(uintptr_t)cur.A, pw = (uint64_t)(uintptr_t)cur.W;
        i32x4 sa, sw;
        sa[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa);
        sa[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((pa >> 32) & 0xffffu));
        sa[2] = -1;
        sa[3] = 0x00020000;
        sw[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)pw);
        sw[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((pw >> 32) & 0xffffu));
        sw[2] = -1;
        sw[3] = 0x00020000;

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        int cnt = nk >> 1;
        int koff = nk > 1 ? BK * 4 : 0;
        const int inc = BK * 4;
#ifdef GNNLM_SCHED_AGPR
#define GNNLM_ACC_RC "+a"
#else
#define GNNLM_ACC_RC "+v"
#endif
#define GNNLM_SCHED_OPERANDS                                                                                                    \
            : [c00] GNNLM_ACC_RC(acc[0][0]), [c01] GNNLM_ACC_RC(acc[0][1]), [c10] GNNLM_ACC_RC(acc[1][0]), [c11] GNNLM_ACC_RC(acc[1][1]), [cnt] "+s"(cnt), [koff] "+s"(koff) \
            : [sa] "s"(sa), [sw] "s"(sw), [inc] "s"(inc),                                                                             \
              [oa0] "v"(cur.oa[0]), [oa1] "v"(cur.oa[1]), [oa2] "v"(cur.oa[2]), [oa3] "v"(cur.oa[3]),                                                  \
              [ow0] "v"(cur.ow[0]), [ow1] "v"(cur.ow[1]), [ow2] "v"(cur.ow[2]), [ow3] "v"(cur.ow[3]), [lw] "v"(lw),                                    \
              [ra0] "v"(ra[0]), [ra1] "v"(ra[1]), [ra2] "v"(ra[2]), [ra3] "v"(ra[3]),                                                  \
              [rb0] "v"(rb[0]), [rb1] "v"(rb[1]), [rb2] "v"(rb[2]), [rb3] "v"(rb[3])                                                   \
            : "memory", "vcc", "scc",
        if constexpr (EPI == EPI_LSE) {          // operands swapped: transposed accumulators (gemm_epilogue.inc)
            asm volatile(
#include "gemm_sched_loop_t.inc"
                GNNLM_SCHED_OPERANDS
#include "gemm_sched_clobbers.inc"
            );
        } else {
            asm volatile(
#include "gemm_sched_loop.inc"
                GNNLM_SCHED_OPERANDS
#include "gemm_sched_clobbers.inc"
            );
        }
#undef GNNLM_SCHED_OPERANDS
        __syncthreads();                                               // every wave left the loop: the LDS image is free
        // in LDS buffer 1: the next tile's stage 0 goes to buffer 0 and buffer 1 is first written at the end of its stage 0, behind
        // the barrier every wave passes after its own epilogue -- so the store epilogue needs no barrier of its own behind it
        int* rowmap_c = reinterpret_cast<int*>(lds) + 8192;
        int* rowmap_a = rowmap_c + BM;
        float* rowmap_g = reinterpret_cast<float*>(rowmap_a + BM);
        if (EPI == EPI_STORE && (p.c_rows || p.a_rows || p.gate)) {
            if (tid < BM) { rowmap_c[tid] = map_c; rowmap_a[tid] = map_a; rowmap_g[tid] = map_g; }
            __syncthreads();
        }

        // the next tile's first stage travels under this tile's epilogue
        const unsigned vn = v + gridDim.x;
        const bool more = vn < n_work;
        Tile nxt = cur;
        if (more) {
            setup(vn, nxt);
            GNNLM_LOAD_STAGE0(nxt)
        }
        if (EPI == EPI_LSE) {
            if (tid < BM) reinterpret_cast<int*>(lds)[tid] = pick_reg;
            __syncthreads();
        }
#define GNNLM_EPI_ROWMAP
#define GNNLM_EPI_BIAS_REG
#define GNNLM_LSE_PICK_STAGED
#include "gemm_epilogue.inc"
#undef GNNLM_LSE_PICK_STAGED
#undef GNNLM_EPI_BIAS_REG
#undef GNNLM_EPI_ROWMAP
        if (EPI == EPI_LSE) __syncthreads();                          // its pick staging sits in buffer 0
        if (!more) break;
        cur = nxt;
        v = vn;
    }
#undef GNNLM_LOAD_STAGE0
#undef GNNLM_LD16
}
}  // namespace

// Where the scheduled kernel is taken: exact f32, K a multiple of 64 (the loop runs stage pairs), 128x128 tiles filling the
// chip, rows addressed with 32-bit byte offsets inside the panel.  GNNLM_GEMM_SCHED=0 sends everything to the other kernels.
bool gemm_sched_eligible(const GemmParams& p) {
    // GNNLM_GEMM_SCHED (A/B runs): 0 = never, 1 = store-epilogue problems only, 3 (default) = log-sum-exp problems too, 7 = the head as well
    static const int on = [] { const char* e = getenv("GNNLM_GEMM_SCHED"); return e ? atoi(e) : 3; }();
    if (!on) return false;
    if (p.precision != 0) return false;
    if (p.lse_part && !(on & 2)) return false;
    // the softmax head (>= 2048 tiles of 256x256) runs in the same time here as on the 256x256 LDS-DMA kernel (2.47 ms), but
    // with 128-wide W panels it pulls the A panel through the fabric twice as often (PMC FETCH_SIZE 2.9 vs 1.4 GB): stays there
    if (p.lse_part && !p.m_dev && !(on & 4) && cdiv(p.M, 256) * cdiv(p.N, 256) >= 2048) return false;
    static const int min_k = [] { const char* e = getenv("GNNLM_GEMM_SCHED_MINK"); return e ? atoi(e) : 256; }();
    if (p.K % 64 != 0 || p.K < min_k) return false;                       // K = 128 (absorbed queries): 191 us on the register-staged kernel, 206 here
    const int64_t nb = (int64_t)p.batch1 * p.batch2;
    if (cdiv(p.M, 128) * cdiv(p.N, 128) * nb < 256) return false;
    // rows are addressed by 32-bit byte offsets from the panel base of the tile's batch
    if ((int64_t)p.N * p.ldw * 4 >= (1ll << 32)) return false;
    const int64_t a_rows = p.a_rows ? p.a_rows_bound : (int64_t)p.M;    // gathered rows index [0, a_rows_bound)
    if (a_rows <= 0 || a_rows * p.lda * 4 >= (1ll << 32)) return false;
    return true;
}

template <int EPI>
static int launch_sched(const GemmParams& p, dim3 grid, hipStream_t stream) {
    constexpr size_t lds_bytes = 2 * 256 * 32 * sizeof(float);
    GNNLM_LDS_OPT_IN(&gemm_nt_f32_sched_kernel<EPI>, lds_bytes);
    hipLaunchKernelGGL((gemm_nt_f32_sched_kernel<EPI>), grid, dim3(256), lds_bytes, stream, p);
    return OK;
}

int gemm_nt_sched(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    const int64_t nb = (int64_t)p.batch1 * p.batch2;
    const int64_t tiles = cdiv(p.M, 128) * cdiv(p.N, 128) * nb;
    GNNLM_REQUIRE(tiles < (1ll << 31), "gemm: grid too large");
    dim3 grid((unsigned)std::min<int64_t>(tiles, 512));
    const double work = 2.0 * p.M * (double)p.N * p.K * nb;
    ProfScope prof(K_GEMM, stream, work, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N) * nb,
                   p.m_dev, (double)p.M, true);
    if (prof.slot) p.m_out = prof.slot;
    const int rc = p.lse_part ? launch_sched<EPI_LSE>(p, grid, stream) : launch_sched<EPI_STORE>(p, grid, stream);
    if (rc != OK) return rc;
    GNNLM_LAUNCH_CHECK();
    return OK;
}

}  // namespace gnnlm

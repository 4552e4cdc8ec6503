// C ABI of libgnnlm_hip.so (include/gnnlm.h): thin wrappers over the kernel launchers plus the two
// host-side orchestrators of the eval hot path -- the HGT forward over the implicit token/neighbour
// graph and the target-only tied adaptive softmax.  Orchestrators only enqueue kernels on the given
// stream (no allocation, no synchronisation), so a caller may capture them into a hipGraph.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace gnnlm {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_last_error = buf;
    return E_HIP;
}

int lds_opt_in(const void* kernel_fn, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;       // (function, device) pairs already opted in
    int dev = 0;
    GNNLM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : done)
        if (e.first == kernel_fn && e.second == dev) return OK;
    GNNLM_HIP(hipFuncSetAttribute(kernel_fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.emplace_back(kernel_fn, dev);
    return OK;
}

// ---------------------------------------------------------------------------------------------
// profiler: (start, stop) event pairs per launch, resolved at gnnlm_profile_end
// ---------------------------------------------------------------------------------------------
unsigned g_prof_mask = 0;
namespace {
struct ProfRec { int kid; hipEvent_t a, b; double flops, bytes; int32_t* host_scale; double den; };
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_event_pool;
std::vector<int32_t*> g_pinned_pool;
// the start event of an open ProfScope is per host thread (one scope per kernel id at a time on a thread); the
// record list and the pools are shared and guarded by g_prof_mu, so launches from several host threads may be
// profiled together
thread_local hipEvent_t g_pending_start[K_COUNT];
std::mutex g_prof_mu;
hipEvent_t take_event() {           // g_prof_mu held; nullptr on failure
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
int32_t* take_pinned() {            // g_prof_mu held; nullptr on failure
    // slots come from slabs of 1024 pinned ints: hipHostMalloc is a heavy, device-synchronising call that must not
    // happen per launch inside a timed region
    if (g_pinned_pool.empty()) {
        int32_t* slab = nullptr;
        if (hipHostMalloc((void**)&slab, 1024 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess || !slab) return nullptr;
        for (int i = 0; i < 1024; ++i) g_pinned_pool.push_back(slab + i);
    }
    int32_t* p = g_pinned_pool.back();
    g_pinned_pool.pop_back();
    return p;
}
}  // namespace
void prof_start(int kid, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_pending_start[kid] = take_event();
    if (g_pending_start[kid]) (void)hipEventRecord(g_pending_start[kid], s);
}
int32_t* prof_take_slot() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return take_pinned();           // nullptr: the launch simply does not report its device-side count
}
void prof_stop(int kid, hipStream_t s, double flops, double bytes, const int32_t* scale_dev, double den, int32_t* host_slot) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{kid, g_pending_start[kid], take_event(), flops, bytes, nullptr, den};
    if (!r.a || !r.b) {             // event creation failed: this launch is not recorded
        if (r.a) g_event_pool.push_back(r.a);
        if (r.b) g_event_pool.push_back(r.b);
        if (host_slot) g_pinned_pool.push_back(host_slot);
        return;
    }
    (void)hipEventRecord(r.b, s);
    if (host_slot) {
        r.host_scale = host_slot;                       // written by the kernel itself
    } else if (scale_dev) {
        r.host_scale = take_pinned();
        if (r.host_scale) (void)hipMemcpyAsync(r.host_scale, scale_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    }
    g_prof_recs.push_back(r);
}

namespace {

// bump allocator over a caller-provided workspace; a null base only measures
struct Carver {
    char* base;
    size_t off = 0, cap;
    Carver(void* b, size_t c) : base(reinterpret_cast<char*>(b)), cap(c) {}
    template <class T>
    T* take(int64_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (size_t)std::max<int64_t>(n, 0) * sizeof(T);
        return p;
    }
    bool fits() const { return !base || off <= cap; }
};

struct HgtBufs {
    // tgt side
    float *ht[2], *q, *k, *vt, *scores, *mc, *U, *Z, *ms, *aout, *has_nb;
    // ntgt side
    float *hn[2], *nq, *nk, *nv;
    uint8_t* valid;
    int32_t *rows_out, *rows_kv;     // slot subsets of the elided ntgt updates
    int32_t *win_counts;             // device-side group count (ABI 9): rows per GEMM window x multiplier
    int32_t *nb_row;                 // merged groups on fetched codes: code row of every neighbour's centre
    // ABI 11, layer 0's K / V keyed by datastore row: the slots' rows, the distinct ones, slot -> position, their count, validity, decoded rows
    int64_t *slot_row, *urows;
    int32_t *slot_u, *row_cnt, *win_counts_u;
    uint8_t* uvalid;
    float* xu;
    int64_t Tp;
};
constexpr int MAX_WINDOWS = 128;

bool needs_ntgt(const gnnlm_hgt_t& m, const gnnlm_hgt_io_t& io) { return m.n_layers > 1 || io.out_ntgt != nullptr; }
// layer 0's K / V projections once per distinct datastore ROW of the batch's slots (ABI 11): merged groups on a local or mapped
// store, rotation-folded layer-0 weights, and a model whose layer 0 computes a subset of the slots (the un-elided forward with
// out_ntgt keeps the slot-keyed path)
bool row_keyed_kv(const gnnlm_hgt_t& m, const gnnlm_hgt_io_t& io) {
    if (!io.row_table || !io.group_ids || io.fetched_codes || io.ntgt_feats || io.out_ntgt || m.n_layers < 2 || !m.opq_at) return false;
    const gnnlm_hgt_layer_t& w0 = m.layers[0];
    return w0.wq_n0 && w0.bq_n0 && w0.wk_n0 && w0.bk_n0 && w0.wv_n0 && w0.bv_n0 && m.n_layers - 2 < std::max(m.left, m.right);
}

void carve_hgt(const gnnlm_hgt_t& m, const gnnlm_hgt_io_t& io, Carver& c, HgtBufs& b) {
    const int64_t Tt = (int64_t)io.n_blocks * io.T, d = m.d, H = m.n_heads;
    const int64_t dpq = (int64_t)m.M * m.dsub, dmax = std::max<int64_t>(dpq, d);
    b.Tp = (io.T + 3) & ~3;
    b.slot_row = b.urows = nullptr; b.slot_u = b.row_cnt = b.win_counts_u = nullptr; b.uvalid = nullptr; b.xu = nullptr;
    b.ht[0] = c.take<float>(Tt * d);
    b.ht[1] = c.take<float>(Tt * d);
    b.q = c.take<float>(Tt * d);
    b.k = c.take<float>(Tt * d);
    b.vt = c.take<float>((int64_t)io.n_blocks * d * b.Tp);
    b.scores = c.take<float>((int64_t)io.n_blocks * H * io.T * b.Tp);
    b.mc = c.take<float>(Tt * d);
    b.U = c.take<float>(Tt * H * dmax);
    b.Z = c.take<float>(Tt * H * dmax);
    b.ms = c.take<float>(Tt * d);
    b.aout = c.take<float>(Tt * d);
    b.has_nb = c.take<float>(Tt);
    b.win_counts = io.n_unique_dev ? c.take<int32_t>(MAX_WINDOWS * 8) : nullptr;
    b.nb_row = (io.group_ids && io.fetched_codes && !io.state_cache) ? c.take<int32_t>(Tt * io.kg) : nullptr;
    if (needs_ntgt(m, io)) {
        const int64_t S = (io.group_ids ? io.n_unique : Tt * io.kg) * (1 + m.left + m.right);
        b.hn[0] = c.take<float>(S * dmax);
        b.hn[1] = c.take<float>(S * dmax);
        b.nq = c.take<float>(S * dmax);
        b.nk = c.take<float>(S * dmax);
        b.nv = c.take<float>(S * dmax);
        b.valid = c.take<uint8_t>(S);
        b.rows_out = c.take<int32_t>(S);
        b.rows_kv = c.take<int32_t>(S);
        if (row_keyed_kv(m, io)) {
            b.slot_row = c.take<int64_t>(S);
            b.urows = c.take<int64_t>(S);
            b.slot_u = c.take<int32_t>(S);
            b.row_cnt = c.take<int32_t>(4);
            b.win_counts_u = c.take<int32_t>(MAX_WINDOWS * 8);
            b.uvalid = c.take<uint8_t>(S);
            b.xu = c.take<float>(S * dpq);
        }
    } else {
        b.hn[0] = b.hn[1] = b.nq = b.nk = b.nv = nullptr;
        b.valid = nullptr;
        b.rows_out = b.rows_kv = nullptr;
    }
}

int linear(const float* A, int64_t lda, const float* W, const float* bias, float* C, int64_t M, int N, int K,
           const float* R, float alpha, hipStream_t s, const int32_t* m_dev = nullptr) {
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.C = C; g.ldc = N;
    g.bias = bias; g.bias_mode = bias ? 1 : 0;
    g.R = R; g.ldr = N; g.alpha = alpha;
    g.M = (int)M; g.N = N; g.K = K;
    g.m_dev = m_dev;
    if (m_dev && M > N) g.tile_order = 1;       // the walk the host-side count would have picked (gemm_nt's default assumes a small device count)
    return gemm_nt(g, s);
}

// the same on a subset of rows: logical row r reads A row rows[r] and writes C row rows[r] (both in their full layouts)
int linear_rows(const float* A, int64_t lda, const float* W, const float* bias, float* C, int64_t ldc, const int32_t* rows,
                int64_t n_rows, int N, int K, float alpha, hipStream_t s, int64_t rows_bound = 0, const int32_t* m_dev = nullptr) {
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.C = C; g.ldc = ldc;
    g.bias = bias; g.bias_mode = bias ? 1 : 0;
    g.alpha = alpha;
    g.a_rows = rows; g.c_rows = rows; g.a_rows_bound = rows_bound;
    g.M = (int)n_rows; g.N = N; g.K = K;
    g.m_dev = m_dev;
    if (m_dev && n_rows > N) g.tile_order = 1;
    return gemm_nt(g, s);
}

#define TRY(x)                 \
    do {                       \
        int _rc = (x);         \
        if (_rc != OK) return _rc; \
    } while (0)

// The ntgt projections of a multi-layer step run over [groups x slots] rows of a few KB each: from ~1 M rows on, the operand
// spans more than 2^32 bytes and the GEMM kernel with 32-bit row offsets (gemm_f32_sched.hip, the fastest one for these
// shapes: 133-136 against 123-128 TFLOP/s, tools/gemm_bigm_bench.py) is no longer eligible.  Rows are independent, so the
// launches walk the groups in WINDOWS whose rows fit 32-bit byte offsets: same kernels per row as a 4-block step, same bits.
struct GroupWindows {
    int64_t G, per;       // groups, groups per window (a multiple of 128)
    int n_g;
    const int32_t* counts = nullptr;      // device-side group count: counts[w * 8 + mult - 1] = rows of window w x mult (window_counts)
    GroupWindows(int64_t G_, int n_g_, int64_t row_bytes) : G(G_), n_g(n_g_) {
        const int64_t max_rows = ((1ll << 32) - 1) / std::max<int64_t>(row_bytes, 1);
        const int64_t max_groups = std::max<int64_t>(128, max_rows / n_g / 128 * 128);
        const int64_t n_win = std::max<int64_t>(1, cdiv(G, max_groups));
        per = std::min<int64_t>(max_groups, cdiv(cdiv(G, n_win), (int64_t)128) * 128);
    }
    int64_t n_windows() const { return std::max<int64_t>(1, cdiv(G, per)); }
    const int32_t* m_dev(int64_t g0, int mult) const { return counts ? counts + (g0 / per) * 8 + (mult - 1) : nullptr; }
};
// every slot row of every group
int linear_windows(const GroupWindows& w, const float* A, int64_t lda, const float* W, const float* bias, float* C, int N, int K, hipStream_t s) {
    for (int64_t g0 = 0; g0 < w.G; g0 += w.per) {
        const int64_t g1 = std::min(w.G, g0 + w.per), r0 = g0 * w.n_g;
        const int rc = linear(A + r0 * lda, lda, W, bias, C + r0 * N, (g1 - g0) * w.n_g, N, K, nullptr, 1.f, s, w.m_dev(g0, w.n_g));
        if (rc != OK) return rc;
    }
    return OK;
}
// a subset of the slots of every group: `rows` = group_rows(n_sel slots of G groups) is group-major with indices g * n_g + slot,
// so its PREFIX is the row map of any window once the operands are based at the window's first row
int linear_rows_windows(const GroupWindows& w, const float* A, int64_t lda, const float* W, const float* bias, float* C, int64_t ldc,
                        const int32_t* rows, int n_sel, int N, int K, hipStream_t s) {
    for (int64_t g0 = 0; g0 < w.G; g0 += w.per) {
        const int64_t g1 = std::min(w.G, g0 + w.per), r0 = g0 * w.n_g;
        const int rc = linear_rows(A + r0 * lda, lda, W, bias, C + r0 * ldc, ldc, rows, (g1 - g0) * n_sel, N, K, 1.f, s, (g1 - g0) * w.n_g, w.m_dev(g0, n_sel));
        if (rc != OK) return rc;
    }
    return OK;
}

int hgt_forward_impl(const gnnlm_hgt_t& m, const gnnlm_hgt_io_t& io, void* ws, size_t ws_bytes, hipStream_t s) {
    GNNLM_REQUIRE(m.layers && m.n_layers >= 1, "hgt: no layers");
    GNNLM_REQUIRE(m.gemm_precision >= 0 && m.gemm_precision <= 2, "hgt: gemm_precision must be 0, 1 or 2");
    GemmPrecisionScope prec_scope(m.gemm_precision);
    GNNLM_REQUIRE(m.d > 0 && m.n_heads > 0 && m.d % m.n_heads == 0, "hgt: d must be divisible by n_heads");
    GNNLM_REQUIRE(io.n_blocks >= 0 && io.T >= 0 && io.kg > 0, "hgt: bad io shape");
    if ((int64_t)io.n_blocks * io.T == 0) return OK;                 // empty batch: nothing to do
    GNNLM_REQUIRE(io.tgt_feats && io.ids && io.out_tgt, "hgt: null io");
    const bool dense0 = io.ntgt_feats != nullptr;        // layer-0 ntgt states given (input adapters): no code store
    GNNLM_REQUIRE(dense0 || (m.centroids && m.M > 0 && m.dsub > 0), "hgt: codec missing");
    GNNLM_REQUIRE(dense0 || io.fetched_codes || m.codes || m.shards, "hgt: no code store");
    GNNLM_REQUIRE(!dense0 || (io.ntgt_valid && io.ld_ntgt >= m.d && io.ld_ntgt % 4 == 0 && !io.fetched_codes),
                  "hgt: ntgt_feats needs ntgt_valid, ld_ntgt >= d (a multiple of 4) and no fetched_codes");
    const bool ntgt = needs_ntgt(m, io);
    GNNLM_REQUIRE(!(io.fetched_centres_only && ntgt), "hgt: fetched_centres_only needs n_layers == 1 and no out_ntgt");
    // the ntgt update keeps a context group in registers / on the stack (chain attention, the row-subset selections) and
    // addresses slot rows with 32-bit indices
    GNNLM_REQUIRE(!ntgt || (m.left >= 0 && m.right >= 0 && 1 + m.left + m.right <= 8), "hgt: ntgt update needs 1 + left + right <= 8");
    GNNLM_REQUIRE(!ntgt || (int64_t)io.n_blocks * io.T * io.kg * (1 + m.left + m.right) < (1ll << 31),
                  "hgt: ntgt update needs n_blocks * T * k_g * (1 + left + right) < 2^31 slot rows per call");
    GNNLM_REQUIRE(!io.fetched_codes || io.fetched_valid || io.fetched_centres_only, "hgt: fetched_codes needs fetched_valid");
    const int64_t Tt = (int64_t)io.n_blocks * io.T;
    if (Tt == 0) return OK;
    const int d = m.d, H = m.n_heads, dk = d / H, T = io.T, kg = io.kg, nb = io.n_blocks;
    const int n_g = 1 + m.left + m.right;
    const int dpq = dense0 ? d : m.M * m.dsub;
    // ABI 6: the ntgt pipeline over the DISTINCT centre rows of the batch (io.group_ids); the star edges reach a group through
    // io.group_index
    const bool dedup = io.group_ids != nullptr;
    GNNLM_REQUIRE(!dedup || (ntgt && !dense0 && io.group_index && io.n_unique >= 0 && !io.out_ntgt && !io.out_valid),
                  "hgt: group_ids needs group_index, a multi-layer model on a code store and no ntgt outputs");
    // ABI 9: the number of groups lives on the device; n_unique is then the capacity of the group arrays
    const int32_t* gdev = io.n_unique_dev;
    GNNLM_REQUIRE(!gdev || dedup, "hgt: n_unique_dev needs group_ids");
    // ABI 7: centre states kept across calls (io.state_cache): group_ids are the groups the cache lacks, group_index holds slots
    const bool cached = io.state_cache != nullptr;
    GNNLM_REQUIRE(!cached || (dedup && io.cache_cap > 0 && (io.group_slot || io.n_unique == 0) && m.n_layers > 1),
                  "hgt: state_cache needs group_ids / group_index / group_slot, cache_cap > 0 and a multi-layer model");
    // ABI 9: merged groups on a sharded store -- the fetched slots are those of the groups; with the cache, the code rows of the
    // cached centres live in io.code_cache (layer 0's star edges read the code of every neighbour, fetched now or not)
    GNNLM_REQUIRE(!(cached && io.fetched_codes) || (io.code_cache && m.M % 16 == 0), "hgt: state_cache on fetched codes needs code_cache and M % 16 == 0");
    const int64_t G = dedup ? io.n_unique : Tt * kg, S = G * n_g;
    GNNLM_REQUIRE(dk % 4 == 0 && d % 4 == 0 && dpq % 4 == 0, "hgt: d_k and the PQ dimension must be multiples of 4");
    GNNLM_REQUIRE(dense0 || m.opq_at || dpq == d, "hgt: without OPQ the PQ dimension must equal d");

    Carver c(ws, ws_bytes);
    HgtBufs b;
    carve_hgt(m, io, c, b);
    GNNLM_REQUIRE(ws && c.fits(), "hgt: workspace too small (see gnnlm_hgt_workspace_bytes)");
    const int64_t Tp = b.Tp;

    const float* hn_cur = nullptr;
    bool fold0 = false;                                  // layer 0 projects the decoded rows (b.hn[1]) with rotation-folded weights
    int64_t ld_hn = d;                                   // row stride of hn_cur (the caller's buffer in the dense0 case)
    const uint8_t* valid = nullptr;
    if (dense0) {
        hn_cur = io.ntgt_feats;
        ld_hn = io.ld_ntgt;
        valid = io.ntgt_valid;
        if (io.out_valid) GNNLM_HIP(hipMemcpyAsync(io.out_valid, valid, (size_t)S, hipMemcpyDeviceToDevice, s));
    } else if (ntgt && S == 0) {                         // a de-duplicated batch without a single valid neighbour
        hn_cur = b.hn[0];
        valid = b.valid;
    } else if (ntgt) {
        // layer-0 ntgt states: PQ lookup of every slot, then the OPQ rotation (pq_wrapper.py:189-202)
        GatherParams g{};
        g.codes = io.fetched_codes ? io.fetched_codes : m.codes;
        if (!io.fetched_codes) g.shards = m.shards;
        g.direct = io.fetched_codes ? 1 : 0;
        g.in_valid = io.fetched_valid;
        g.in_index = io.fetched_index;
        g.vals = nullptr; g.vals_itemsize = 4;
        g.n_store = m.n_store; g.row0 = m.row0; g.n_local = m.n_local;
        g.M = m.M; g.dsub = m.dsub; g.centroids = m.centroids;
        g.ids = dedup ? io.group_ids : io.ids; g.n_groups = G; g.left = m.left; g.right = m.right;
        g.n_groups_dev = gdev;
        g.out_valid = b.valid;
        const gnnlm_hgt_layer_t& w0 = m.layers[0];
        fold0 = m.opq_at && w0.wq_n0 && w0.bq_n0 && w0.wk_n0 && w0.bk_n0 && w0.wv_n0 && w0.bv_n0;
        if (fold0) {
            // layer 0's ntgt projections take the decoded rows (rotation folded into their weights); the rotation itself is
            // computed inside the layer, for the rows whose residual is needed only
            g.out_x = b.hn[1]; g.ld_x = dpq;
            TRY(gather_decode(g, s));
        } else if (m.opq_at) {
            g.out_x = b.nq; g.ld_x = dpq;
            TRY(gather_decode(g, s));
            TRY(linear(b.nq, dpq, m.opq_at, m.opq_nba, b.hn[0], S, d, dpq, nullptr, 1.f, s));
        } else {
            g.out_x = b.hn[0]; g.ld_x = d;
            TRY(gather_decode(g, s));
        }
        hn_cur = b.hn[0];
        valid = b.valid;
        if (io.out_valid) GNNLM_HIP(hipMemcpyAsync(io.out_valid, b.valid, (size_t)S, hipMemcpyDeviceToDevice, s));
        if (fold0 && row_keyed_kv(m, io)) {
            // ABI 11: the distinct ROWS among the slots (a claim pass over the slots' rows against the caller's second row table, handed
            // back clean), decoded once; their count stays on the device (row_cnt[0]).  The slots' own decode above stays: layer 0's Q
            // and the LayerNorm residual are per slot
            TRY(slot_rows(io.group_ids, G, gdev, m.left, m.right, m.n_layers - 1, m.n_store, b.slot_row, s));   // (layer 0's K / V reach: n_layers - 2 + 1 positions)
            gnnlm_group_assign_t ra{};
            ra.ids = b.slot_row; ra.n = S; ra.n_store = m.n_store; ra.slot_of = io.row_table;
            ra.group_ids = b.urows; ra.group_index = b.slot_u; ra.counters = b.row_cnt;
            TRY(group_assign(ra, s));
            GatherParams gu = g;
            gu.ids = b.urows; gu.n_groups = S; gu.left = 0; gu.right = 0; gu.n_groups_dev = b.row_cnt;
            gu.out_valid = b.uvalid; gu.out_x = b.xu; gu.ld_x = dpq;
            TRY(gather_decode(gu, s));
        }
    }
    // V^T buffer of the GEMM path: its padding columns (t >= T) are read by the P.V GEMM against zero probabilities
    if (!causal_attn_fused_ok(T, dk)) GNNLM_HIP(hipMemsetAsync(b.vt, 0, sizeof(float) * (size_t)nb * d * Tp, s));

    const float* ht_in = io.tgt_feats;
    for (int l = 0; l < m.n_layers; ++l) {
        const gnnlm_hgt_layer_t& w = m.layers[l];
        const bool last = l == m.n_layers - 1;
        const int din = w.din;
        GNNLM_REQUIRE(w.wq_t && w.wk_t && w.wv_t && w.wa_t && w.ln_g_t && w.ln_b_t && w.wku && w.wvz_t,
                      "hgt: layer has null tgt weights");
        GNNLM_REQUIRE(din == (l == 0 ? dpq : d), "hgt: layer.din must be M*dsub for layer 0 and d afterwards");

        // ---- tgt projections (hgt.py:315-322 with the 'intra' relation folded into K and V)
        // q, k', v' of the batch: [Tt, ldq] with the three at column offsets 0 / d / 2d when they come out of one GEMM
        const float *qp = b.q, *kp = b.k, *vp = b.vt;
        int64_t ldq = d;
        const bool qkv_one = causal_attn_fused_ok(T, dk) && w.bq_t && w.wk_t == w.wq_t + (int64_t)d * d &&
                             w.wv_t == w.wk_t + (int64_t)d * d && w.bk_t == w.bq_t + d && w.bv_t == w.bk_t + d &&
                             b.k == b.q + Tt * d && b.vt == b.k + Tt * d;
        if (qkv_one) {
            // weights stored back to back (hgt.py does): ONE 3d-wide GEMM instead of three d-wide ones (same input; 1536
            // tiles on the 512 workgroup slots instead of 3 x 512 with a prologue and an epilogue each)
            TRY(linear(ht_in, d, w.wq_t, w.bq_t, b.q, Tt, 3 * d, d, nullptr, 1.f, s));
            kp = b.q + d; vp = b.q + 2 * d; ldq = 3 * (int64_t)d;
        } else {
            TRY(linear(ht_in, d, w.wq_t, w.bq_t, b.q, Tt, d, d, nullptr, 1.f, s));
            TRY(linear(ht_in, d, w.wk_t, w.bk_t, b.k, Tt, d, d, nullptr, 1.f, s));
        }
        if (qkv_one) {
        } else if (causal_attn_fused_ok(T, dk)) {
            // recipe shape: V' row-major, then scores + masked softmax + P.V in one kernel (attn.hip)
            // (the kernel itself runs after the star branch and adds its result into the message sum, see below)
            TRY(linear(ht_in, d, w.wv_t, w.bv_t, b.vt, Tt, d, d, nullptr, 1.f, s));
        } else {
            {   // V'^T[blk][n][t] = sum_k Wv'[n,k] h[blk*T + t, k] + bv'[n]
                GemmParams g{};
                g.A = w.wv_t; g.lda = d; g.W = ht_in; g.ldw = d; g.C = b.vt; g.ldc = Tp;
                g.bias = w.bv_t; g.bias_mode = 2;
                g.M = d; g.N = T; g.K = d; g.batch1 = nb;
                g.sW1 = (int64_t)T * d; g.sC1 = (int64_t)d * Tp;
                TRY(gemm_nt(g, s));
            }
            {   // causal scores S[blk,h] = Q_h K'_h^T   (scale folded into K')
                GemmParams g{};
                g.A = b.q; g.lda = d; g.W = b.k; g.ldw = d; g.C = b.scores; g.ldc = Tp;
                g.M = T; g.N = T; g.K = dk; g.batch1 = nb; g.batch2 = H;
                g.sA1 = (int64_t)T * d; g.sA2 = dk; g.sW1 = (int64_t)T * d; g.sW2 = dk;
                g.sC1 = (int64_t)H * T * Tp; g.sC2 = (int64_t)T * Tp;
                TRY(gemm_nt(g, s));
            }
            TRY(causal_softmax(b.scores, (int64_t)nb * H, T, Tp, m.max_intra_context, s));
            {   // m_causal[blk, :, h] = P[blk,h] V'_h
                GemmParams g{};
                g.A = b.scores; g.lda = Tp; g.W = b.vt; g.ldw = Tp; g.C = b.mc; g.ldc = d;
                g.M = T; g.N = dk; g.K = (int)Tp; g.batch1 = nb; g.batch2 = H;
                g.sA1 = (int64_t)H * T * Tp; g.sA2 = (int64_t)T * Tp;
                g.sW1 = (int64_t)d * Tp; g.sW2 = (int64_t)dk * Tp;
                g.sC1 = (int64_t)T * d; g.sC2 = dk;
                TRY(gemm_nt(g, s));
            }
        }
        {   // absorbed star queries U[i,h,:] = Wku_h q[i,h,:]
            GemmParams g{};
            g.A = qp; g.lda = ldq; g.W = w.wku; g.ldw = dk; g.C = b.U; g.ldc = (int64_t)H * din;
            g.M = (int)Tt; g.N = din; g.K = dk; g.batch1 = H;
            g.sA1 = dk; g.sW1 = (int64_t)din * dk; g.sC1 = din;
            TRY(gemm_nt(g, s));
        }
        {
            StarAttnParams a{};
            a.U = b.U; a.ids = io.ids; a.T = (int)Tt; a.H = H; a.D = din; a.kg = kg;
            a.Z = b.Z; a.has_nb = b.has_nb;
            a.n_store = m.n_store;
            if (cached) { }                                      // a cached group is a valid one: x_index >= 0 says it all
            else if (ntgt || dense0) { a.nb_valid = valid; a.nb_valid_stride = n_g; }        // centre slot of each group
            else if (io.fetched_valid) { a.nb_valid = io.fetched_valid; a.nb_valid_stride = io.fetched_centres_only ? 1 : n_g; }
            if (dedup) a.x_index = io.group_index;
            if (cached && l > 0) {
                a.X = io.state_cache + (int64_t)(l - 1) * io.cache_cap * d; a.ldx = d; a.x_group_stride = 1;
            } else if (l == 0 && !dense0 && dedup && io.fetched_codes) {
                // merged groups on a sharded store: neighbour e's code row is the centre slot of its group
                if (cached) {
                    TRY(scatter_code_rows(io.fetched_codes, io.fetched_index, n_g, io.fetched_valid, io.code_cache, io.group_slot, gdev, G, m.M, s));
                    a.codes = io.code_cache; a.codes_index = io.group_index;
                } else {
                    TRY(nb_code_rows(io.group_index, io.fetched_index, n_g, Tt * kg, b.nb_row, s));
                    a.codes = io.fetched_codes; a.codes_index = b.nb_row;
                }
                a.codes_direct = 1;
                a.row0 = 0; a.n_local = 0; a.M = m.M; a.dsub = m.dsub; a.centroids = m.centroids;
            } else if (l == 0 && !dense0) {
                a.codes = io.fetched_codes ? io.fetched_codes : m.codes;
                if (!io.fetched_codes) a.shards = m.shards;
                a.codes_direct = io.fetched_codes ? (io.fetched_centres_only ? 1 : n_g) : 0;
                a.codes_index = io.fetched_codes ? io.fetched_index : nullptr;
                a.row0 = m.row0; a.n_local = m.n_local; a.M = m.M; a.dsub = m.dsub; a.centroids = m.centroids;
            } else {
                a.X = hn_cur; a.ldx = ld_hn; a.x_group_stride = n_g;
            }
            TRY(star_attn(a, s));
        }
        {   // 2*agg = Z Wvz + has_nb * bvz + m_causal
            GemmParams g{};
            g.A = b.Z; g.lda = (int64_t)H * din; g.W = w.wvz_t; g.ldw = din; g.C = b.ms; g.ldc = d;
            g.bias = w.bvz; g.bias_mode = 1; g.gate = b.has_nb;
            const bool fused = causal_attn_fused_ok(T, dk);
            if (!fused) { g.R = b.mc; g.ldr = d; }
            g.M = (int)Tt; g.N = dk; g.K = din; g.batch1 = H;
            g.sA1 = din; g.sW1 = (int64_t)dk * din; g.sC1 = dk; g.sB1 = dk; g.sR1 = dk;
            TRY(gemm_nt(g, s));
            // recipe shape: the causal branch adds itself into the sum with coalesced row accesses -- cheaper than
            // the residual loads of the GEMM epilogue (same value: star + causal, fp32 addition commutes)
            if (fused) TRY(causal_attn_fused(qp, kp, vp, ldq, b.ms, d, nb, T, H, dk, m.max_intra_context, s, true));
        }
        // a_linear on the cross-type mean (0.5 folded into alpha) + residual, then LayerNorm (hgt.py:397-405)
        TRY(linear(b.ms, d, w.wa_t, w.ba_t, b.aout, Tt, d, d, nullptr, 0.5f, s));
        float* ht_out = last ? io.out_tgt : b.ht[l & 1];
        TRY(layernorm(b.aout, d, w.ln_g_t, w.ln_b_t, ht_out, d, Tt, d, m.ln_eps, nullptr, s, ht_in, d));    // + h (hgt.py:403)

        // ---- ntgt update (only when a later layer -- or the caller -- consumes it)
        if (ntgt && (!last || io.out_ntgt) && S > 0) {
            GNNLM_REQUIRE(w.wq_n && w.wk_n && w.wv_n && w.wa_n && w.ln_g_n && w.ln_b_n, "hgt: layer has null ntgt weights");
            // Only what a later layer reads is computed (outputs identical to the full update): the last layer's star edges
            // read the CENTRE slot of each group, and a slot's update depends on its path neighbours at distance 1
            // (build_ntgt_edges(context=1)) -- so layer l has to deliver the slots within r = n_layers - 2 - l positions of
            // the centre, from K / V of the slots within r + 1.  L = 3, l = r = 2: layer 0 writes 3 of 5 slots (Q, a_linear,
            // LayerNorm on 3/5 of the rows), layer 1 the centre only (Q, a_linear on 1/5, K and V on 3/5): 3.2 of the 8
            // 655360 x 1024 x 1024 GEMMs of the two updates go away.  With out_ntgt (API parity, tests) everything is computed.
            const int rad = io.out_ntgt ? n_g : m.n_layers - 2 - l;
            const bool all_slots = rad >= std::max(m.left, m.right);
            float* hn_out = (last && io.out_ntgt) ? io.out_ntgt : (hn_cur == b.hn[0] ? b.hn[1] : b.hn[0]);
            // projection inputs: the layer's ntgt states, or (layer 0 with folded weights) the decoded rows
            const bool f0 = l == 0 && fold0;
            const float* pin = f0 ? b.hn[1] : hn_cur;
            const int64_t ld_pin = f0 ? dpq : ld_hn;
            const int kin = f0 ? dpq : d;
            const float *Wq = f0 ? w.wq_n0 : w.wq_n, *Bq = f0 ? w.bq_n0 : w.bq_n, *Wk = f0 ? w.wk_n0 : w.wk_n,
                        *Bk = f0 ? w.bk_n0 : w.bk_n, *Wv = f0 ? w.wv_n0 : w.wv_n, *Bv = f0 ? w.bv_n0 : w.bv_n;
            GroupWindows win(G, n_g, 4 * (int64_t)std::max<int64_t>(std::max<int64_t>(ld_pin, ld_hn), std::max(d, dpq)));
            if (gdev) {
                GNNLM_REQUIRE(win.n_windows() <= MAX_WINDOWS, "hgt: too many GEMM windows for a device-side group count");
                TRY(window_counts(gdev, G, win.per, (int)win.n_windows(), b.win_counts, s));
                win.counts = b.win_counts;
            }
            if (all_slots) {
                if (f0) TRY(linear_windows(win, b.hn[1], dpq, m.opq_at, m.opq_nba, b.hn[0], d, dpq, s));      // residual of every row
                TRY(linear_windows(win, pin, ld_pin, Wq, Bq, b.nq, d, kin, s));
                TRY(linear_windows(win, pin, ld_pin, Wk, Bk, b.nk, d, kin, s));
                TRY(linear_windows(win, pin, ld_pin, Wv, Bv, b.nv, d, kin, s));
                ChainAttnParams ca{};
                ca.Q = b.nq; ca.K = b.nk; ca.V = b.nv; ca.ld = d; ca.valid = valid;
                ca.n_groups = G; ca.left = m.left; ca.right = m.right; ca.H = H; ca.dk = dk;
                ca.n_groups_dev = gdev;
                ca.out = b.nq; ca.ldo = d;      // in place over Q: a (group, head) task loads before it stores
                TRY(chain_attn(ca, s));
                TRY(linear_windows(win, b.nq, d, w.wa_n, w.ba_n, b.nk, d, d, s));
                TRY(layernorm(b.nk, d, w.ln_g_n, w.ln_b_n, hn_out, d, S, d, m.ln_eps, valid, s, hn_cur, ld_hn, nullptr, gdev, n_g));
            } else {
                // slot of the position `o` relative to the centre: centre first, then o - left .. o - 1, then o + 1 .. o + right
                auto slot = [&](int o) { return o == 0 ? 0 : (o < 0 ? m.left + 1 + o : m.left + o); };
                int sel_out[8], sel_kv[8], n_out = 0, n_kv = 0;
                for (int o = -std::min(rad, m.left); o <= std::min(rad, m.right); ++o) sel_out[n_out++] = slot(o);
                for (int o = -std::min(rad + 1, m.left); o <= std::min(rad + 1, m.right); ++o) sel_kv[n_kv++] = slot(o);
                TRY(group_rows(b.rows_out, G, n_g, sel_out, n_out, s));
                TRY(group_rows(b.rows_kv, G, n_g, sel_kv, n_kv, s));
                const int64_t R_out = G * n_out;
                if (f0) TRY(linear_rows_windows(win, b.hn[1], dpq, m.opq_at, m.opq_nba, b.hn[0], d, b.rows_out, n_out, d, dpq, s));   // residual rows only
                TRY(linear_rows_windows(win, pin, ld_pin, Wq, Bq, b.nq, d, b.rows_out, n_out, d, kin, s));
                const bool by_row = f0 && b.xu != nullptr;                     // (ABI 11) K / V of layer 0 once per distinct datastore row
                if (by_row) {
                    GroupWindows winu(S, 1, 4 * (int64_t)std::max(d, dpq));
                    GNNLM_REQUIRE(winu.n_windows() <= MAX_WINDOWS, "hgt: too many GEMM windows for the row-keyed projections");
                    TRY(window_counts(b.row_cnt, S, winu.per, (int)winu.n_windows(), b.win_counts_u, s));
                    winu.counts = b.win_counts_u;
                    TRY(linear_windows(winu, b.xu, dpq, Wk, Bk, b.nk, d, kin, s));
                    TRY(linear_windows(winu, b.xu, dpq, Wv, Bv, b.nv, d, kin, s));
                } else {
                    TRY(linear_rows_windows(win, pin, ld_pin, Wk, Bk, b.nk, d, b.rows_kv, n_kv, d, kin, s));
                    TRY(linear_rows_windows(win, pin, ld_pin, Wv, Bv, b.nv, d, b.rows_kv, n_kv, d, kin, s));
                }
                ChainAttnParams ca{};
                ca.Q = b.nq; ca.K = b.nk; ca.V = b.nv; ca.ld = d; ca.valid = valid;
                ca.n_groups = G; ca.left = m.left; ca.right = m.right; ca.H = H; ca.dk = dk;
                ca.out = b.nq; ca.ldo = d;
                ca.radius_p1 = rad + 1;
                ca.n_groups_dev = gdev;
                if (by_row) ca.kv_index = b.slot_u;
                TRY(chain_attn(ca, s));
                TRY(linear_rows_windows(win, b.nq, d, w.wa_n, w.ba_n, b.nk, d, b.rows_out, n_out, d, d, s));
                TRY(layernorm(b.nk, d, w.ln_g_n, w.ln_b_n, hn_out, d, R_out, d, m.ln_eps, valid, s, hn_cur, ld_hn, b.rows_out, gdev, n_out));
            }
            hn_cur = hn_out;
            ld_hn = d;
            // the centre slot (row g * n_g) of every computed group -> its cache slot, for layer l + 1's star edges now and later
            if (cached && !last) TRY(scatter_rows(hn_out, (int64_t)n_g * d, io.state_cache + (int64_t)l * io.cache_cap * d, d, io.group_slot, G, d, s, gdev));
        }
        ht_in = ht_out;
    }
    return OK;
}

struct AsmBufs {
    float *head_part, *head_lse, *head_picked, *xi, *tail_part, *tail_lse, *tail_picked;
    int32_t *head_pick, *band_rows, *band_pick, *band_count;
};

void carve_asm(const gnnlm_adaptive_softmax_t& w, int64_t n, Carver& c, AsmBufs& b) {
    const int nt = w.n_bands - 1;
    const int64_t head_n = w.cutoff[0] + nt;
    int64_t max_size = 0, max_dim = 0;
    for (int i = 1; i < w.n_bands; ++i) {
        max_size = std::max<int64_t>(max_size, w.cutoff[i] - w.cutoff[i - 1]);
        max_dim = std::max<int64_t>(max_dim, w.dim[i]);
    }
    b.head_part = c.take<float>(n * 4 * cdiv(head_n, 128));
    b.head_lse = c.take<float>(n);
    b.head_picked = c.take<float>(n);
    b.head_pick = c.take<int32_t>(n);
    b.band_rows = c.take<int32_t>(std::max(nt, 1) * n);
    b.band_pick = c.take<int32_t>(std::max(nt, 1) * n);
    b.band_count = c.take<int32_t>(8);
    b.xi = c.take<float>(n * max_dim);
    b.tail_part = c.take<float>(n * 4 * cdiv(std::max<int64_t>(max_size, 1), 128));
    b.tail_lse = c.take<float>(n);
    b.tail_picked = c.take<float>(n);
}

int adaptive_impl(const gnnlm_adaptive_softmax_t& w, const float* x, int64_t ldx, const int64_t* target, int64_t n,
                  float* lm_logp, void* ws, size_t ws_bytes, hipStream_t s) {
    GNNLM_REQUIRE(w.n_bands >= 1 && w.n_bands <= 8 && w.head_w && w.d > 0 && w.d % 4 == 0, "adaptive: bad weights");
    GNNLM_REQUIRE(w.gemm_precision >= 0 && w.gemm_precision <= 2, "adaptive: gemm_precision must be 0, 1 or 2");
    GemmPrecisionScope prec_scope(w.gemm_precision);
    GNNLM_REQUIRE(x && target && lm_logp, "adaptive: null io");
    if (n == 0) return OK;
    GNNLM_REQUIRE(n < (1ll << 31), "adaptive: too many rows");
    Carver c(ws, ws_bytes);
    AsmBufs b;
    carve_asm(w, n, c, b);
    GNNLM_REQUIRE(ws && c.fits(), "adaptive: workspace too small (see gnnlm_adaptive_workspace_bytes)");
    const int nt = w.n_bands - 1;
    const int head_n = w.cutoff[0] + nt;

    BandSplitParams bs;
    bs.target = target; bs.n = n; bs.n_bands = w.n_bands;
    for (int i = 0; i < w.n_bands; ++i) bs.cutoff[i] = w.cutoff[i];
    bs.head_pick = b.head_pick; bs.band_rows = b.band_rows; bs.band_pick = b.band_pick; bs.band_count = b.band_count;
    TRY(band_split(bs, s));

    {   // head: [E_0 ; class_proj] (adaptive_softmax.py:24-47,184-188)
        GemmParams g{};
        g.A = x; g.lda = ldx; g.W = w.head_w; g.ldw = w.d;
        g.lse_part = b.head_part; g.lse_pick = b.head_pick; g.lse_picked = b.head_picked;
        g.M = (int)n; g.N = head_n; g.K = w.d;
        // tile walk: bands of 4 m-tiles, n slow inside a band.  An XCD's 32 concurrently running 256 x 256 tiles then form a 4 x 8
        // block (4 A panels + 8 W panels in flight through its L2) instead of a 32 x 1 column (32 + 1): PMC FETCH_SIZE x 2 of the
        // 8192 x 20004 x 1024 head 2.74 GB -> 0.97 GB per launch (tools/pmc_head_order.sh; GM = 8: 1.02, GM = 16: 1.53).  That is
        // the floor of this tile size -- tiles x (1 MiB / 8 + 1 MiB / 4) = 0.95 GB: a panel is 1 MiB and the L2 4 MiB, nothing
        // survives from one block of tiles to the next -- and the time does not move (2.457 ms either way: MFMA-bound)
        if (n >= 1024) g.tile_order = 2 + 4;
        TRY(gemm_nt(g, s));     // logits are reduced in the epilogue, never written
    }
    TRY(lse_reduce(b.head_part, 2 * (int)cdiv(head_n, 128), n, nullptr, b.head_lse, s));
    TRY(head_logp(b.head_picked, b.head_lse, lm_logp, n, s));

    for (int i = 1; i < w.n_bands; ++i) {   // tails (adaptive_softmax.py:91-115,199-203), target rows only
        GNNLM_REQUIRE(w.proj_t[i] && w.emb[i] && w.dim[i] > 0 && w.dim[i] % 4 == 0, "adaptive: bad tail band");
        const int size = w.cutoff[i] - w.cutoff[i - 1];
        const int32_t* rows = b.band_rows + (int64_t)(i - 1) * n;
        const int32_t* pick = b.band_pick + (int64_t)(i - 1) * n;
        const int32_t* cnt = b.band_count + (i - 1);
        GemmParams g{};
        g.A = x; g.lda = ldx; g.a_rows = rows; g.W = w.proj_t[i]; g.ldw = w.d; g.C = b.xi; g.ldc = w.dim[i];
        g.M = (int)n; g.N = w.dim[i]; g.K = w.d; g.m_dev = cnt;
        TRY(gemm_nt(g, s));
        GemmParams t{};
        t.A = b.xi; t.lda = w.dim[i]; t.W = w.emb[i]; t.ldw = w.dim[i];
        t.lse_part = b.tail_part; t.lse_pick = pick; t.lse_picked = b.tail_picked;
        t.M = (int)n; t.N = size; t.K = w.dim[i]; t.m_dev = cnt;
        TRY(gemm_nt(t, s));
        TRY(lse_reduce(b.tail_part, 2 * (int)cdiv(size, 128), n, cnt, b.tail_lse, s));
        TRY(tail_combine(b.tail_picked, b.tail_lse, rows, cnt, n, lm_logp, s));
    }
    return OK;
}

}  // namespace
}  // namespace gnnlm

using namespace gnnlm;

struct gnnlm_store {
    int64_t n_store, row0, n_local;
    int32_t M, vals_itemsize, device;
    uint8_t* codes;
    void* vals;
};

extern "C" {

const char* gnnlm_last_error(void) { return g_last_error.c_str(); }
int gnnlm_abi_version(void) { return GNNLM_ABI_VERSION; }
const char* gnnlm_target_arch(void) { return "gfx950"; }
size_t gnnlm_sizeof(const char* name) {
    if (!name) return 0;
#define GNNLM_SZ(t) if (!strcmp(name, #t)) return sizeof(t);
    GNNLM_SZ(gnnlm_group_assign_t) GNNLM_SZ(gnnlm_gemm_t) GNNLM_SZ(gnnlm_gather_t) GNNLM_SZ(gnnlm_star_attn_t) GNNLM_SZ(gnnlm_chain_attn_t)
    GNNLM_SZ(gnnlm_adaptive_softmax_t) GNNLM_SZ(gnnlm_knn_interp_t) GNNLM_SZ(gnnlm_hgt_layer_t)
    GNNLM_SZ(gnnlm_hgt_t) GNNLM_SZ(gnnlm_hgt_io_t) GNNLM_SZ(gnnlm_profile_entry_t) GNNLM_SZ(gnnlm_topk_t) GNNLM_SZ(gnnlm_ivfpq_scan_t) GNNLM_SZ(gnnlm_ivfpq_scan8_t) GNNLM_SZ(gnnlm_ivfpq_rescore_t) GNNLM_SZ(gnnlm_ivfpq_tau_t) GNNLM_SZ(gnnlm_ivfpq_refine_t) GNNLM_SZ(gnnlm_peer_gather_t) GNNLM_SZ(gnnlm_shards_t)
#undef GNNLM_SZ
    return 0;
}

#define GNNLM_DESC(d)                                       \
    if (!(d)) {                                             \
        set_error("invalid argument: null descriptor");     \
        return E_INVALID;                                   \
    }

int gnnlm_gemm_nt(const gnnlm_gemm_t* d, void* stream) { GNNLM_DESC(d); return gemm_nt(*d, (hipStream_t)stream); }
int gnnlm_group_assign(const gnnlm_group_assign_t* d, void* stream) { GNNLM_DESC(d); return group_assign(*d, (hipStream_t)stream); }
int gnnlm_gather_rows_peer(const gnnlm_peer_gather_t* d, void* stream) { GNNLM_DESC(d); return gather_rows_peer(*d, (hipStream_t)stream); }
int gnnlm_enable_peer_access(int32_t peer_device) {
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return OK; }
    GNNLM_HIP(e);
    return OK;
}
int gnnlm_lse_reduce(const float* part, int32_t n_parts, int64_t rows, const int32_t* m_dev, float* lse, void* stream) {
    return lse_reduce(part, n_parts, rows, m_dev, lse, (hipStream_t)stream);
}
int gnnlm_pq_encode(const float* x, int64_t ldx, const float* centroids, const float* norm2, int32_t M, int32_t dsub,
                    int64_t n, uint8_t* codes, void* stream) {
    return pq_encode(x, ldx, centroids, norm2, M, dsub, n, codes, (hipStream_t)stream);
}
int gnnlm_bucket_rows(const int64_t* rows, int64_t n, int64_t n_store, int64_t rows_per_rank, int32_t world,
                      int32_t self_rank, int64_t* counts, int64_t* cursor, int64_t* send_rows, int32_t* inv, void* stream) {
    return bucket_rows(rows, n, n_store, rows_per_rank, world, self_rank, counts, cursor, send_rows, inv, (hipStream_t)stream);
}
int gnnlm_bucket_rows_padded(const int64_t* rows, int64_t n, int64_t n_store, int64_t rows_per_rank, int32_t world,
                             int64_t cap, int64_t* cursor, int64_t* send_rows, int32_t* inv, int64_t* overflow, void* stream) {
    return bucket_rows_padded(rows, n, n_store, rows_per_rank, world, cap, cursor, send_rows, inv, overflow, (hipStream_t)stream);
}
int gnnlm_pq_gather_decode(const gnnlm_gather_t* d, void* stream) { GNNLM_DESC(d); return gather_decode(*d, (hipStream_t)stream); }
int gnnlm_star_attn(const gnnlm_star_attn_t* d, void* stream) { GNNLM_DESC(d); return star_attn(*d, (hipStream_t)stream); }
int gnnlm_chain_attn(const gnnlm_chain_attn_t* d, void* stream) { GNNLM_DESC(d); return chain_attn(*d, (hipStream_t)stream); }
int gnnlm_causal_softmax(float* S, int64_t n_mats, int32_t T, int64_t ld, int32_t max_ctx, void* stream) {
    return causal_softmax(S, n_mats, T, ld, max_ctx, (hipStream_t)stream);
}
int gnnlm_causal_attn(const float* Q, const float* K, const float* V, int64_t ld, float* out, int64_t ldo,
                      int32_t n_blocks, int32_t T, int32_t H, int32_t dk, int32_t max_ctx, void* stream) {
    return causal_attn_fused(Q, K, V, ld, out, ldo, n_blocks, T, H, dk, max_ctx, (hipStream_t)stream);
}
int gnnlm_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int64_t ldo,
                    int64_t rows, int32_t d, float eps, const uint8_t* valid, void* stream) {
    return layernorm(x, ldx, gamma, beta, out, ldo, rows, d, eps, valid, (hipStream_t)stream);
}
int gnnlm_gelu(float* x, int64_t n, void* stream) { return gelu(x, n, (hipStream_t)stream); }
int gnnlm_filter_neighbors(const int64_t* ids, const int64_t* tok_pos, int64_t n_tok, int32_t kg, int64_t invalid_ctx, int64_t* out,
                           void* stream) {
    return filter_neighbors(ids, tok_pos, n_tok, kg, invalid_ctx, out, (hipStream_t)stream);
}
int gnnlm_half_to_float(const void* src, float* dst, int64_t n, void* stream) {
    return half_to_float(src, dst, n, (hipStream_t)stream);
}
int gnnlm_row_lse_pick(const float* logits, int64_t ld, int64_t rows, const int32_t* m_dev, int32_t n,
                       const int32_t* pick, float* lse, float* picked, void* stream) {
    return row_lse_pick(logits, ld, rows, m_dev, n, pick, lse, picked, (hipStream_t)stream);
}
size_t gnnlm_adaptive_workspace_bytes(const gnnlm_adaptive_softmax_t* w, int64_t n) {
    if (!w) return 0;
    Carver c(nullptr, 0);
    AsmBufs b;
    carve_asm(*w, n, c, b);
    return c.off + 256;
}
int gnnlm_adaptive_target_logp(const gnnlm_adaptive_softmax_t* w, const float* x, int64_t ldx, const int64_t* target,
                               int64_t n, float* lm_logp, void* workspace, size_t workspace_bytes, void* stream) {
    GNNLM_DESC(w);
    return adaptive_impl(*w, x, ldx, target, n, lm_logp, workspace, workspace_bytes, (hipStream_t)stream);
}
int gnnlm_knn_interp(const gnnlm_knn_interp_t* d, void* stream) { GNNLM_DESC(d); return knn_interp(*d, (hipStream_t)stream); }
size_t gnnlm_knn_interp_scratch_bytes(int64_t n, int32_t k, int64_t n_local) { return knn_interp_scratch_bytes(n, k, n_local); }
int gnnlm_label_tags(const void* vals, int32_t vals_itemsize, int64_t n, uint8_t* tag, void* stream) { return label_tags(vals, vals_itemsize, n, tag, (hipStream_t)stream); }
int gnnlm_topk_merge(const gnnlm_topk_t* d, void* stream) { GNNLM_DESC(d); return topk_merge(*d, (hipStream_t)stream); }
int gnnlm_ivfpq_scan(const gnnlm_ivfpq_scan_t* d, void* stream) { GNNLM_DESC(d); return ivfpq_scan(*d, (hipStream_t)stream); }
int gnnlm_ivfpq_pack_codes(const uint8_t* codes, int64_t N, int32_t M, uint8_t* out, void* stream) {
    return ivfpq_pack_codes(codes, N, M, out, (hipStream_t)stream);
}
int gnnlm_ivfpq_pack_lut(const float* lut, int64_t ld_lut, int64_t n, int32_t M, float* out, void* stream) {
    return ivfpq_pack_lut(lut, ld_lut, n, M, out, (hipStream_t)stream);
}
int gnnlm_ivfpq_pack_tiles(const uint8_t* codes, int64_t N, int32_t M, uint8_t* out, void* stream) {
    return ivfpq_pack_tiles(codes, N, M, out, (hipStream_t)stream);
}
int gnnlm_ivfpq_build_groups(const int64_t* probe_list, int64_t ld_probe, int64_t n, int32_t P, int32_t nlist, int64_t seg, int32_t* grp_list,
                             int32_t* grp_q, int64_t* grp_out, int32_t* n_groups, int32_t* scratch, void* stream) {
    return ivfpq_build_groups(probe_list, ld_probe, n, P, nlist, seg, grp_list, grp_q, grp_out, n_groups, scratch, (hipStream_t)stream);
}
int gnnlm_ivfpq_quantize_lut(const float* lut, int64_t ld_lut, int64_t n, int32_t M, uint8_t* qlut, float* qmeta, void* stream) {
    return ivfpq_quantize_lut(lut, ld_lut, n, M, qlut, qmeta, (hipStream_t)stream);
}
int gnnlm_ivfpq_scan8(const gnnlm_ivfpq_scan8_t* d, void* stream) { GNNLM_DESC(d); return ivfpq_scan8(*d, (hipStream_t)stream); }
int gnnlm_ivfpq_rescore(const gnnlm_ivfpq_rescore_t* d, void* stream) { GNNLM_DESC(d); return ivfpq_rescore(*d, (hipStream_t)stream); }
int gnnlm_ivfpq_tau(const gnnlm_ivfpq_tau_t* d, void* stream) { GNNLM_DESC(d); return ivfpq_tau(*d, (hipStream_t)stream); }
int gnnlm_ivfpq_refine(const gnnlm_ivfpq_refine_t* d, void* stream) { GNNLM_DESC(d); return ivfpq_refine(*d, (hipStream_t)stream); }
int gnnlm_ivfpq_split_payload(int64_t* idx, int64_t n, int32_t label_bits, int32_t val_last, int32_t* out_vals, void* stream) { return ivfpq_split_payload(idx, n, label_bits, val_last, out_vals, (hipStream_t)stream); }
int gnnlm_masked_sum_f64(const float* x, const uint8_t* mask, int64_t n, double* out, void* stream) {
    return masked_sum_f64(x, mask, n, out, (hipStream_t)stream);
}

size_t gnnlm_hgt_workspace_bytes(const gnnlm_hgt_t* m, const gnnlm_hgt_io_t* io) {
    if (!m || !io) return 0;
    Carver c(nullptr, 0);
    HgtBufs b;
    carve_hgt(*m, *io, c, b);
    return c.off + 256;
}
int gnnlm_hgt_forward(const gnnlm_hgt_t* m, const gnnlm_hgt_io_t* io, void* workspace, size_t workspace_bytes,
                      void* stream) {
    GNNLM_DESC(m);
    GNNLM_DESC(io);
    return hgt_forward_impl(*m, *io, workspace, workspace_bytes, (hipStream_t)stream);
}

static const char* kKernelNames[K_COUNT] = {"gemm_nt_f32_kernel", "gather_decode_kernel", "star_attn_kernel",
                                            "chain_attn_kernel", "causal_attn_kernel", "layernorm_kernel",
                                            "row_lse_pick_kernel", "knn_interp_kernel", "misc", "split_planes_kernel",
                                            "topk_merge_kernel", "ivfpq_scan_kernel", "ivfpq_scan8_kernel", "ivfpq_rescore_kernel", "ivfpq_sums_kernel", "ivfpq_tau_kernel"};
const char* gnnlm_kernel_name(int32_t kernel_id) {
    return kernel_id >= 0 && kernel_id < K_COUNT ? kKernelNames[kernel_id] : nullptr;
}
int gnnlm_profile_begin(uint32_t kernel_mask) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    GNNLM_REQUIRE(g_prof_recs.empty(), "profile_begin: a profile is already open");
    g_prof_mask = kernel_mask;
    return OK;
}
int gnnlm_profile_end(gnnlm_profile_entry_t* out, int32_t n_max, int32_t* n_out) {
    GNNLM_REQUIRE(out && n_out && n_max >= K_COUNT, "profile_end: need room for every kernel id");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_mask = 0;
    for (int k = 0; k < K_COUNT; ++k) out[k] = gnnlm_profile_entry_t{k, 0, 0.0, 0.0, 0.0};
    for (auto& r : g_prof_recs) {
        GNNLM_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        GNNLM_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        double scale = 1.0;
        if (r.host_scale) {
            scale = std::min(1.0, (double)*r.host_scale / r.den);
            g_pinned_pool.push_back(r.host_scale);
        }
        out[r.kid].launches += 1;
        out[r.kid].total_ms += ms;
        out[r.kid].flops += r.flops * scale;
        out[r.kid].bytes += r.bytes * scale;
        g_event_pool.push_back(r.a);
        g_event_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    *n_out = K_COUNT;
    return OK;
}

int gnnlm_store_create(int64_t n_store, int64_t row0, int64_t n_local, int32_t M, int32_t vals_itemsize,
                       int32_t device, gnnlm_store_t** out) {
    GNNLM_REQUIRE(out && n_store > 0 && row0 >= 0 && n_local >= 0 && row0 + n_local <= n_store && M > 0,
                  "store_create: bad shape");
    GNNLM_REQUIRE(vals_itemsize == 2 || vals_itemsize == 4, "store_create: vals must be int16 or int32");
    GNNLM_HIP(hipSetDevice(device));
    gnnlm_store* s = new gnnlm_store{n_store, row0, n_local, M, vals_itemsize, device, nullptr, nullptr};
    // an empty shard (more ranks than rows) keeps a 1-row allocation so that the pointers stay non-null
    hipError_t e = hipMalloc((void**)&s->codes, (size_t)std::max<int64_t>(n_local, 1) * M);
    if (e == hipSuccess) e = hipMalloc(&s->vals, (size_t)std::max<int64_t>(n_local, 1) * vals_itemsize);
    if (e != hipSuccess) {
        if (s->codes) (void)hipFree(s->codes);
        delete s;
        set_error(std::string("store_create: hipMalloc failed: ") + hipGetErrorString(e));
        return E_NOMEM;
    }
    *out = s;
    return OK;
}
int gnnlm_store_upload_codes(gnnlm_store_t* s, const uint8_t* host, int64_t first, int64_t n, void* stream) {
    GNNLM_REQUIRE(s && host && first >= 0 && n >= 0 && first + n <= s->n_local, "store_upload_codes: bad range");
    GNNLM_HIP(hipMemcpyAsync(s->codes + first * s->M, host, (size_t)n * s->M, hipMemcpyHostToDevice, (hipStream_t)stream));
    GNNLM_HIP(hipStreamSynchronize((hipStream_t)stream));
    return OK;
}
int gnnlm_store_upload_vals(gnnlm_store_t* s, const void* host, int64_t first, int64_t n, void* stream) {
    GNNLM_REQUIRE(s && host && first >= 0 && n >= 0 && first + n <= s->n_local, "store_upload_vals: bad range");
    GNNLM_HIP(hipMemcpyAsync((char*)s->vals + first * s->vals_itemsize, host, (size_t)n * s->vals_itemsize,
                             hipMemcpyHostToDevice, (hipStream_t)stream));
    GNNLM_HIP(hipStreamSynchronize((hipStream_t)stream));
    return OK;
}
const uint8_t* gnnlm_store_codes(const gnnlm_store_t* s) { return s ? s->codes : nullptr; }
const void* gnnlm_store_vals(const gnnlm_store_t* s) { return s ? s->vals : nullptr; }
int gnnlm_store_destroy(gnnlm_store_t* s) {
    if (!s) return OK;
    if (s->codes) (void)hipFree(s->codes);
    if (s->vals) (void)hipFree(s->vals);
    delete s;
    return OK;
}

}  // extern "C"

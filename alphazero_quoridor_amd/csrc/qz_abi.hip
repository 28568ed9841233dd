// qz_abi.hip -- host side of libqzero_hip.so: the C ABI declared in include/qz_abi.h.
// Owns the engine's HBM arenas, validates arguments, launches the kernels of
// qz_kernels.hip on the caller's stream.  No CPU compute path exists: without a HIP
// device every entry point reports QZ_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/qz_abi.h"
#include "qz_device.h"

namespace qzl {
hipError_t movegen_encode(const uint64_t*, const uint64_t*, const uint64_t*, int, uint32_t*, float*, const uint8_t*, void*, const RulesOpts&, hipStream_t, const int* n_dev = nullptr);
hipError_t advance(const EngineDev&, int, unsigned int, int, int, hipStream_t);
hipError_t advance_lanes(const EngineDev&, int, unsigned int, int, hipStream_t);
hipError_t moves(const EngineDev&, unsigned int, hipStream_t);
hipError_t round_tail(const EngineDev&, int, hipStream_t);
hipError_t memo_flush(const EngineDev&, hipStream_t);
size_t movegen_scratch_bytes(int);
hipError_t step(uint64_t*, uint64_t*, uint64_t*, const uint8_t*, int, uint8_t*, uint8_t*, hipStream_t);
hipError_t select(const EngineDev&, hipStream_t);
hipError_t expand_backup(const EngineDev&, const float*, const float*, hipStream_t);
hipError_t expand_backup_select(const EngineDev&, const float*, const float*, hipStream_t);
hipError_t root_pi(const EngineDev&, double*, int32_t*, hipStream_t);
hipError_t root_children(const EngineDev&, int32_t*, double*, float*, int32_t*, hipStream_t);
hipError_t update_with_move(const EngineDev&, const uint8_t*, hipStream_t);
hipError_t finish_move(const EngineDev&, const uint8_t*, float*, uint8_t*, hipStream_t);
hipError_t reset(const EngineDev&, int, hipStream_t);
hipError_t pool_init(const EngineDev&, hipStream_t);
hipError_t harvest(const EngineDev&, uint64_t*, uint64_t*, uint64_t*, float*, float*, int32_t*, int32_t*, long long, hipStream_t);
hipError_t sqrt_table(double*, int, hipStream_t);
hipError_t conv3x3_norm(const float*, const void*, const float*, const float*, const float*, float*, long long, const float*, int, float, hipStream_t);
struct TrunkInput {  // qz_conv.hip
    const uint64_t *hb, *vb, *meta;
    const uint8_t* terminal;
    const float *hot9, *base0, *wd, *gamma0, *beta0;
};
hipError_t trunk(float*, float*, long long, int, const void* const*, const float* const*, const float* const*, const float*, float, int, hipStream_t,
                 const void*, const float*, const float*, float*, const TrunkInput*, const int* n_live = nullptr, int single_product = 0);
hipError_t head_fc(const float*, long long, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, hipStream_t,
                   const int* n_live = nullptr);
hipError_t rollout_begin(const uint64_t*, const uint64_t*, const uint64_t*, int, uint8_t*, uint8_t*, int8_t*, int*, hipStream_t);
hipError_t rollout_step(uint64_t*, uint64_t*, uint64_t*, const uint32_t*, int, const uint8_t*, uint8_t*, int8_t*, int*, uint64_t, int, int, hipStream_t);
hipError_t instnorm_act(const float*, const float*, const float*, const float*, float*, long long, int, int, float, hipStream_t);
hipError_t instnorm_act_nhwc(const float*, const float*, const float*, const float*, float*, long long, int, int, float, hipStream_t);
hipError_t input_layer(const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, long long, const float*, const float*,
                       const float*, const float*, const float*, float*, float, hipStream_t);
hipError_t head(const float*, long long, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                const float*, const float*, float*, float*, float, hipStream_t);
}  // namespace qzl

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(QZ_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#include <map>
#include <mutex>
#include <utility>
// Scratch for the stateless rules entry points (board records + path tables of the pooled
// move-generation kernels): one cached allocation per device, grown on demand.  Growing
// calls hipMalloc, so the first call of a given size must happen outside graph capture.
// Scratch of the stateless rules entry points (the pooled pipeline hands path records from its
// first to its second launch through HBM): one buffer per (device, stream), so calls issued on
// different streams never share it.  Engines own theirs.
static std::mutex g_scratch_mu;
struct ScratchBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
static std::map<std::pair<int, void*>, ScratchBuf> g_scratch;
static int get_scratch(int n, void* stream, void** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    size_t need = qzl::movegen_scratch_bytes(n);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchBuf& sb = g_scratch[std::make_pair(dev, stream)];
    if (sb.bytes < need) {
        if (sb.p) {
            HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(sb.p);
            sb.p = nullptr;
            sb.bytes = 0;
        }
        hipError_t e = hipMalloc(&sb.p, need);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            sb.p = nullptr;
            return fail(QZ_E_OOM, "hipMalloc(%zu) for move-generation scratch failed: %s", need, hipGetErrorString(e));
        }
        sb.bytes = need;
    }
    *out = sb.p;
    return 0;
}

static RulesOpts rules_opts(const qz_rules_opts* o) {
    RulesOpts r;  // the defaults
    if (!o) return r;
    r.variant = o->variant;
    if (o->detour_pooled > 0) r.detour_pooled = o->detour_pooled - 1;
    if (o->detour_wave > 0) r.detour_wave = o->detour_wave - 1;
    if (o->enc_split_pct > 0) r.enc_split_pct = o->enc_split_pct > 100 ? 100 : o->enc_split_pct;
    return r;
}

struct qz_engine {
    qz_config cfg;
    EngineDev dev;
    RulesOpts rules;
    void* scratch = nullptr;
    float* feat = nullptr;     // [B][486] head features of the miss list (qz_selfplay_evaluate)
    EngineDev* dev_mem = nullptr;  // `dev` once more in device memory (k_advance's rarely-run paths read it from there)
    int par = 0;               // which of the two miss counters the round in progress uses
    bool async_moves = false;  // qz_selfplay_advance has run with auto_finish: boards may hold a move whose subtree copy is not done
    // qz_selfplay_round: the rules op of the miss list and the finished boards' moves run beside the network's trunk
    hipStream_t side = nullptr, side2 = nullptr;  // (the rules op / the moves: each on a stream of its own beside the network)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
    hipStream_t lanes = nullptr;           // k_rows (select_opts bit 5) runs beside k_advance on a stream of its own
    hipEvent_t ev_lfork = nullptr, ev_ljoin = nullptr;
    size_t memo_small_bytes = 0, memo_big_bytes = 0;
    unsigned flushes = 0;
    std::vector<void*> allocs;
    int64_t bytes = 0;
};

static int device_check() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(QZ_E_NO_DEVICE, "no HIP device visible: libqzero_hip has no CPU path");
    }
    return 0;
}

template <typename T>
static int dev_alloc(qz_engine* e, T** out, size_t count) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t err = hipMalloc(&p, bytes);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return fail(QZ_E_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(err));
    }
    e->allocs.push_back(p);
    e->bytes += (int64_t)bytes;
    *out = (T*)p;
    return 0;
}

extern "C" {

int qz_version(void) { return QZ_ABI_VERSION; }
const char* qz_last_error(void) { return g_err; }
int qz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ------------------------------------------------------------------ stateless kernels
static int check_boards(const qz_boards* b, int n) {
    if (n < 0) return fail(QZ_E_INVALID, "n < 0");
    if (n > 0 && (!b || !b->hbits || !b->vbits || !b->meta)) return fail(QZ_E_INVALID, "null board arrays");
    return 0;
}

int qz_movegen_encode_opts(const qz_boards* boards, int n, uint32_t* mask5, float* planes, const qz_rules_opts* opts, void* stream) {
    int r;
    if ((r = device_check()) || (r = check_boards(boards, n))) return r;
    if (n > 0 && !mask5 && !planes) return fail(QZ_E_INVALID, "mask5 and planes are both null");
    if (n == 0) return 0;
    void* scratch = nullptr;
    if ((r = get_scratch(n, stream, &scratch))) return r;
    HIP_TRY(qzl::movegen_encode(boards->hbits, boards->vbits, boards->meta, n, mask5, planes, nullptr, scratch, rules_opts(opts),
                                (hipStream_t)stream));
    return 0;
}
int qz_movegen(const qz_boards* boards, int n, uint32_t* mask5, void* stream) {
    if (n > 0 && !mask5) return fail(QZ_E_INVALID, "mask5 is null");
    return qz_movegen_encode_opts(boards, n, mask5, nullptr, nullptr, stream);
}
int qz_encode(const qz_boards* boards, int n, float* planes, void* stream) {
    if (n > 0 && !planes) return fail(QZ_E_INVALID, "planes is null");
    return qz_movegen_encode_opts(boards, n, nullptr, planes, nullptr, stream);
}
int qz_movegen_encode(const qz_boards* boards, int n, uint32_t* mask5, float* planes, void* stream) {
    if (n > 0 && (!mask5 || !planes)) return fail(QZ_E_INVALID, "mask5/planes is null");
    return qz_movegen_encode_opts(boards, n, mask5, planes, nullptr, stream);
}
int qz_step(qz_boards* boards, const uint8_t* action, int n, uint8_t* done, uint8_t* winner, void* stream) {
    int r;
    if ((r = device_check()) || (r = check_boards(boards, n))) return r;
    if (n > 0 && !action) return fail(QZ_E_INVALID, "action is null");
    if (n == 0) return 0;
    HIP_TRY(qzl::step(boards->hbits, boards->vbits, boards->meta, action, n, done, winner, (hipStream_t)stream));
    return 0;
}

// scratch layout of qz_rollout: mask5 [n][5] u32 | player0 [n] u8 | done [n] u8 | n_done int (16-byte aligned blocks)
static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
int64_t qz_rollout_scratch_bytes(int n) {
    if (n < 0) return 0;
    return (int64_t)(align16((size_t)n * 20) + 2 * align16((size_t)n) + 16 + qzl::movegen_scratch_bytes(n));
}
int qz_rollout(qz_boards* boards, int n, int limit, uint64_t seed, int8_t* value, void* scratch, void* stream) {
    int r;
    if ((r = device_check()) || (r = check_boards(boards, n))) return r;
    if (n == 0) return 0;
    if (!value || !scratch || limit <= 0) return fail(QZ_E_INVALID, "value / scratch is null or limit <= 0");
    hipStream_t s = (hipStream_t)stream;
    uint8_t* base = (uint8_t*)scratch;
    uint32_t* mask5 = (uint32_t*)base;
    uint8_t* player0 = base + align16((size_t)n * 20);
    uint8_t* done = player0 + align16((size_t)n);
    int* n_done = (int*)(done + align16((size_t)n));
    void* mg_scratch = (uint8_t*)n_done + 16;
    const RulesOpts ro;
    HIP_TRY(qzl::rollout_begin(boards->hbits, boards->vbits, boards->meta, n, player0, done, value, n_done, s));
    for (int step = 0; step < limit; step++) {
        // actions() of every board still playing (finished ones are skipped through the terminal flags)
        HIP_TRY(qzl::movegen_encode(boards->hbits, boards->vbits, boards->meta, n, mask5, nullptr, done, mg_scratch, ro, s));
        HIP_TRY(qzl::rollout_step(boards->hbits, boards->vbits, boards->meta, mask5, n, player0, done, value, n_done, seed, step, limit, s));
        if ((step & 15) == 15 || step == limit - 1) {
            int h = 0;
            HIP_TRY(hipMemcpyAsync(&h, n_done, sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (h >= n) break;
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// ------------------------------------------------------------------ engine
int qz_engine_destroy(qz_engine* e) {
    if (!e) return 0;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->side2) (void)hipStreamDestroy(e->side2);
    if (e->lanes) (void)hipStreamDestroy(e->lanes);
    if (e->ev_lfork) (void)hipEventDestroy(e->ev_lfork);
    if (e->ev_ljoin) (void)hipEventDestroy(e->ev_ljoin);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->ev_join2) (void)hipEventDestroy(e->ev_join2);
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
    return 0;
}

int qz_engine_create(const qz_config* cfg, qz_engine** out) {
    int r;
    if (!cfg || !out) return fail(QZ_E_INVALID, "null cfg/out");
    *out = nullptr;
    if ((r = device_check())) return r;
    if (cfg->n_boards <= 0 || cfg->n_playout <= 0) return fail(QZ_E_INVALID, "n_boards and n_playout must be > 0");
    if (!(cfg->temp > 0.f)) return fail(QZ_E_INVALID, "temp must be > 0");
    if (cfg->device < 0 || cfg->device >= qz_device_count()) return fail(QZ_E_INVALID, "device %d out of range", cfg->device);
    HIP_TRY(hipSetDevice(cfg->device));
    qz_engine* e = new (std::nothrow) qz_engine();
    if (!e) return fail(QZ_E_OOM, "host allocation failed");
    e->cfg = *cfg;
    qz_config& c = e->cfg;
    // Trees: a shared pool of 64-KB pages (include/qz_abi.h, qz_device.h).  Measured on round-1
    // runs at n_playout = 400: a tree peaks at 65 k edges early in the game (400 new nodes x <= 131
    // edges per ply on top of the kept subtree) and holds 10-20 k in the long late game; during a
    // re-root the old and the new tree coexist.  250 edges per playout per board on average
    // (3.2 MB at n_playout = 400) covers the lock-step start from the opening, where every
    // board peaks at once; qz_stats.tree_pages_peak reports what a run really used.
    if (c.node_cap < 0) c.node_cap = 0;
    const int edge_reach = QZ_TREE_PT * (int)QZ_PAGE_EDGES;
    if (c.edge_cap <= 0 || c.edge_cap > edge_reach) c.edge_cap = edge_reach;
    if (c.max_plies < 0) c.max_plies = 0;
    if (c.tree_pool_pages <= 0) {
        long long per = (250LL * c.n_playout + QZ_PAGE_EDGES - 1) / QZ_PAGE_EDGES;
        if (per < 4) per = 4;
        long long tot = per * c.n_boards;
        // few boards share little: one tree alone may fill its whole page table, twice during a re-root
        if (tot < 2 * QZ_TREE_PT + 8) tot = 2 * QZ_TREE_PT + 8;
        if (tot > (1LL << 21) - 1) tot = (1LL << 21) - 1;  // physical edge indices are 32 bit
        c.tree_pool_pages = (int)tot;
    }
    if (c.tree_pool_pages > (1 << 21) - 1) return (delete e, fail(QZ_E_INVALID, "tree_pool_pages > 2^21 - 1 (32-bit edge indices)"));
    if (c.traj_pool_pages <= 0) {
        // a page holds ~1,000 late-game plies; reference-faithful games have a median of ~5,000 plies and a tail beyond 250,000
        // (profiles/round3/game_length_400playouts.json): 64 pages per board (4 MB) up to 8,192 boards, 16 beyond
        long long tot = (c.n_boards <= 8192 ? 64LL : 16LL) * c.n_boards;
        if (tot > 0x7fffffffLL) tot = 0x7fffffffLL;
        c.traj_pool_pages = (int)tot;
    }
    if (c.traj_page_dwords <= 0) c.traj_page_dwords = QZ_TPAGE_DWORDS;
    if (c.traj_page_dwords < 256) return (delete e, fail(QZ_E_INVALID, "traj_page_dwords must be >= 256"));
    e->rules = rules_opts(&c.rules);
    if (c.dirichlet_alpha <= 0.f) c.dirichlet_alpha = 0.3f;
    EngineDev& d = e->dev;
    memset(&d, 0, sizeof(d));
    d.n_boards = c.n_boards;
    d.node_cap = c.node_cap;
    d.edge_cap = c.edge_cap;
    d.max_plies = c.max_plies;
    d.n_playout = c.n_playout;
    if (c.max_depth < 0) c.max_depth = 0;
    d.max_depth = c.max_depth;
    if (c.compact_edges == 0) {  // a third of a board's share of the page pool (garbage + live tree + the copy's target must fit), 1..24 pages
        long long pages = (long long)c.tree_pool_pages / c.n_boards / 3;
        pages = pages < 1 ? 1 : (pages > 24 ? 24 : pages);
        c.compact_edges = (int)pages * (int)QZ_PAGE_EDGES;
    }
    if (c.node_cap > 0 || c.edge_cap < edge_reach) c.compact_edges = -1;  // the caps count live nodes: every move compacts
    if (c.compact_edges > edge_reach / 2) c.compact_edges = edge_reach / 2;  // (a ply may add 400 x 131 edges on top)
    d.compact_edges = c.compact_edges;
    d.tree_pool_pages = c.tree_pool_pages;
    d.traj_pool_pages = c.traj_pool_pages;
    d.traj_page_dwords = (uint32_t)c.traj_page_dwords;
    d.c_puct = c.c_puct;
    d.temp = c.temp;
    d.dirichlet_alpha = c.dirichlet_alpha;
    d.noise_frac = c.noise_frac;
    d.seed = c.seed;
    d.is_selfplay = c.is_selfplay;
    d.fix_terminal_sign = c.fix_terminal_sign;
    d.select_opts = c.select_opts;
    const size_t B = (size_t)c.n_boards;
    int rc = 0;
#define ALLOC(field, count) \
    if (!rc) rc = dev_alloc(e, &d.field, (count))
    ALLOC(root_hb, B);
    ALLOC(root_vb, B);
    ALLOC(root_meta, B);
    ALLOC(leaf_hb, B);
    ALLOC(leaf_vb, B);
    ALLOC(leaf_meta, B);
    ALLOC(leaf_mask, B * 5);
    ALLOC(leaf_pedge, B);
    ALLOC(leaf_term, B);
    ALLOC(edge_pool, (size_t)c.tree_pool_pages * QZ_PAGE_EDGES);
    ALLOC(tree_ptab, 2 * B * QZ_TREE_PT);
    ALLOC(tree_npages, 2 * B);
    ALLOC(free_tree, (size_t)c.tree_pool_pages);
    ALLOC(traj_pool, (size_t)c.traj_pool_pages * (size_t)c.traj_page_dwords);
    ALLOC(traj_ptab, B * QZ_TRAJ_PT);
    ALLOC(traj_npages, B);
    ALLOC(traj_cursor, B);
    ALLOC(free_traj, (size_t)c.traj_pool_pages);
    ALLOC(pool_words, (size_t)QZ_P_COUNT);
    ALLOC(release, B);
    ALLOC(root_eoff, B);
    ALLOC(root_ne, B);
    ALLOC(path_edges, B * (size_t)(QZ_PATH_RECS + 1) * QZ_PATH_CAP);
    ALLOC(path_blocks, B * (size_t)(QZ_PATH_RECS + 1) * QZ_PATH_CAP);
    ALLOC(path_len, B);
    ALLOC(rec_len, B * QZ_PATH_RECS);
    ALLOC(rec_stamp, B * QZ_PATH_RECS);
    ALLOC(rec_last, B);
    ALLOC(rec_clock, B);
    ALLOC(tree_half, B);
    ALLOC(n_nodes, B);
    ALLOC(n_edges, B);
    ALLOC(root_N, B);
    ALLOC(ply, B);
    ALLOC(game_serial, B);
    ALLOC(harvest_off, B);
    ALLOC(harvest_gid, B);
    ALLOC(status, B);
    ALLOC(winner, B);
    ALLOC(counters, (size_t)QZ_C_TOTAL);
    ALLOC(drop_log, (size_t)QZ_DROP_LOG * 4);
    ALLOC(bc_playouts, B);
    ALLOC(bc_terminal, B);
    ALLOC(bc_overflow, B);
    ALLOC(bc_nonfinite, B);
    ALLOC(bc_maxdepth, B);
    ALLOC(bc_levels, B);
    ALLOC(bc_scanned, B);
    ALLOC(bc_expanded, B);
    ALLOC(pl_done, B);
    ALLOC(pend_slot, B);
    ALLOC(reroot_pend, B);
    ALLOC(compact_state, B * 8);
    ALLOC(compact_at, B);
    ALLOC(miss_count, (size_t)8);
    ALLOC(miss_hb, B);
    ALLOC(miss_vb, B);
    ALLOC(miss_meta, B);
    ALLOC(miss_mask, B * 5);
    ALLOC(miss_p, B * QZ_N_ACT);
    ALLOC(miss_v, B);
    ALLOC(bc_memo_hits, B);
    ALLOC(bc_evals, B);
    ALLOC(bc_open_rounds, B);
    ALLOC(bc_open_plies, B);
    ALLOC(rows_list, B + 2);
    ALLOC(memo.epoch, (size_t)1);
    if (!rc) rc = dev_alloc(e, &e->feat, B * 486);
    if (!rc) rc = dev_alloc(e, &e->dev_mem, (size_t)1);
    // memo tables: powers of two; auto = 4,096 small + 512 big entries per board
    if (c.memo_small_log2 >= 0 && c.memo_big_log2 >= 0) {
        auto log2_ceil = [](unsigned long long x) { int l = 0; while ((1ull << l) < x) l++; return l; };
        if (c.memo_small_log2 == 0) {  // 16,384 entries per board (2 MB), at most 2^24 buckets (8 GB)
            c.memo_small_log2 = log2_ceil((unsigned long long)B * 16384ull / QZ_MEMO_S_WAYS);
            if (c.memo_small_log2 > 24) c.memo_small_log2 = 24;
        }
        if (c.memo_big_log2 == 0) c.memo_big_log2 = log2_ceil((unsigned long long)B * 512ull / QZ_MEMO_B_WAYS);
        if (c.memo_small_log2 > 31 || c.memo_big_log2 > 31) rc = rc ? rc : fail(QZ_E_INVALID, "memo table too large");
        const size_t sb = ((size_t)1 << c.memo_small_log2), bb = ((size_t)1 << c.memo_big_log2);
        e->memo_small_bytes = sb * QZ_MEMO_S_WAYS * QZ_MEMO_S_DW * 4;
        e->memo_big_bytes = bb * QZ_MEMO_B_WAYS * QZ_MEMO_B_DW * 4;
        ALLOC(memo.small, sb * QZ_MEMO_S_WAYS * QZ_MEMO_S_DW);
        ALLOC(memo.big, bb * QZ_MEMO_B_WAYS * QZ_MEMO_B_DW);
        d.memo.small_mask = (uint32_t)(sb - 1);
        d.memo.big_mask = (uint32_t)(bb - 1);
    } else {
        c.memo_small_log2 = c.memo_big_log2 = -1;
    }
#undef ALLOC
    if (!rc) {
        uint8_t* sc = nullptr;
        rc = dev_alloc(e, &sc, qzl::movegen_scratch_bytes(c.n_boards));
        e->scratch = sc;
    }
    if (rc) {
        qz_engine_destroy(e);
        return rc;
    }
    hipError_t he = hipMemset(d.tree_half, 0, B);
    if (he == hipSuccess) he = hipMemset(d.game_serial, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.counters, 0, QZ_C_TOTAL * sizeof(unsigned long long));
    if (he == hipSuccess) he = hipMemset(d.leaf_term, 0, B);
    if (he == hipSuccess) he = hipMemset(d.path_len, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.rec_len, 0, B * QZ_PATH_RECS * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.rec_stamp, 0, B * QZ_PATH_RECS * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.rec_last, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.rec_clock, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_playouts, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_terminal, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_overflow, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_nonfinite, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_maxdepth, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.tree_npages, 0, 2 * B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.traj_npages, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.traj_cursor, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.release, 0, B);
    if (he == hipSuccess) he = qzl::pool_init(d, nullptr);
    if (he == hipSuccess) he = hipMemset(d.bc_levels, 0, B * sizeof(unsigned long long));
    if (he == hipSuccess) he = hipMemset(d.bc_scanned, 0, B * sizeof(unsigned long long));
    if (he == hipSuccess) he = hipMemset(d.bc_expanded, 0, B * sizeof(unsigned long long));
    if (he == hipSuccess) he = hipMemset(d.leaf_mask, 0, B * 5 * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.miss_count, 0, 8 * sizeof(int));
    if (he == hipSuccess) he = hipMemset(d.reroot_pend, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.compact_state, 0, B * 8 * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_memo_hits, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_evals, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_open_rounds, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.bc_open_plies, 0, B * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMemset(d.rows_list, 0, (B + 2) * sizeof(uint32_t));
    if (he == hipSuccess) {
        const uint32_t one = 1u;  // epoch 0 never matches: an all-zero entry is dead
        he = hipMemcpy(d.memo.epoch, &one, sizeof(one), hipMemcpyHostToDevice);
    }
    if (he == hipSuccess && d.memo.small) he = hipMemset(d.memo.small, 0, e->memo_small_bytes);
    if (he == hipSuccess && d.memo.big) he = hipMemset(d.memo.big, 0, e->memo_big_bytes);
    if (he == hipSuccess) he = qzl::reset(d, 1, nullptr);
    if (he == hipSuccess) he = hipMemcpy(e->dev_mem, &d, sizeof(d), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) {
        qz_engine_destroy(e);
        return fail(QZ_E_HIP, "engine init failed: %s", hipGetErrorString(he));
    }
    *out = e;
    return 0;
}

#define ENGINE_CHECK(e)                                          \
    do {                                                         \
        if (!(e)) return fail(QZ_E_INVALID, "null engine");      \
        HIP_TRY(hipSetDevice((e)->cfg.device));                  \
    } while (0)
// the lock-step tree entry points: not on an engine whose boards play on their own clocks (a board may be waiting for an
// evaluation, or hold a move whose subtree copy is left for its next launch -- its root fields still describe the old tree)
#define LOCKSTEP_CHECK(e)                                                                                                   \
    do {                                                                                                                    \
        ENGINE_CHECK(e);                                                                                                    \
        if ((e)->async_moves)                                                                                               \
            return fail(QZ_E_INVALID, "engine is in asynchronous self-play (qz_selfplay_advance with auto_finish): "       \
                                      "reset it (qz_engine_reset / qz_engine_set_boards with reset_trees) before a lock-step call"); \
    } while (0)

// the round state of the asynchronous loop that lives outside k_reset's per-board words: both miss counters (a reset
// between qz_selfplay_advance and qz_selfplay_round_tail would leave one non-zero with every pend_slot cleared) and the
// host-side parity
static hipError_t reset_round_state(qz_engine* e, hipStream_t s) {
    e->async_moves = false;
    e->par = 0;
    return hipMemsetAsync(e->dev.miss_count, 0, 8 * sizeof(int), s);
}

int qz_engine_reset(qz_engine* e, void* stream) {
    ENGINE_CHECK(e);
    HIP_TRY(qzl::reset(e->dev, 1, (hipStream_t)stream));
    HIP_TRY(reset_round_state(e, (hipStream_t)stream));
    return 0;
}

int qz_engine_set_boards(qz_engine* e, const qz_boards* src, int reset_trees, void* stream) {
    ENGINE_CHECK(e);
    int r;
    if ((r = check_boards(src, e->cfg.n_boards))) return r;
    hipStream_t s = (hipStream_t)stream;
    size_t nb = (size_t)e->cfg.n_boards * sizeof(uint64_t);
    HIP_TRY(hipMemcpyAsync(e->dev.root_hb, src->hbits, nb, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dev.root_vb, src->vbits, nb, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dev.root_meta, src->meta, nb, hipMemcpyDeviceToDevice, s));
    if (reset_trees) {
        HIP_TRY(qzl::reset(e->dev, 0, s));
        HIP_TRY(reset_round_state(e, s));
    }
    return 0;
}

int qz_engine_get_boards(qz_engine* e, const qz_boards* dst, void* stream) {
    ENGINE_CHECK(e);
    int r;
    if ((r = check_boards(dst, e->cfg.n_boards))) return r;
    hipStream_t s = (hipStream_t)stream;
    size_t nb = (size_t)e->cfg.n_boards * sizeof(uint64_t);
    HIP_TRY(hipMemcpyAsync(dst->hbits, e->dev.root_hb, nb, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(dst->vbits, e->dev.root_vb, nb, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(dst->meta, e->dev.root_meta, nb, hipMemcpyDeviceToDevice, s));
    return 0;
}

int qz_engine_get_plies(qz_engine* e, int32_t* plies, void* stream) {
    ENGINE_CHECK(e);
    if (!plies) return fail(QZ_E_INVALID, "plies is null");
    HIP_TRY(hipMemcpyAsync(plies, e->dev.ply, (size_t)e->cfg.n_boards * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int qz_engine_dropped_games(qz_engine* e, qz_dropped_game* out, int cap, int64_t* total, void* stream) {
    ENGINE_CHECK(e);
    if (cap < 0 || (cap > 0 && !out)) return fail(QZ_E_INVALID, "bad out / cap");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long n = 0;
    HIP_TRY(hipMemcpyAsync(&n, e->dev.counters + QZ_C_DROPS_LOGGED, sizeof(n), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (total) *total = (int64_t)n;
    const unsigned long long have = n < QZ_DROP_LOG ? n : QZ_DROP_LOG;
    const int k = (int)(have < (unsigned long long)cap ? have : (unsigned long long)cap);
    if (k == 0) return 0;
    std::vector<unsigned long long> h((size_t)QZ_DROP_LOG * 4);
    HIP_TRY(hipMemcpyAsync(h.data(), e->dev.drop_log, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < k; i++) {  // the newest k, oldest of them first
        const unsigned long long* r = h.data() + ((n - (unsigned long long)k + (unsigned long long)i) % QZ_DROP_LOG) * 4ull;
        out[i].hbits = r[0];
        out[i].vbits = r[1];
        out[i].meta = r[2];
        const int c = (int)(r[3] & 0xFFull);
        out[i].cause = c == QZ_C_ABORT_NO_MOVE ? QZ_DROP_NO_MOVE : c == QZ_C_ABORT_DEPTH ? QZ_DROP_DEPTH : c == QZ_C_ABORT_MAX_PLIES ? QZ_DROP_MAX_PLIES : QZ_DROP_POOL;
        out[i].ply = (int32_t)((r[3] >> 8) & 0xFFFFFFFFull);
        out[i].board = (int32_t)(r[3] >> 40);
        out[i].reserved = 0;
    }
    return k;
}

int qz_engine_set_temp(qz_engine* e, float temp) {
    if (!e) return fail(QZ_E_INVALID, "null engine");
    if (!(temp > 0.f)) return fail(QZ_E_INVALID, "temp must be > 0");
    e->cfg.temp = temp;
    e->dev.temp = temp;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipMemcpy(e->dev_mem, &e->dev, sizeof(e->dev), hipMemcpyHostToDevice));  // (synchronous: the next launch sees it)
    return 0;
}

int qz_engine_set_playouts(qz_engine* e, int n_playout) {
    if (!e) return fail(QZ_E_INVALID, "null engine");
    if (n_playout < 1) return fail(QZ_E_INVALID, "n_playout must be >= 1");
    e->cfg.n_playout = n_playout;
    e->dev.n_playout = n_playout;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipMemcpy(e->dev_mem, &e->dev, sizeof(e->dev), hipMemcpyHostToDevice));
    return 0;
}

int qz_engine_set_rules_opts(qz_engine* e, const qz_rules_opts* opts) {
    if (!e) return fail(QZ_E_INVALID, "null engine");
    if (opts) e->cfg.rules = *opts;
    else memset(&e->cfg.rules, 0, sizeof(e->cfg.rules));
    e->rules = rules_opts(&e->cfg.rules);
    return 0;
}

int qz_mcts_descend(qz_engine* e, void* stream) {
    LOCKSTEP_CHECK(e);
    HIP_TRY(qzl::select(e->dev, (hipStream_t)stream));
    return 0;
}

int qz_mcts_leaf_inputs(qz_engine* e, float* leaf_planes, uint32_t* leaf_mask5, uint8_t* leaf_terminal, void* stream) {
    ENGINE_CHECK(e);
    hipStream_t s = (hipStream_t)stream;
    const EngineDev& d = e->dev;
    HIP_TRY(qzl::movegen_encode(d.leaf_hb, d.leaf_vb, d.leaf_meta, d.n_boards, d.leaf_mask, leaf_planes, d.leaf_term, e->scratch, e->rules, s));
    if (leaf_mask5)
        HIP_TRY(hipMemcpyAsync(leaf_mask5, d.leaf_mask, (size_t)d.n_boards * 5 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (leaf_terminal) HIP_TRY(hipMemcpyAsync(leaf_terminal, d.leaf_term, (size_t)d.n_boards, hipMemcpyDeviceToDevice, s));
    return 0;
}

int qz_mcts_select(qz_engine* e, float* leaf_planes, uint32_t* leaf_mask5, uint8_t* leaf_terminal, void* stream) {
    int r = qz_mcts_descend(e, stream);
    if (r) return r;
    return qz_mcts_leaf_inputs(e, leaf_planes, leaf_mask5, leaf_terminal, stream);
}

int qz_mcts_select_boards(qz_engine* e, const qz_boards* leaf_out, uint32_t* leaf_mask5, uint8_t* leaf_terminal, void* stream) {
    LOCKSTEP_CHECK(e);
    int r;
    if ((r = check_boards(leaf_out, e->cfg.n_boards))) return r;
    hipStream_t s = (hipStream_t)stream;
    const EngineDev& d = e->dev;
    HIP_TRY(qzl::select(d, s));
    HIP_TRY(qzl::movegen_encode(d.leaf_hb, d.leaf_vb, d.leaf_meta, d.n_boards, d.leaf_mask, nullptr, d.leaf_term, e->scratch, e->rules, s));
    size_t nb = (size_t)d.n_boards * sizeof(uint64_t);
    HIP_TRY(hipMemcpyAsync(leaf_out->hbits, d.leaf_hb, nb, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(leaf_out->vbits, d.leaf_vb, nb, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(leaf_out->meta, d.leaf_meta, nb, hipMemcpyDeviceToDevice, s));
    if (leaf_mask5)
        HIP_TRY(hipMemcpyAsync(leaf_mask5, d.leaf_mask, (size_t)d.n_boards * 5 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (leaf_terminal) HIP_TRY(hipMemcpyAsync(leaf_terminal, d.leaf_term, (size_t)d.n_boards, hipMemcpyDeviceToDevice, s));
    return 0;
}

int qz_mcts_expand_backup(qz_engine* e, const float* p, const float* v, void* stream) {
    LOCKSTEP_CHECK(e);
    if (!p || !v) return fail(QZ_E_INVALID, "p/v is null");
    HIP_TRY(qzl::expand_backup(e->dev, p, v, (hipStream_t)stream));
    return 0;
}

int qz_mcts_expand_backup_descend(qz_engine* e, const float* p, const float* v, void* stream) {
    LOCKSTEP_CHECK(e);
    if (!p || !v) return fail(QZ_E_INVALID, "p/v is null");
    HIP_TRY(qzl::expand_backup_select(e->dev, p, v, (hipStream_t)stream));
    return 0;
}

int qz_mcts_root_pi(qz_engine* e, double* pi, int32_t* visits, void* stream) {
    ENGINE_CHECK(e);
    if (!pi && !visits) return fail(QZ_E_INVALID, "pi and visits are both null");
    HIP_TRY(qzl::root_pi(e->dev, pi, visits, (hipStream_t)stream));
    return 0;
}

int qz_mcts_root_children(qz_engine* e, int32_t* visits, double* q, float* prior, int32_t* root_visits, void* stream) {
    ENGINE_CHECK(e);
    HIP_TRY(qzl::root_children(e->dev, visits, q, prior, root_visits, (hipStream_t)stream));
    return 0;
}

int qz_mcts_update_with_move(qz_engine* e, const uint8_t* moves, void* stream) {
    LOCKSTEP_CHECK(e);
    if (!moves) return fail(QZ_E_INVALID, "moves is null");
    HIP_TRY(qzl::update_with_move(e->dev, moves, (hipStream_t)stream));
    return 0;
}

int qz_mcts_finish_move(qz_engine* e, const uint8_t* forced_move, float* pi_out, uint8_t* move_out, void* stream) {
    LOCKSTEP_CHECK(e);
    HIP_TRY(qzl::finish_move(e->dev, forced_move, pi_out, move_out, (hipStream_t)stream));
    return 0;
}

int qz_harvest_counts(qz_engine* e, int64_t counts[2], void* stream) {
    ENGINE_CHECK(e);
    if (!counts) return fail(QZ_E_INVALID, "counts is null");
    unsigned long long h[2] = {0, 0};
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h, e->dev.counters + QZ_C_PENDING_GAMES, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    counts[0] = (int64_t)h[0];
    counts[1] = (int64_t)h[1];
    return 0;
}

int qz_harvest(qz_engine* e, const qz_boards* t_boards, float* t_pi, float* t_z, int32_t* t_game, int32_t* g_board, int64_t cap, void* stream) {
    ENGINE_CHECK(e);
    if (cap < 0) return fail(QZ_E_INVALID, "cap < 0");
    if (cap > 0 && (!t_boards || !t_boards->hbits || !t_boards->vbits || !t_boards->meta || !t_pi || !t_z))
        return fail(QZ_E_INVALID, "null tuple buffers");
    HIP_TRY(qzl::harvest(e->dev, cap ? t_boards->hbits : nullptr, cap ? t_boards->vbits : nullptr,
                         cap ? t_boards->meta : nullptr, t_pi, t_z, t_game, g_board, (long long)cap, (hipStream_t)stream));
    return 0;
}

int qz_engine_stats(qz_engine* e, qz_stats* out, void* stream) {
    ENGINE_CHECK(e);
    if (!out) return fail(QZ_E_INVALID, "out is null");
    unsigned long long h[QZ_C_TOTAL];
    hipStream_t s = (hipStream_t)stream;
    const size_t B = (size_t)e->cfg.n_boards;
    std::vector<uint32_t> bp(B), bt(B), bo(B), nn(B), ne(B), bf(B), bd(B), mh(B), me_(B), orr(B), opl(B), ps(B);
    int pw[QZ_P_COUNT];
    std::vector<unsigned long long> bl(B), bs(B), be(B);
    HIP_TRY(hipMemcpyAsync(h, e->dev.counters, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(bp.data(), e->dev.bc_playouts, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(bt.data(), e->dev.bc_terminal, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(bo.data(), e->dev.bc_overflow, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(bf.data(), e->dev.bc_nonfinite, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(bd.data(), e->dev.bc_maxdepth, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(pw, e->dev.pool_words, sizeof(pw), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(bl.data(), e->dev.bc_levels, B * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(bs.data(), e->dev.bc_scanned, B * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(be.data(), e->dev.bc_expanded, B * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(nn.data(), e->dev.n_nodes, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ne.data(), e->dev.n_edges, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(mh.data(), e->dev.bc_memo_hits, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(me_.data(), e->dev.bc_evals, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(orr.data(), e->dev.bc_open_rounds, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(opl.data(), e->dev.bc_open_plies, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ps.data(), e->dev.pend_slot, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // per-board counters wrap at 2^32 playouts per board (years); sums are 64-bit
    unsigned long long sp = 0, st = 0, so = 0, sl = 0, sf = 0, ss = 0, se = 0;
    uint32_t mn = 0, me = 0, md = 0;
    for (size_t i = 0; i < B; i++) {
        mn = nn[i] > mn ? nn[i] : mn;
        me = ne[i] > me ? ne[i] : me;
        sp += bp[i];
        st += bt[i];
        so += bo[i];
        sf += bf[i];
        md = bd[i] > md ? bd[i] : md;
        ss += bs[i];
        se += be[i];
        sl += bl[i];
    }
    memset(out, 0, sizeof(*out));
    out->games_finished = (int64_t)h[QZ_C_GAMES];
    out->plies_played = (int64_t)h[QZ_C_PLIES];
    out->playouts = (int64_t)sp;
    out->leaf_terminal = (int64_t)st;
    out->node_overflow = (int64_t)so;
    out->aborted_no_move = (int64_t)h[QZ_C_ABORT_NO_MOVE];
    out->aborted_max_plies = (int64_t)h[QZ_C_ABORT_MAX_PLIES];
    out->aborted_pool = (int64_t)h[QZ_C_ABORT_POOL];
    out->aborted_depth = (int64_t)h[QZ_C_ABORT_DEPTH];
    out->games_aborted = out->aborted_no_move + out->aborted_max_plies + out->aborted_pool + out->aborted_depth;
    out->bad_forced_moves = (int64_t)h[QZ_C_BAD_FORCED];
    out->nonfinite_values = (int64_t)sf;
    out->max_depth = (int64_t)md;
    out->deep_descents = (int64_t)h[QZ_C_DEEP_DESCENTS];
    out->deep_descents_cold = (int64_t)h[QZ_C_DEEP_COLD];
    out->deep_levels = (int64_t)h[QZ_C_DEEP_LEVELS];
    out->deep_levels_replayed = (int64_t)h[QZ_C_DEEP_REPLAYED];
    out->edges_scanned = (int64_t)ss;
    out->edges_expanded = (int64_t)se;
    out->tree_pages_total = e->cfg.tree_pool_pages;
    out->tree_pages_in_use = e->cfg.tree_pool_pages - pw[QZ_P_TREE_TOP];
    out->tree_pages_peak = e->cfg.tree_pool_pages - pw[QZ_P_TREE_LOW];
    out->traj_pages_total = e->cfg.traj_pool_pages;
    out->traj_pages_in_use = e->cfg.traj_pool_pages - pw[QZ_P_TRAJ_TOP];
    out->traj_pages_peak = e->cfg.traj_pool_pages - pw[QZ_P_TRAJ_LOW];
    out->pending_games = (int64_t)h[QZ_C_PENDING_GAMES];
    out->pending_plies = (int64_t)h[QZ_C_PENDING_PLIES];
    out->descent_levels = (int64_t)sl;
    out->arena_bytes = e->bytes;
    out->max_nodes = (int64_t)mn;
    out->max_edges = (int64_t)me;
    unsigned long long smh = 0, sme = 0, sor = 0, sop = 0, sw = 0;
    for (size_t i = 0; i < B; i++) {
        smh += mh[i];
        sme += me_[i];
        sor += orr[i];
        sop += opl[i];
        sw += ps[i] != QZ_NONE;
    }
    out->rounds = (int64_t)h[QZ_C_ROUNDS];
    out->memo_hits = (int64_t)smh;
    out->nn_evals = (int64_t)sme;
    unsigned long long ins = h[QZ_C_MEMO_INSERTS], lck = h[QZ_C_MEMO_LOCKED];
    for (int i = 0; i < QZ_C_SPREAD; i++) {  // (k_round_tail counts in QZ_C_SPREAD words each: qz_device.h)
        ins += h[QZ_C_COUNT + i];
        lck += h[QZ_C_COUNT + QZ_C_SPREAD + i];
    }
    out->memo_inserts = (int64_t)ins;
    out->memo_locked = (int64_t)lck;
    out->open_rounds = (int64_t)sor;
    out->open_plies = (int64_t)sop;
    out->waiting_boards = (int64_t)sw;
    out->runaway_descents = (int64_t)h[QZ_C_RUNAWAY];
    out->compact_slices = (int64_t)h[QZ_C_COMPACT_SLICES];
    out->miss_overflow = (int64_t)h[QZ_C_MISS_OVERFLOW];
    return 0;
}

// Leaf-evaluator glue: per-(sample, channel) normalisation + affine (+ residual) (+ ReLU) of an
// NCHW float32 tensor with 9x9 planes, in one pass (see qz_nn.hip).
int qz_nn_instnorm_act(const float* x, const float* gamma, const float* beta, const float* residual, float* out,
                       int64_t n_planes, int channels, int relu, float eps, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n_planes < 0 || channels <= 0) return fail(QZ_E_INVALID, "bad n_planes/channels");
    if (n_planes > 0 && (!x || !gamma || !beta || !out)) return fail(QZ_E_INVALID, "null tensor");
    HIP_TRY(qzl::instnorm_act(x, gamma, beta, residual, out, (long long)n_planes, channels, relu, eps, (hipStream_t)stream));
    return 0;
}

// the same for channels-last memory ([B][81][C]); channels <= 64
int qz_nn_instnorm_act_nhwc(const float* x, const float* gamma, const float* beta, const float* residual, float* out,
                            int64_t n_samples, int channels, int relu, float eps, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n_samples < 0 || channels <= 0 || channels > 64) return fail(QZ_E_INVALID, "bad n_samples/channels (1..64)");
    if (n_samples > 0 && (!x || !gamma || !beta || !out)) return fail(QZ_E_INVALID, "null tensor");
    HIP_TRY(qzl::instnorm_act_nhwc(x, gamma, beta, residual, out, (long long)n_samples, channels, relu, eps, (hipStream_t)stream));
    return 0;
}

int qz_nn_input_layer(const qz_boards* boards, const uint8_t* terminal, int64_t n, const float* hot9, const float* base0,
                      const float* wd, const float* gamma, const float* beta, float* out, float eps, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n < 0 || n > 0x7fffffff) return fail(QZ_E_INVALID, "bad n");
    if ((r = check_boards(boards, (int)n))) return r;
    if (n > 0 && (!hot9 || !base0 || !wd || !beta || !out)) return fail(QZ_E_INVALID, "null tensor");
    if (n == 0) return 0;
    if ((((uintptr_t)hot9 | (uintptr_t)base0 | (uintptr_t)wd | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) != 0)
        return fail(QZ_E_INVALID, "tables / out must be 16-byte aligned");
    HIP_TRY(qzl::input_layer((const uint64_t*)boards->hbits, (const uint64_t*)boards->vbits, (const uint64_t*)boards->meta, terminal,
                             (long long)n, hot9, base0, wd, gamma, beta, out, eps, (hipStream_t)stream));
    return 0;
}
int qz_nn_head(const float* t, int64_t n, const float* w6k, const float* gamma6, const float* beta6, const float* w1t, const float* b1,
               const float* w2, const float* b2, const float* w3t, const float* b3, float* p_out, float* v_out, float eps, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n < 0) return fail(QZ_E_INVALID, "n < 0");
    if (n == 0) return 0;
    if (!t || !w6k || !beta6 || !w1t || !b1 || !w2 || !b2 || !w3t || !b3 || !p_out || !v_out) return fail(QZ_E_INVALID, "null tensor");
    if (((uintptr_t)t & 15) != 0) return fail(QZ_E_INVALID, "t must be 16-byte aligned");
    HIP_TRY(qzl::head(t, (long long)n, w6k, gamma6, beta6, w1t, b1, w2, b2, w3t, b3, p_out, v_out, eps, (hipStream_t)stream));
    return 0;
}
int qz_nn_conv3x3_norm(const float* x, const void* w16, const float* gamma, const float* beta, const float* residual, float* out, int64_t n,
                       const float* inv_scale, int relu, float eps, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n < 0) return fail(QZ_E_INVALID, "n < 0");
    if (n == 0) return 0;
    if (!x || !w16 || !gamma || !beta || !out || !inv_scale) return fail(QZ_E_INVALID, "null tensor");
    if ((((uintptr_t)x | (uintptr_t)w16 | (uintptr_t)out | (uintptr_t)residual) & 15) != 0) return fail(QZ_E_INVALID, "tensors must be 16-byte aligned");
    HIP_TRY(qzl::conv3x3_norm(x, w16, gamma, beta, residual, out, (long long)n, inv_scale, relu, eps, (hipStream_t)stream));
    return 0;
}
int qz_nn_trunk(float* x, float* tmp, int64_t n, int n_blocks, const void* const* w16, const float* const* gamma, const float* const* beta,
                const float* inv_scale, float eps, int fused, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n < 0 || n_blocks < 0) return fail(QZ_E_INVALID, "n < 0 or n_blocks < 0");
    if (n == 0 || n_blocks == 0) return 0;
    if (!x || !w16 || !gamma || !beta || !inv_scale || ((!fused || n_blocks > 8) && !tmp))
        return fail(QZ_E_INVALID, "null argument (tmp is needed unless fused with n_blocks <= 8)");
    if ((((uintptr_t)x | (uintptr_t)tmp) & 15) != 0) return fail(QZ_E_INVALID, "tensors must be 16-byte aligned");
    for (int l = 0; l < 2 * n_blocks; l++)
        if (!w16[l] || !gamma[l] || !beta[l]) return fail(QZ_E_INVALID, "null layer tensor");
    HIP_TRY(qzl::trunk(x, tmp, (long long)n, n_blocks, w16, gamma, beta, inv_scale, eps, fused, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, nullptr));
    return 0;
}
int qz_nn_trunk_heads(const float* x, int64_t n, int n_blocks, const void* const* w16, const float* const* gamma, const float* const* beta,
                      const float* inv_scale, const void* w6_16, const float* gamma6, const float* beta6, const float* w1t,
                      const float* b1, const float* w2, const float* b2, const float* w3t, const float* b3, float* feat, float* p_out,
                      float* v_out, float eps, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n < 0 || n_blocks <= 0 || n_blocks > 8) return fail(QZ_E_INVALID, "n < 0 or n_blocks outside 1..8");
    if (n == 0) return 0;
    if (!x || !w16 || !gamma || !beta || !inv_scale || !w6_16 || !gamma6 || !beta6 || !w1t || !b1 || !w2 || !b2 || !w3t || !b3 || !feat || !p_out || !v_out)
        return fail(QZ_E_INVALID, "null argument");
    if ((((uintptr_t)x | (uintptr_t)w6_16) & 15) != 0 || ((uintptr_t)feat & 7) != 0) return fail(QZ_E_INVALID, "x / w6_16 must be 16-byte, feat 8-byte aligned");
    for (int l = 0; l < 2 * n_blocks; l++)
        if (!w16[l] || !gamma[l] || !beta[l]) return fail(QZ_E_INVALID, "null layer tensor");
    HIP_TRY(qzl::trunk(const_cast<float*>(x), nullptr, (long long)n, n_blocks, w16, gamma, beta, inv_scale, eps, 1, (hipStream_t)stream, w6_16, gamma6, beta6,
                       feat, nullptr));
    HIP_TRY(qzl::head_fc(feat, (long long)n, w1t, b1, w2, b2, w3t, b3, p_out, v_out, (hipStream_t)stream));
    return 0;
}
static int nn_weights_check(const qz_nn_weights* w) {
    if (!w) return fail(QZ_E_INVALID, "null weights");
    if (w->n_blocks <= 0 || w->n_blocks > 8) return fail(QZ_E_INVALID, "n_blocks outside 1..8");
    if (w->precision != 0 && w->precision != 1) return fail(QZ_E_INVALID, "precision must be 0 (split fp16 operands, fp32 accuracy) or 1 (fp16 operands)");
    if (!w->hot9 || !w->base0 || !w->wd || !w->gamma0 || !w->beta0 || !w->w16 || !w->gamma || !w->beta || !w->inv_scale || !w->w6_16 || !w->gamma6 ||
        !w->beta6 || !w->w1t || !w->b1 || !w->w2 || !w->b2 || !w->w3t || !w->b3)
        return fail(QZ_E_INVALID, "null argument");
    if ((((uintptr_t)w->wd | (uintptr_t)w->w6_16) & 15) != 0) return fail(QZ_E_INVALID, "wd / w6_16 must be 16-byte aligned");
    for (int l = 0; l < 2 * w->n_blocks; l++)
        if (!w->w16[l] || !w->gamma[l] || !w->beta[l]) return fail(QZ_E_INVALID, "null layer tensor");
    return 0;
}
// boards -> (p, v) for the first n (or *n_live) boards: the fused trunk launch + the fully connected launch
static int nn_evaluate(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, const uint8_t* terminal, int64_t n, const qz_nn_weights* w, float* feat,
                       float* p_out, float* v_out, const int* n_live, hipStream_t s) {
    const qzl::TrunkInput in = {hb, vb, meta, terminal, w->hot9, w->base0, w->wd, w->gamma0, w->beta0};
    HIP_TRY(qzl::trunk(nullptr, nullptr, (long long)n, w->n_blocks, w->w16, w->gamma, w->beta, w->inv_scale, w->eps, 1, s, w->w6_16, w->gamma6, w->beta6, feat, &in,
                       n_live, w->precision == 1));
    HIP_TRY(qzl::head_fc(feat, (long long)n, w->w1t, w->b1, w->w2, w->b2, w->w3t, w->b3, p_out, v_out, s, n_live));
    return 0;
}
int qz_nn_evaluate(const qz_boards* boards, const uint8_t* terminal, int64_t n, const float* hot9, const float* base0, const float* wd,
                   const float* gamma0, const float* beta0, int n_blocks, const void* const* w16, const float* const* gamma,
                   const float* const* beta, const float* inv_scale, const void* w6_16, const float* gamma6,
                   const float* beta6, const float* w1t, const float* b1, const float* w2, const float* b2, const float* w3t, const float* b3,
                   float* feat, float* p_out, float* v_out, float eps, void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n < 0) return fail(QZ_E_INVALID, "n < 0");
    const qz_nn_weights w = {hot9, base0, wd, gamma0, beta0, n_blocks, w16, gamma, beta, inv_scale, w6_16, gamma6, beta6, w1t, b1, w2, b2, w3t, b3, eps, 0};
    if ((r = nn_weights_check(&w))) return r;
    if (n == 0) return 0;
    if (!boards || !boards->hbits || !boards->vbits || !boards->meta || !feat || !p_out || !v_out) return fail(QZ_E_INVALID, "null argument");
    if (((uintptr_t)feat & 7) != 0) return fail(QZ_E_INVALID, "feat must be 8-byte aligned");
    return nn_evaluate(boards->hbits, boards->vbits, boards->meta, terminal, n, &w, feat, p_out, v_out, nullptr, (hipStream_t)stream);
}
int qz_engine_leaf_boards(qz_engine* e, qz_boards* boards_out, const uint8_t** terminal_out) {
    if (!e || !boards_out) return fail(QZ_E_INVALID, "null argument");
    boards_out->hbits = e->dev.leaf_hb;
    boards_out->vbits = e->dev.leaf_vb;
    boards_out->meta = e->dev.leaf_meta;
    if (terminal_out) *terminal_out = e->dev.leaf_term;
    return 0;
}

int qz_nn_evaluate_w(const qz_boards* boards, const uint8_t* terminal, int64_t n, const qz_nn_weights* w, float* feat, float* p_out, float* v_out,
                     void* stream) {
    int r;
    if ((r = device_check())) return r;
    if (n < 0) return fail(QZ_E_INVALID, "n < 0");
    if ((r = nn_weights_check(w))) return r;
    if (n == 0) return 0;
    if (!boards || !boards->hbits || !boards->vbits || !boards->meta || !feat || !p_out || !v_out) return fail(QZ_E_INVALID, "null argument");
    if (((uintptr_t)feat & 7) != 0) return fail(QZ_E_INVALID, "feat must be 8-byte aligned");
    return nn_evaluate(boards->hbits, boards->vbits, boards->meta, terminal, n, w, feat, p_out, v_out, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------ asynchronous self-play
// k_moves (auto_finish), then the playouts: k_advance for every board -- or, with select_opts bit 5, k_advance for the boards that
// still have walls and k_rows (sixteen lanes per board, qz_rows.h) for the others, side by side on two streams (independent boards;
// the miss list's counter and the page pool are shared through atomics).  Fork / join by events: captures into a HIP graph.
static int launch_advance(qz_engine* e, int max_playouts, unsigned int ticks, int auto_finish, hipStream_t s) {
    const EngineDev& d = e->dev;
    if (!(d.select_opts & 32)) {
        HIP_TRY(qzl::advance(d, max_playouts, ticks, auto_finish, e->par, s));
        return 0;
    }
    if (!e->lanes) HIP_TRY(hipStreamCreateWithFlags(&e->lanes, hipStreamNonBlocking));
    if (!e->ev_lfork) HIP_TRY(hipEventCreateWithFlags(&e->ev_lfork, hipEventDisableTiming));
    if (!e->ev_ljoin) HIP_TRY(hipEventCreateWithFlags(&e->ev_ljoin, hipEventDisableTiming));
    if (auto_finish) HIP_TRY(qzl::moves(d, ticks, s));
    HIP_TRY(hipEventRecord(e->ev_lfork, s));
    HIP_TRY(hipStreamWaitEvent(e->lanes, e->ev_lfork, 0));
    HIP_TRY(qzl::advance_lanes(d, max_playouts, ticks, e->par, e->lanes));
    HIP_TRY(hipEventRecord(e->ev_ljoin, e->lanes));
    HIP_TRY(qzl::advance(d, max_playouts, ticks, 0, e->par, s));
    HIP_TRY(hipStreamWaitEvent(s, e->ev_ljoin, 0));
    return 0;
}
int qz_selfplay_advance(qz_engine* e, int max_playouts, int budget_us, int auto_finish, void* stream) {
    ENGINE_CHECK(e);
    if (max_playouts <= 0) return fail(QZ_E_INVALID, "max_playouts must be > 0");
    // (a board whose move's subtree copy is not done sits out of k_advance -- reroot_pend -- and only k_moves, the launch
    // auto_finish adds, continues the copy: without it such a board would sit out for ever, silently: ADVICE r4)
    if (e->async_moves && !auto_finish)
        return fail(QZ_E_INVALID, "engine is in asynchronous self-play (boards may hold a move whose subtree copy only a launch with auto_finish continues): "
                                  "keep auto_finish set, or reset the engine (qz_engine_reset / qz_engine_set_boards with reset_trees) first");
    const unsigned int ticks = budget_us > 0 ? (unsigned int)budget_us * 100u : 0xFFFFFFFFu;  // s_memrealtime: 100 MHz
    int r;
    if ((r = launch_advance(e, max_playouts, ticks, auto_finish, (hipStream_t)stream))) return r;
    if (auto_finish) e->async_moves = true;
    return 0;
}
int qz_selfplay_leaf_rules(qz_engine* e, void* stream) {
    ENGINE_CHECK(e);
    const EngineDev& d = e->dev;
    HIP_TRY(qzl::movegen_encode(d.miss_hb, d.miss_vb, d.miss_meta, d.n_boards, d.miss_mask, nullptr, nullptr, e->scratch, e->rules, (hipStream_t)stream,
                                d.miss_count + e->par));
    return 0;
}
int qz_selfplay_evaluate(qz_engine* e, const qz_nn_weights* w, void* stream) {
    ENGINE_CHECK(e);
    int r;
    if ((r = nn_weights_check(w))) return r;
    const EngineDev& d = e->dev;
    return nn_evaluate(d.miss_hb, d.miss_vb, d.miss_meta, nullptr, d.n_boards, w, e->feat, d.miss_p, d.miss_v, d.miss_count + e->par, (hipStream_t)stream);
}
int qz_selfplay_round_tail(qz_engine* e, void* stream) {
    ENGINE_CHECK(e);
    HIP_TRY(qzl::round_tail(e->dev, e->par, (hipStream_t)stream));
    e->par ^= 1;
    return 0;
}
int qz_selfplay_round(qz_engine* e, const qz_nn_weights* w, int max_playouts, int budget_us, int auto_finish, void* stream) {
    // The four pieces, with what does not depend on the network beside it: after k_advance the legal sets of the miss list
    // (read by the round's tail and the next launch, not by the network) and the moves of the boards that have done their
    // playouts (k_moves: the boards' own trees, roots and trajectories; it only pops pages, the tail only pushes, and the
    // tail waits for both) run on a second stream while the trunk has the matrix cores.  Per board the order of operations
    // is the one of the separate calls: ... playouts, move, playouts ...
    ENGINE_CHECK(e);
    int r;
    if (max_playouts <= 0) return fail(QZ_E_INVALID, "max_playouts must be > 0");
    if (e->async_moves && !auto_finish)  // (see qz_selfplay_advance: a pending subtree copy is only continued by the moves' launch)
        return fail(QZ_E_INVALID, "engine is in asynchronous self-play: keep auto_finish set, or reset the engine first");
    if ((r = nn_weights_check(w))) return r;
    hipStream_t s = (hipStream_t)stream;
    // (every handle on its own: after a partial failure the next call creates what is missing instead of using a null stream -- ADVICE r5)
    if (!e->side) HIP_TRY(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    if (!e->side2) HIP_TRY(hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking));
    if (!e->ev_fork) HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    if (!e->ev_join) HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    if (!e->ev_join2) HIP_TRY(hipEventCreateWithFlags(&e->ev_join2, hipEventDisableTiming));
    const EngineDev& d = e->dev;
    const unsigned int ticks = budget_us > 0 ? (unsigned int)budget_us * 100u : 0xFFFFFFFFu;  // s_memrealtime: 100 MHz
    if ((r = launch_advance(e, max_playouts, ticks, 0, s))) return r;
    HIP_TRY(hipEventRecord(e->ev_fork, s));
    // (the network is queued first: its workgroups -- 4 per CU, most of the LDS -- should be placed before the side kernels'; queuing
    // the side kernels first and / or a high-priority side stream measured the same: profiles/round4/SUMMARY.md)
    if ((r = nn_evaluate(d.miss_hb, d.miss_vb, d.miss_meta, nullptr, d.n_boards, w, e->feat, d.miss_p, d.miss_v, d.miss_count + e->par, s)))
        return r;
    // The rules op and the moves are independent of each other (the miss list / the boards' own trees, roots and trajectories)
    // and each gets a stream of its own: on ONE side stream the moves queued behind the rules op, whose few workgroups wait
    // for room beside the trunk's for most of the trunk's duration -- the moves then started when the network was nearly
    // done, and the round's tail waited 133 us for them (rocprofv3 trace, benchmarks/trace_round_gaps.py; round 5).
    HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
    HIP_TRY(qzl::movegen_encode(d.miss_hb, d.miss_vb, d.miss_meta, d.n_boards, d.miss_mask, nullptr, nullptr, e->scratch, e->rules, e->side, d.miss_count + e->par));
    HIP_TRY(hipEventRecord(e->ev_join, e->side));
    if (auto_finish) {
        HIP_TRY(hipStreamWaitEvent(e->side2, e->ev_fork, 0));
        HIP_TRY(qzl::moves(d, ticks, e->side2));
        e->async_moves = true;
        HIP_TRY(hipEventRecord(e->ev_join2, e->side2));
    }
    HIP_TRY(hipStreamWaitEvent(s, e->ev_join, 0));
    if (auto_finish) HIP_TRY(hipStreamWaitEvent(s, e->ev_join2, 0));
    return qz_selfplay_round_tail(e, stream);
}
int qz_selfplay_parity(qz_engine* e) { return e ? e->par : fail(QZ_E_INVALID, "null engine"); }
int qz_selfplay_misses(qz_engine* e, qz_boards* boards_out, const int32_t** n_dev_out, uint32_t** mask5_out, float** p_out, float** v_out) {
    if (!e) return fail(QZ_E_INVALID, "null engine");
    if (boards_out) {
        boards_out->hbits = e->dev.miss_hb;
        boards_out->vbits = e->dev.miss_vb;
        boards_out->meta = e->dev.miss_meta;
    }
    if (n_dev_out) *n_dev_out = e->dev.miss_count + e->par;
    if (mask5_out) *mask5_out = e->dev.miss_mask;
    if (p_out) *p_out = e->dev.miss_p;
    if (v_out) *v_out = e->dev.miss_v;
    return 0;
}
int qz_memo_flush(qz_engine* e, void* stream) {
    ENGINE_CHECK(e);
    if (!e->dev.memo.small) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (++e->flushes >= 0xFFF0u) {  // the 16-bit epoch is about to wrap: really clear the tables and start over
        const uint32_t zero = 0u;
        HIP_TRY(hipMemsetAsync(e->dev.memo.small, 0, e->memo_small_bytes, s));
        HIP_TRY(hipMemsetAsync(e->dev.memo.big, 0, e->memo_big_bytes, s));
        HIP_TRY(hipMemcpyAsync(e->dev.memo.epoch, &zero, sizeof(zero), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        e->flushes = 1;
    }
    HIP_TRY(qzl::memo_flush(e->dev, s));
    return 0;
}

// self-test hook used by the GPU tests: device sqrt(double(i)) for i in [0, n)
int qz_selftest_sqrt(double* out_dev, int n, void* stream) {
    int r;
    if ((r = device_check())) return r;
    HIP_TRY(qzl::sqrt_table(out_dev, n, (hipStream_t)stream));
    return 0;
}

}  // extern "C"

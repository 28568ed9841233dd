// qz_rows.h -- k_rows: the asynchronous loop (MCTS._playout, mcts.py:103-127) with SIXTEEN LANES PER BOARD -- four boards per
// wavefront -- for the boards on which neither player has a wall left (the regime a reference-faithful game spends 99 % of its
// plies in).  Included by qz_kernels.hip.  qz_config.select_opts bit 5; off by default (measured: below).
//
// k_advance gives a board a whole wavefront and keeps everything that is the same for the board's 64 lanes in scalar registers.
// Its throughput is (boards resident) / (time of a board's playout chain): measured at eight 64-register wavefronts per SIMD (8,192
// boards; seven since, qz_kernels.hip) ~18 us per playout, with the SIMDs' vector pipes 76-81 % and their scalar units ~70 % busy (SQ
// counters: 911-992 vector + ~900 scalar instructions per playout).  A late-game descent is ~15 levels of nodes with two to six children: it uses a
// quarter of the 64 lanes of a replay round.  Here a board has a ROW of 16 lanes (what the DPP row operations address), a wavefront
// carries four boards, and what was wave-uniform is row-uniform: a value every lane of the row holds in a vector register or -- the
// launch-level state -- one LDS word per row.  117 registers, no scratch, 3.75 KB of LDS per wavefront: four wavefronts per SIMD by
// the registers = 16,384 boards; beside k_advance's launch for the boards with walls three per SIMD find room (12,288 boards, see
// advance_lanes: register fragmentation); and the four rows of a wavefront share every instruction they execute
// in the same phase -- the loop is written so that they mostly are: all rows start a playout together, replay together (a row whose
// record confirmed less idles for a round), walk a level together, probe, expand and back up together: 595 vector + 483 scalar
// instructions per playout.  A row whose board leaves the launch (a leaf for the network, its n_playout playouts done) stores it and
// takes the next board from the round's queue (k_rows_scout's list + one atomic), like a wavefront slot of k_advance: an engine holds
// more boards than the chip holds rows.
//
// MEASURED (profiles/round6/SUMMARY.md 2): a wavefront's iteration -- four playouts -- takes ~28 us with three wavefronts per SIMD
// and 22 k cycles alone, ~19 k of them the wavefront's own 4,300 instructions issued one after the other: a wavefront issues at most
// one instruction per turn of its SIMD, so with three wavefronts the SIMD issues 0.94 instructions per turn where k_advance's eight
// offer enough of a mix for 1.44 (vector pipe 40 %, scalar unit 32 % busy).  12,288 busy rows make 444 M playouts/s inside the launch
// -- k_advance's 448 M, at 20,480 boards instead of 13,312 -- and 322 M per second of a whole round (k_advance: 343-347 M at 13,312
// boards).  Taking 17 % of its vector instructions out (the descriptor's fields by scalar loads of their own) moved it 0.6 %; a grid of
// 3,840 wavefronts (k_advance's stream held back so that they find room) 1.5 %.  Parity with the default at 1.5 x the boards is not a
// reason to switch: the default stays k_advance.
//
// The algorithm is k_advance's, piece for piece (select_core / expand_node / backup_leaf / the memo, qz_kernels.hip): sixteen descent
// records per board (lane r of the row holds record r's length), replay rounds of 16 levels (lane = level) instead of 64, the
// first level that does not come out as recorded selected by its own lane, the hint per edge (Edge::rid) that names the record
// to go on in, the descent's first 40 levels mirrored in LDS, the leaf's board from the path's moves in one row reduction, the
// memo's small table, the miss list.  Records, trees and pending evaluations are k_advance's own formats: a board moves from
// k_advance to k_rows with the ply that places the last wall, as it is.  Per board the operations and their float64 / float32
// arithmetic are the lock-step engine's: tests/test_gpu_lanes.py runs every regime on this kernel against oracle.OracleMCTS, and
// tests/test_gpu_async*.py their games and the reference's fixtures.
#pragma once

#ifdef QZ_ROWS_STAMPS  // diagnostic build only (benchmarks/rows_stamps.py): where a wavefront's time goes, by code section
__device__ unsigned long long g_rows_stamps[24];
#define QZ_RS_DECL unsigned long long rs_t = __builtin_amdgcn_s_memtime(), rs_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define QZ_RS(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); rs_acc[k] += n_ - rs_t; rs_t = n_; }
#define QZ_RS_N(k, v) { rs_acc[k] += (v); }
#else
#define QZ_RS_DECL
#define QZ_RS(k)
#define QZ_RS_N(k, v)
#endif
namespace rows {
constexpr int W = 16;             // lanes per board
constexpr int NR = 64 / W;        // boards per wavefront
constexpr uint32_t LCAP = 40;     // levels of a descent mirrored in LDS per board (576 B; deeper levels live in the descent buffer in memory): a row's LDS is 1 KB, a wavefront's 4 KB -- sixteen wavefronts fit the 64 KB a CU gives a kernel here
constexpr uint32_t RMASK = (1u << W) - 1u;

__device__ __forceinline__ int rbase(const int lane) { return lane & (64 - W); }
// the row's share of a ballot (rows are active as a whole: every condition of the loop is row-uniform)
__device__ __forceinline__ uint32_t rballot(const bool p, const int lane) { return (uint32_t)(__ballot(p) >> rbase(lane)) & RMASK; }
// lane l of this lane's row (l row-uniform, 0..15)
__device__ __forceinline__ uint32_t rread(const uint32_t x, const int l, const int lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((rbase(lane) + (l & (W - 1))) << 2, (int)x);
}
__device__ __forceinline__ double rread_f64(const double v, const int l, const int lane) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    return __longlong_as_double((long long)((uint64_t)rread((uint32_t)u, l, lane) | ((uint64_t)rread((uint32_t)(u >> 32), l, lane) << 32)));
}
// sum over the row (DPP row shifts: lane 15 of the row ends up with the total), returned row-uniform
__device__ __forceinline__ uint32_t rsum(const uint32_t v, const int lane) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    return rread((uint32_t)x, W - 1, lane);
}
// (value, index) of the row's maximum, FIRST index on ties (max() over the children dict, mcts.py:42), in every lane of the row:
// a butterfly over the row's rotations (row_ror: 8, 4, 2, 1)
__device__ __forceinline__ void rargmax(double& v, int& k) {
#define QZ_ROW_STEP(ctrl)                                                                                     \
    {                                                                                                          \
        const uint64_t u_ = (uint64_t)__double_as_longlong(v);                                                 \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u_, ctrl, 0xf, 0xf, false); \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u_ >> 32), ctrl, 0xf, 0xf, false); \
        const int ok_ = __builtin_amdgcn_update_dpp(0, k, ctrl, 0xf, 0xf, false);                              \
        const double ov_ = __longlong_as_double((long long)((uint64_t)lo_ | ((uint64_t)hi_ << 32)));           \
        /* a pair without a candidate (k = INT_MAX) never wins; NaN values: every comparison false, the smaller index stays */ \
        if (ok_ != 0x7fffffff && (k == 0x7fffffff || ov_ > v || (!(v > ov_) && ok_ < k))) {                    \
            v = ov_;                                                                                           \
            k = ok_;                                                                                           \
        }                                                                                                      \
    }
    QZ_ROW_STEP(0x128)  // row_ror:8
    QZ_ROW_STEP(0x124)  // row_ror:4
    QZ_ROW_STEP(0x122)  // row_ror:2
    QZ_ROW_STEP(0x121)  // row_ror:1
#undef QZ_ROW_STEP
}
enum { RC_PLAYOUTS = 0, RC_TERMINAL, RC_OVERFLOW, RC_NONFINITE, RC_MAXDEPTH, RC_HITS, RC_EVALS, RC_LEVELS, RC_SCANNED, RC_EXPANDED, RC_STAMP0 = 16, RC_WORDS = 32 };
}  // namespace rows

// the boards k_rows plays this round (lane = board): playing, nobody has a wall left, no subtree copy pending
// a board k_rows plays: both players out of walls (meta bits 16..31 = walls of player 1 / 2)
__device__ __forceinline__ bool lanes_eligible(const uint64_t meta) { return ((meta >> 16) & 0xFFFFull) == 0ull; }
__global__ __launch_bounds__(64) void k_rows_scout(EngineDev E) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool need = b < E.n_boards && E.status[b] == QZ_PLAYING && lanes_eligible(E.root_meta[b]) && E.reroot_pend[b] == 0u && !(E.release[b] & 2u);
    const uint64_t m = __ballot(need);
    if (m == 0ull) return;
    const int lane = lane_id();
    const int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t pos = 0u;
    if (lane == leader) pos = atomicAdd(E.rows_list, (uint32_t)__popcll(m));  // ([0] count, [1] the queue's cursor, [2..] the boards)
    pos = rdl(pos, leader);
    const uint32_t at = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (need && at < (uint32_t)E.n_boards) E.rows_list[2u + at] = (uint32_t)b;
}

__device__ __forceinline__ void rows_boards(EngineDev& E, const int max_iters, const unsigned int budget, const int par) {
    using namespace rows;
    constexpr uint32_t R = QZ_PATH_RECS, CAP = QZ_PATH_CAP;
    static_assert(QZ_PATH_RECS == W, "lane r of a board's row holds record r");
    struct RowShared {
        unsigned long long wb[LCAP];  // the current descent: (move << 56 | block << 8 | child count) per level ...
        uint32_t we[LCAP];            // ... and the chosen edge
        uint32_t pt[64];              // the board's page table (entries 0..63; beyond: the table in memory)
        uint32_t lc[RC_WORDS];        // counter deltas of the launch + the records' last-use stamps
        uint32_t st[24];              // the board's launch state (row-uniform words every lane of the row reads / writes alike): as loop-carried registers they spilled
    };
    __shared__ RowShared s_row[NR];   // (one structure per board: one base address per lane, the fields at constant offsets)
    const int lane = lane_id();
    const int row = lane / W, rl = lane & (W - 1);
    // The round's boards, compacted by k_rows_scout (E.rows_list: [0] count, [1] the queue's cursor, [2..] the boards).  A row starts
    // with list entry (wavefront, row); when its board leaves the launch -- a leaf for the network, its n_playout playouts done -- it
    // stores the board and takes the next entry nobody has taken (one atomic), like a wavefront slot of k_advance: an engine holds
    // more boards than the chip holds rows, and no row idles while a board waits.
    {
        const uint32_t n_list0 = min(E.rows_list[0], (uint32_t)E.n_boards);
        if (__ballot(blockIdx.x * (uint32_t)NR + (uint32_t)row < n_list0) == 0ull) return;
    }
    // the descriptor's fields by scalar loads of their own (QZ_KARG_*, qz_kernels.hip): the hot ones here, the others where a board is
    // loaded / stored / handed to the miss list, so that no 16-dword piece of the argument segment lives (spilled) across the loop
    const unsigned long long kp_ = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    QZ_KARG_P(Edge, edge_pool); QZ_KARG_P(uint32_t, path_edges); QZ_KARG_P(unsigned long long, path_blocks); QZ_KARG_P(uint32_t, memo.small);
    QZ_KARG_P(uint32_t, free_tree); QZ_KARG_P(int, pool_words); QZ_KARG_P(unsigned long long, counters); QZ_KARG_P(uint32_t, rows_list);
    QZ_KARG_P(uint64_t, root_hb); QZ_KARG_P(uint64_t, root_vb); QZ_KARG_P(uint32_t, tree_ptab);
    QZ_KARG_S(memo.small_mask); QZ_KARG_S(c_puct); QZ_KARG_S(n_playout); QZ_KARG_S(max_depth); QZ_KARG_S(node_cap); QZ_KARG_S(edge_cap);
    QZ_KARG_S(tree_pool_pages); QZ_KARG_S(n_boards);
    typedef __attribute__((address_space(3))) RowShared lds_row;
    lds_row* const sh = (lds_row*)&s_row[row];
#define pt (sh->pt)
#define we (sh->we)
#define wb (sh->wb)
#define lc (sh->lc)
#define st_rootN (sh->st[0])
#define st_root_ne (sh->st[1])
#define st_root_eoff (sh->st[2])
#define st_nn (sh->st[3])
#define st_neu (sh->st[4])
#define st_np (sh->st[5])
#define st_rec_last (sh->st[6])
#define st_rec_clock (sh->st[7])
#define st_done (sh->st[8])
#define st_iters (sh->st[9])
#define st_half (sh->st[10])
#define st_t_it (sh->st[11])
#define st_mvalid (sh->st[12])
#define st_slot0 (sh->st[13])
#define st_epoch (sh->st[14])
#define st_pool_edges (sh->st[15])
#define st_n_list (sh->st[16])
#define st_first_free (sh->st[17])
#define st_t0 (sh->st[18])
#define st_rpos (sh->st[19])
#define st_li (sh->st[20])
    Edge* const pool = E.edge_pool;
    // (the board's arrays in memory: addresses computed where they are used from `bb`, a copy of the board index the optimiser cannot
    // see through -- held as pointers they were ten registers of a kernel that has 128)
    int bb = 0;
#define ptab_g (E.tree_ptab + tree_slot(E, bb, st_half) * QZ_TREE_PT)
#define pe0 (E.path_edges + (size_t)bb * (R + 1u) * CAP)
#define pb0 (E.path_blocks + (size_t)bb * (R + 1u) * CAP)
#define gwe (pe0 + (size_t)R * CAP)   /* this descent beyond the mirror / for the backup of a later launch */
#define gwb (pb0 + (size_t)R * CAP)
    // (launch constants every row uses now and then: in LDS, not in registers -- the four-wavefront build has 128 and needs them all)
    st_pool_edges = (uint32_t)E.tree_pool_pages * QZ_PAGE_EDGES;
    st_epoch = *E.memo.epoch;
    st_n_list = min(E.rows_list[0], (uint32_t)E.n_boards);
    st_first_free = gridDim.x * (uint32_t)NR;
    st_li = blockIdx.x * (uint32_t)NR + (uint32_t)row;   // the list entry the row takes next
    uint32_t rlen = 0u;            // lane r: record r's length

    auto phys = [&](const uint32_t e) -> uint32_t {  // physical index of logical edge e (per lane)
        const uint32_t pg = e >> QZ_PAGE_SHIFT;
        uint32_t page;
        if (pg < 64u) page = pt[pg];
        else page = __hip_atomic_load(ptab_g + (pg < (uint32_t)QZ_TREE_PT ? pg : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (page << QZ_PAGE_SHIFT) | (e & (QZ_PAGE_EDGES - 1u));
    };
    auto count = [&](const int i, const uint32_t v) {
        if (rl == 0) lc[i] += v;
    };
    // TreeNode.expand (expand_node): pawn codes only -- the mover has no wall left --, lane a of the row owns action a
    auto expand = [&](const uint32_t pe, const uint32_t bits12, const float prior) -> unsigned long long {
        const uint32_t k = (uint32_t)__popc(bits12);
        if (k == 0u) return 0ull;
        uint32_t off = QZ_NONE;
        if (E.node_cap <= 0 || st_nn < (uint32_t)E.node_cap) {  // tree_alloc
            uint32_t o = st_neu;
            const uint32_t np_now = st_np;
            if ((o & (QZ_PAGE_EDGES - 1u)) + k > QZ_PAGE_EDGES) o = (o + QZ_PAGE_EDGES - 1u) & ~(QZ_PAGE_EDGES - 1u);
            const uint32_t pg = o >> QZ_PAGE_SHIFT;
            bool ok = pg < (uint32_t)QZ_TREE_PT && o + k <= (uint32_t)E.edge_cap;
            if (ok && pg >= np_now) {
                uint32_t page = QZ_NONE;
                if (rl == 0) {
                    const int old = atomicSub(E.pool_words + QZ_P_TREE_TOP, 1);
                    if (old <= 0) {
                        atomicAdd(E.pool_words + QZ_P_TREE_TOP, 1);  // empty: undo
                    } else {
                        atomicMin(E.pool_words + QZ_P_TREE_LOW, old - 1);
                        page = E.free_tree[old - 1];
                        ptab_g[pg] = page;
                        if (pg < 64u) pt[pg] = page;
                        else __threadfence();
                    }
                }
                page = rread(page, 0, lane);
                if (page == QZ_NONE) ok = false;
                else st_np = pg + 1u;
                wave_sync();
            }
            if (ok) {
                off = o;
                st_neu = o + k;
            }
        }
        if (off == QZ_NONE) {
            count(RC_OVERFLOW, 1u);
            return 0ull;
        }
        const uint32_t nb = phys(off);
        if (rl < 12 && ((bits12 >> rl) & 1u)) {
            uint4* const q = reinterpret_cast<uint4*>(&pool[nb + (uint32_t)__popc(bits12 & ((1u << rl) - 1u))]);  // order_index: pawn codes ascending
            q[0] = make_uint4(0u, 0u, 0u, __float_as_uint(prior));          // Q = 0.0 | N = 0 | P
            q[1] = make_uint4(0u, (uint32_t)rl, pe, 0xFFFFFFFFu);           // coff = 0 | act, cne = 0, rid = 0 | pedge | spare
        }
        if (pe != QZ_NONE) {
            if (rl == 0) {
                pool[pe].coff = off;
                pool[pe].cne = (uint8_t)k;
            }
        } else {
            st_root_eoff = off;
            st_root_ne = k;
        }
        st_nn += 1u;
        count(RC_EXPANDED, k);
        return ((unsigned long long)nb << 8) | (unsigned long long)k;
    };
    // note_expansion: the new node's block one past the record's end, where the next descent's replay round finds it
    auto note = [&](const uint32_t plen, const unsigned long long blk, const bool mirror) {
        if (blk == 0ull || plen >= CAP) return;
        if (rl == 0) {
            pb0[(size_t)(st_rec_last & (R - 1u)) * CAP + plen] = blk;
            if (mirror && plen < LCAP) wb[plen] = blk;
        }
    };
    // node.update_recursive(-leaf_value) (backup_leaf): lane = level
    auto backup = [&](const double leaf_value, const uint32_t pedge, const uint32_t plen, const uint32_t term, const bool mirror) {
        if (plen <= CAP) {
            for (uint32_t i = (uint32_t)rl; i < plen; i += (uint32_t)W) {
                const uint32_t pe = (mirror && i < LCAP) ? we[i] : gwe[i];
                const double val = ((plen - 1u - i) & 1u) ? leaf_value : -leaf_value;
                const uint32_t N = pool[pe].N + 1u;  // mcts.py:51
                double Q = pool[pe].Q;
                Q += 1.0 * (val - Q) / (double)N;    // mcts.py:53
                pool[pe].N = N;
                pool[pe].Q = Q;
            }
        } else if (rl == 0) {
            double val = -leaf_value;
            uint32_t pe = pedge;
            while (pe != QZ_NONE) {
                const uint32_t N = pool[pe].N + 1u;
                double Q = pool[pe].Q;
                Q += 1.0 * (val - Q) / (double)N;
                pool[pe].N = N;
                pool[pe].Q = Q;
                val = -val;
                pe = pool[pe].pedge;
            }
        }
        st_rootN += 1u;  // the root is updated too
        count(RC_PLAYOUTS, 1u);
        count(RC_LEVELS, plen);
        if (term != 0u) count(RC_TERMINAL, 1u);
    };

    // ---- a board's state into the row (regs_load) + the evaluation it was waiting for; and back to memory (regs_store)
    auto load_board = [&]() {
        asm volatile("" : "+v"(bb));
        QZ_KARG_P(uint64_t, root_meta); QZ_KARG_P(uint8_t, tree_half); QZ_KARG_P(uint32_t, root_N); QZ_KARG_P(uint32_t, root_ne); QZ_KARG_P(uint32_t, root_eoff);
        QZ_KARG_P(uint32_t, n_nodes); QZ_KARG_P(uint32_t, n_edges); QZ_KARG_P(uint32_t, tree_npages); QZ_KARG_P(uint32_t, rec_last); QZ_KARG_P(uint32_t, rec_clock);
        QZ_KARG_P(uint32_t, pl_done); QZ_KARG_P(uint32_t, pend_slot); QZ_KARG_P(uint32_t, rec_len); QZ_KARG_P(uint32_t, rec_stamp);
        const uint64_t rmeta = E.root_meta[bb];
        st_rpos = (uint32_t)(rmeta & 0xFFFFull) | ((uint32_t)((rmeta >> 32) & 0xFFull) << 16);
        const uint32_t half = E.tree_half[bb];
        st_half = half;
        st_rootN = E.root_N[bb];
        st_root_ne = E.root_ne[bb];
        st_root_eoff = E.root_eoff[bb];
        st_nn = E.n_nodes[bb];
        st_neu = E.n_edges[bb];
        const uint32_t npg = E.tree_npages[tree_slot(E, bb, half)];
        st_np = npg;
        st_rec_last = E.rec_last[bb];
        st_rec_clock = E.rec_clock[bb];
        st_done = E.pl_done[bb];
        const uint32_t slot0 = E.pend_slot[bb];
        st_slot0 = slot0;
        st_iters = 0u;
        st_mvalid = 0u;  // levels of the previous descent of THIS launch the mirror still holds
        rlen = E.rec_len[(size_t)bb * R + rl];
        const uint32_t* const ptg = E.tree_ptab + tree_slot(E, bb, half) * QZ_TREE_PT;
        for (int i = rl; i < 64; i += W) pt[i] = (uint32_t)i < npg ? ptg[i] : 0u;
        lc[rl] = 0u;
        lc[RC_STAMP0 + rl] = E.rec_stamp[(size_t)bb * R + rl];
        wave_sync();
    };
    auto store_board = [&](const bool waiting) {
        wave_sync();
        QZ_KARG_P(uint32_t, root_N); QZ_KARG_P(uint32_t, root_ne); QZ_KARG_P(uint32_t, root_eoff); QZ_KARG_P(uint32_t, n_nodes); QZ_KARG_P(uint32_t, n_edges);
        QZ_KARG_P(uint32_t, tree_npages); QZ_KARG_P(uint32_t, rec_last); QZ_KARG_P(uint32_t, rec_clock); QZ_KARG_P(uint32_t, pl_done); QZ_KARG_P(uint32_t, pend_slot);
        QZ_KARG_P(uint32_t, rec_len); QZ_KARG_P(uint32_t, rec_stamp); QZ_KARG_P(uint32_t, bc_playouts); QZ_KARG_P(uint32_t, bc_terminal); QZ_KARG_P(uint32_t, bc_overflow);
        QZ_KARG_P(uint32_t, bc_nonfinite); QZ_KARG_P(uint32_t, bc_maxdepth); QZ_KARG_P(uint32_t, bc_memo_hits); QZ_KARG_P(uint32_t, bc_evals);
        QZ_KARG_P(unsigned long long, bc_levels); QZ_KARG_P(unsigned long long, bc_scanned); QZ_KARG_P(unsigned long long, bc_expanded);
        E.rec_len[(size_t)bb * R + rl] = rlen;
        E.rec_stamp[(size_t)bb * R + rl] = lc[RC_STAMP0 + rl];
        if (rl == 0) {
            E.root_N[bb] = st_rootN;
            E.root_ne[bb] = st_root_ne;
            E.root_eoff[bb] = st_root_eoff;
            E.n_nodes[bb] = st_nn;
            E.n_edges[bb] = st_neu;
            E.tree_npages[tree_slot(E, bb, st_half)] = st_np;
            E.rec_last[bb] = st_rec_last;
            E.rec_clock[bb] = st_rec_clock;
            E.pl_done[bb] = st_done;
            if (!waiting && st_slot0 != QZ_NONE) E.pend_slot[bb] = QZ_NONE;
            E.bc_playouts[bb] += lc[RC_PLAYOUTS];
            E.bc_terminal[bb] += lc[RC_TERMINAL];
            E.bc_overflow[bb] += lc[RC_OVERFLOW];
            E.bc_nonfinite[bb] += lc[RC_NONFINITE];
            if (lc[RC_MAXDEPTH] > E.bc_maxdepth[bb]) E.bc_maxdepth[bb] = lc[RC_MAXDEPTH];
            E.bc_memo_hits[bb] += lc[RC_HITS];
            E.bc_evals[bb] += lc[RC_EVALS];
            E.bc_levels[bb] += (unsigned long long)lc[RC_LEVELS];
            E.bc_scanned[bb] += (unsigned long long)lc[RC_SCANNED];
            E.bc_expanded[bb] += (unsigned long long)lc[RC_EXPANDED];
        }
        wave_sync();
    };

    st_t0 = (unsigned int)__builtin_amdgcn_s_memrealtime();  // (100 MHz: 32 bits wrap after 43 s; the launch's deadline counts from here)
    bool have = false, first_board = true;
    QZ_RS_DECL
    for (;;) {
        QZ_RS(0)  // 0: leaf handling of the iteration before
        if (!have) {
            // ---- the row's next board: its first one by position, the following ones from the queue
            if (!first_board) {
                uint32_t t = 0u;
                if (rl == 0) t = atomicAdd(E.rows_list + 1, 1u);
                st_li = st_first_free + rread(t, 0, lane);
                if ((unsigned int)__builtin_amdgcn_s_memrealtime() - st_t0 > budget) break;  // (a row's FIRST board always gets a playout: the 1-us regimes)
            }
            first_board = false;
            if (st_li >= st_n_list) break;
            bb = (int)E.rows_list[2u + st_li];
            load_board();
            // the evaluation this board was waiting for: TreeNode.expand + update_recursive with the network's answer (path from memory)
            const uint32_t slot0 = st_slot0;
            if (slot0 != QZ_NONE) {
                QZ_KARG_P(uint32_t, miss_mask); QZ_KARG_P(float, miss_p); QZ_KARG_P(float, miss_v); QZ_KARG_P(uint32_t, leaf_pedge); QZ_KARG_P(uint32_t, path_len);
                const uint32_t m0 = E.miss_mask[(size_t)slot0 * 5];
                const float pr = E.miss_p[(size_t)slot0 * QZ_N_ACT + (rl < 12 ? rl : 0)];
                const double value = (double)E.miss_v[slot0];
                const uint32_t pedge0 = E.leaf_pedge[bb], plen0 = E.path_len[bb];
                const unsigned long long blk = expand(pedge0, m0 & 0xFFFu, pr);
                note(plen0, blk, false);
                backup(value, pedge0, plen0, 0u, false);
                st_done++;
                wave_sync();
            }
            st_t_it = (unsigned int)__builtin_amdgcn_s_memrealtime();
            have = true;
            QZ_RS(11)  // 11: board switches (store + load + the pending evaluation)
        }
        asm volatile("" : "+v"(bb));
        // ================================================================ the loop's head: n_playout reached, or the budget
        {
            bool stop = st_done >= (uint32_t)E.n_playout || st_iters >= (uint32_t)max_iters;  // (the move is k_moves' job)
            if (!stop) {
                const unsigned int now = (unsigned int)__builtin_amdgcn_s_memrealtime();
                const unsigned int last = now - st_t_it;
                st_t_it = now;
                if (st_iters > 0u && (now - st_t0) + last > budget) stop = true;
            }
            if (stop) {
                store_board(false);
                have = false;
                continue;
            }
        }
        st_iters++;
        // ================================================================ the descent (select_core)
        int p1 = (int)(int8_t)(st_rpos & 0xFFu), p2 = (int)(int8_t)((st_rpos >> 8) & 0xFFu), cur_pl = (int)((st_rpos >> 16) & 0xFFu);
        uint32_t pedge = QZ_NONE, plen = 0u;
        int ne = (int)st_root_ne;
        bool gdone = false, nonfinite = false;
        uint32_t scanned = 0u;
        int ex_even = 0, ex_odd = 0;  // moves of levels beyond the descent buffer (>= 2,048: they have no entry), summed as they are walked
        uint32_t pr_d0 = 0u, pr_d1 = 0u, pr_d2 = 0u, pr_d3 = 0u, pr_d4 = 0u, pr_d5 = 0u, pr_d6 = 0u, pr_d7 = 0u;  // the memo bucket: dwords rl + 16 k
        if (ne > 0) {
            uint32_t base = phys(st_root_eoff);
            double sq = sqrt_count(st_rootN);  // st_np.sqrt(self._parent._n_visits), float64
            const uint32_t src = st_rec_last & (R - 1u);     // the record of the previous descent
            const uint32_t src_len = rread(rlen, (int)src, lane);
            uint32_t cur = src_len > 0u ? src : QZ_NONE, cur_len = src_len;
            uint32_t left_rec = QZ_NONE, left_at = 0u, used = 0u, walk_credit = 0u;
            bool at_leaf = false;
            QZ_RS(1)  // 1: head + descent set-up
            while (!at_leaf) {
                // ---- replay of record cur from level plen, 16 levels per round (lane = level)
                bool left = false, sel_valid = false, sel_nan = false;
                int sel_kk = 0;
                uint32_t sel_N = 0u, sel_rec = QZ_NONE;
                while (cur != QZ_NONE && walk_credit == 0u && plen + (uint32_t)QZ_REPLAY_MIN <= cur_len && !left) {
                    const uint32_t* const pe = pe0 + (size_t)cur * CAP;
                    const unsigned long long* const pb = pb0 + (size_t)cur * CAP;
                    const uint32_t i = plen + (uint32_t)rl;
                    const bool rec = i < cur_len;
                    bool ok = i <= cur_len && i < CAP;
                    uint32_t lbase = 0u, chosen = QZ_NONE, prev = 0u;
                    int lne = 0;
                    const bool mir = cur == src && cur_len <= st_mvalid;  // the previous descent of this launch: its entries are still in the mirror
                    if (ok) {
                        unsigned long long w;
                        if (mir) {
                            w = i < LCAP ? wb[i] : 0ull;
                            if (rec) chosen = we[i];
                            if (rl > 0) prev = we[i - 1u];
                        } else {
                            w = pb[i];
                            if (rec) chosen = pe[i];
                            if (rl > 0) prev = pe[i - 1u];
                        }
                        lbase = (uint32_t)(w >> 8);
                        lne = (int)(w & 0xFFull);
                    }
                    // nothing in a record is believed before the link test below; an entry that cannot be a block of this pool is not even loaded from
                    ok = ok && lne >= 1 && lne <= 8 && lbase <= st_pool_edges - 8u;
                    if (!ok) {
                        lbase = 0u;
                        lne = 1;
                    }
                    const uint32_t* const pp = reinterpret_cast<const uint32_t*>(&pool[(rl > 0 && prev < st_pool_edges) ? prev : 0u]);
                    const uint32_t* const cq = reinterpret_cast<const uint32_t*>(&pool[(ok && rec && chosen < st_pool_edges) ? chosen : 0u]);
                    const uint32_t last_child = (uint32_t)lne - 1u;
                    const bool any5 = __ballot(lne >= 5) != 0ull;
                    const uint32_t pN = pp[2], pcoff = pp[4], pmisc = pp[5];
                    const uint32_t ccoff = cq[4], cmisc = cq[5];
                    uint4 c0 = *reinterpret_cast<const uint4*>(&pool[lbase]);
                    uint4 c1 = *reinterpret_cast<const uint4*>(&pool[lbase + (1u < last_child ? 1u : last_child)]);
                    uint4 c2 = *reinterpret_cast<const uint4*>(&pool[lbase + (2u < last_child ? 2u : last_child)]);
                    uint4 c3 = *reinterpret_cast<const uint4*>(&pool[lbase + (3u < last_child ? 3u : last_child)]);
                    // an entry counts only if its block IS the child block of the entry above (lane 0: the current node), with that node's child count
                    const uint32_t linked = phys(pcoff);
                    ok = ok && lbase == (rl > 0 ? linked : base) && lne == (rl > 0 ? (int)((pmisc >> 8) & 0xFFu) : ne);
                    const double lsq = rl > 0 ? sqrt_count(pN) : sq;
                    double lbest;
                    uint32_t arg = 0u, lN;
#define QZ_PUCT_OF(c) (__hiloint2double((int)(c).y, (int)(c).x) + (double)(E.c_puct * __uint_as_float((c).w)) * lsq / (double)(1u + (c).z))
#define QZ_PUCT_NEXT(c, j)                                                      \
    {                                                                           \
        const double val_ = QZ_PUCT_OF(c);                                      \
        if (val_ > lbest) { /* first maximum, like max() over the children dict */ \
            lbest = val_;                                                       \
            arg = (j);                                                          \
            lN = (c).z;                                                         \
        }                                                                       \
    }
                    lbest = QZ_PUCT_OF(c0);
                    lN = c0.z;
                    QZ_PUCT_NEXT(c1, 1u)   // (a lane whose node has fewer children sees its last child again: equal, never greater)
                    QZ_PUCT_NEXT(c2, 2u)
                    QZ_PUCT_NEXT(c3, 3u)
                    if (any5) {  // (rare in the late game: a mover without walls has two to five moves)
                        c0 = *reinterpret_cast<const uint4*>(&pool[lbase + (4u < last_child ? 4u : last_child)]);
                        c1 = *reinterpret_cast<const uint4*>(&pool[lbase + (5u < last_child ? 5u : last_child)]);
                        c2 = *reinterpret_cast<const uint4*>(&pool[lbase + (6u < last_child ? 6u : last_child)]);
                        c3 = *reinterpret_cast<const uint4*>(&pool[lbase + (7u < last_child ? 7u : last_child)]);
                        QZ_PUCT_NEXT(c0, 4u)
                        QZ_PUCT_NEXT(c1, 5u)
                        QZ_PUCT_NEXT(c2, 6u)
                        QZ_PUCT_NEXT(c3, 7u)
                    }
#undef QZ_PUCT_NEXT
#undef QZ_PUCT_OF
                    // ok: the lane's node IS the node of its level (given that the levels above came out as recorded), arg is what
                    // TreeNode.select picks there; the level is CONFIRMED if that is the recorded edge
                    const bool match = ok && rec && (lbase + arg == chosen) && (lbest == lbest);
                    const uint32_t lact = cmisc & 0xFFu, lcne = (cmisc >> 8) & 0xFFu;
                    const uint32_t okm = rballot(ok, lane);
                    const uint32_t bad = ~rballot(match, lane) & RMASK;
                    const int nconf = bad ? (__ffs((int)bad) - 1) : W;
                    if (nconf > 0) {
                        if (rl < nconf && i < CAP) {
                            const unsigned long long ent = ((unsigned long long)lact << 56) | ((unsigned long long)lbase << 8) | (unsigned long long)lne;
                            if (i < LCAP) {
                                if (!mir) {
                                    we[i] = chosen;
                                    wb[i] = ent;
                                }
                            } else {
                                gwe[i] = chosen;
                                gwb[i] = ent;
                            }
                        }
                        used |= 1u << cur;
                        plen += (uint32_t)nconf;
                        const int lastl = nconf - 1;
                        pedge = rread(chosen, lastl, lane);
                        const int cne = (int)rread(lcne, lastl, lane);
                        if (cne == 0) {  // the confirmed prefix ends on a leaf (or a finished game)
                            at_leaf = true;
                            break;
                        }
                        sq = sqrt_count(rread(lN, lastl, lane));
                        base = phys(rread(ccoff, lastl, lane));
                        ne = cne;
                    }
                    // the first level that did NOT come out as recorded (or the open level past the record's end): if its lane's node
                    // is right, the lane has just st_done what the walk would do there -- its pick IS the level's selection
                    if (nconf < W && ((okm >> nconf) & 1u)) {
                        sel_valid = true;
                        sel_kk = (int)rread(arg, nconf, lane);
                        sel_N = rread(lN, nconf, lane);
                        sel_nan = rread((uint32_t)!(lbest == lbest), nconf, lane) != 0u;
                        sel_rec = rread(chosen, nconf, lane);  // the recorded edge of that level (QZ_NONE: the open level)
                    }
                    if (nconf < W) left = true;
                    if (nconf < QZ_REPLAY_MIN) walk_credit = (uint32_t)QZ_WALK_CREDIT;
                    QZ_RS_N(8, 1)  // 8: replay rounds executed (by the wavefront)
                }
                QZ_RS(2)  // 2: replay rounds
                if (at_leaf) break;
                if (E.max_depth > 0 && plen > (uint32_t)E.max_depth) break;
                if (walk_credit > 0u) walk_credit--;
                // ---- one level of the walk: lane j of the row takes child j (j + 16, ...)
                uint32_t recorded = QZ_NONE;
                if (sel_valid) recorded = sel_rec;
                else if (cur != QZ_NONE && plen < cur_len) recorded = pe0[(size_t)cur * CAP + plen];
                int kk;
                uint32_t misc, w_coff;
                double w_sq;
                if (sel_valid) {
                    kk = sel_kk;
                    const uint32_t* const q = reinterpret_cast<const uint32_t*>(&pool[base + (uint32_t)kk]);
                    w_coff = q[4];
                    misc = q[5];
                    w_sq = sqrt_count(sel_N);
                    nonfinite = nonfinite || sel_nan;
                } else {
                    double best = 0.0;
                    int bestk = 0x7fffffff;
                    uint32_t mCOff = 0u, mMisc = 0u;
                    double mSq = 0.0;
                    for (int j = rl; j < ne; j += W) {
                        const uint4* const q = reinterpret_cast<const uint4*>(&pool[base + (uint32_t)j]);
                        const uint4 qa = q[0], qc = q[1];
                        const float cp = E.c_puct * __uint_as_float(qa.w);                  // c_puct * self._P in float32
                        const double u = (double)cp * sq / (double)(1u + qa.z);             // mcts.py:69
                        const double val = __hiloint2double((int)qa.y, (int)qa.x) + u;      // mcts.py:70
                        if (bestk == 0x7fffffff || val > best) {  // (a lane's first candidate is always taken: with non-finite values Python's max() keeps the first child)
                            best = val;
                            bestk = j;
                            mSq = sqrt_count(qa.z);
                            mCOff = qc.x;
                            mMisc = qc.y;
                        }
                    }
                    rargmax(best, bestk);
                    kk = bestk;
                    nonfinite = nonfinite || !(best == best);
                    const int wl = kk & (W - 1);  // the winning edge is the winning lane's own best candidate
                    misc = rread(mMisc, wl, lane);
                    w_coff = rread(mCOff, wl, lane);
                    w_sq = rread_f64(mSq, wl, lane);
                }
                const uint32_t e = base + (uint32_t)kk;
                const uint32_t a = misc & 0xFFu;
                if (plen >= CAP && a < 12u) {
                    if (plen & 1u) ex_odd += action_delta((int)a);
                    else ex_even += action_delta((int)a);
                }
                if (rl == 0 && plen < CAP) {
                    const unsigned long long blk = ((unsigned long long)a << 56) | ((unsigned long long)base << 8) | (unsigned long long)(ne > 255 ? 255 : ne);
                    if (plen < LCAP) {
                        we[plen] = e;
                        wb[plen] = blk;
                    } else {
                        gwe[plen] = e;
                        gwb[plen] = blk;
                    }
                }
                if (recorded != e && !(cur != QZ_NONE && plen >= cur_len)) {  // (beyond the end of the record it follows, a descent extends it)
                    // the path leaves the record it was following (or follows none): go on in the record that took this edge last,
                    // if that record still holds the edge at this level
                    if (cur != QZ_NONE) {
                        left_rec = cur;
                        left_at = plen;
                    }
                    const uint32_t rid = misc >> 16;
                    cur = QZ_NONE;
                    if (rid >= 1u && rid <= R && plen < CAP) {
                        const uint32_t rlv = rread(rlen, (int)(rid - 1u), lane);
                        if (rlv > plen && pe0[(size_t)(rid - 1u) * CAP + plen] == e) {
                            cur = rid - 1u;
                            cur_len = rlv;
                        }
                    }
                }
                pedge = e;
                plen++;
                const int cne = (int)((misc >> 8) & 0xFFu);
                if (cne == 0) break;  // TreeNode.is_leaf(): never expanded (or terminal)
                if (E.max_depth > 0 && plen > (uint32_t)E.max_depth) break;
                if (plen > (uint32_t)QZ_TREE_PT * QZ_PAGE_EDGES) {  // deeper than a tree has edges: corrupted storage.  Never hang the GPU
                    if (rl == 0) atomicAdd(&E.counters[QZ_C_RUNAWAY], 1ull);
                    break;
                }
                sq = w_sq;
                base = phys(w_coff);
                ne = cne;
                QZ_RS_N(9, 1)  // 9: walked levels (executions of the section)
                QZ_RS(3)  // 3: walked levels
            }
            QZ_RS(3)
            // ---- the leaf's board = the root's + the moves of all levels (pawn moves: nobody has a wall), the descent's edge records summed
            wave_sync();
            {
                // action_delta(a) + 18 for a = 0..11, six bits each: N S E W NN | SS EE WW NE NW | SE SW
                const uint32_t K0 = 27u | (9u << 6) | (19u << 12) | (17u << 18) | (36u << 24), K1 = 0u | (20u << 6) | (16u << 12) | (28u << 18) | (26u << 24),
                               K2 = 10u | (8u << 6);
                const uint32_t upto = plen < CAP ? plen : CAP;
                uint32_t sc = 0u, tot = 0u;
                for (uint32_t i0 = 0u; i0 < upto; i0 += (uint32_t)W) {
                    const uint32_t i = i0 + (uint32_t)rl;
                    unsigned long long w = 0ull;
                    if (i < upto) w = i < LCAP ? wb[i] : gwb[i];
                    const uint32_t av = (uint32_t)(w >> 56);
                    const uint32_t aa = av < 12u ? av : 0u;
                    const uint32_t tb = aa < 5u ? K0 : (aa < 10u ? K1 : K2), sh = 6u * (aa < 5u ? aa : (aa < 10u ? aa - 5u : aa - 10u));
                    const uint32_t dl = (tb >> sh) & 63u;
                    tot += (i < upto && av < 12u) ? (dl << (16u * (i & 1u))) : 0u;   // even levels: the root's mover; odd: the other (16-bit fields: 1,024 levels of a parity x 36 fit)
                    sc += (uint32_t)(w & 0xFFull);
                }
                tot = rsum(tot, lane);
                scanned = rsum(sc, lane);
                const int n_even = (int)((upto + 1u) >> 1), n_odd = (int)(upto >> 1);
                const int d_even = (int)(tot & 0xFFFFu) - 18 * n_even + ex_even, d_odd = (int)(tot >> 16) - 18 * n_odd + ex_odd;
                if (cur_pl == 1) {
                    p1 += d_even;
                    p2 += d_odd;
                } else {
                    p2 += d_even;
                    p1 += d_odd;
                }
                // only the last move can end the game (a finished position has no children): rotate plen times, or plen - 1
                gdone = (p2 < 9 || p1 > 71) && plen > 0u;
                const uint32_t rot = gdone ? plen - 1u : plen;
                if (rot & 1u) cur_pl = 3 - cur_pl;
            }
            // ---- the memo's bucket, requested the moment the leaf is known: in flight under the record commit
            if (!gdone && E.memo.small) {
                const uint32_t mlo = ((uint32_t)p1 & 0xFFu) | (((uint32_t)p2 & 0xFFu) << 8), mhi = (uint32_t)cur_pl;
                const uint64_t h = memo_hash(E.root_hb[bb], E.root_vb[bb], (uint64_t)mlo | ((uint64_t)mhi << 32));
                const uint32_t* const B = E.memo.small + (size_t)((uint32_t)h & E.memo.small_mask) * (QZ_MEMO_S_WAYS * QZ_MEMO_S_DW);
                pr_d0 = B[rl];
                pr_d1 = B[rl + 16];
                pr_d2 = B[rl + 32];
                pr_d3 = B[rl + 48];
                pr_d4 = B[rl + 64];
                pr_d5 = B[rl + 80];
                pr_d6 = B[rl + 96];
                pr_d7 = B[rl + 112];
            }
            QZ_RS(4)  // 4: the leaf's board + the probe's issue
            // ---- put this descent on record (select_core's commit)
            {
                const uint32_t n = plen < CAP ? plen : CAP;
                const uint32_t clock = st_rec_clock + 1u;
                uint32_t dest, from;
                if (cur != QZ_NONE) {
                    dest = cur;  // the descent is record cur, or extends it
                    from = n > cur_len ? cur_len : n;
                } else {
                    const uint32_t l_left = left_rec != QZ_NONE ? rread(rlen, (int)left_rec, lane) : 0u;
                    if (left_rec != QZ_NONE && n - left_at + QZ_INPLACE_SLACK >= l_left - left_at) {
                        dest = left_rec;
                        from = left_at;
                    } else {
                        // lane r weighs record r; the minimum over the row, the FIRST record that has it wins
                        uint32_t v = 0xFFFFFFFFu;
                        if ((uint32_t)rl != left_rec && !((used >> rl) & 1u)) v = rlen == 0u ? 0u : lc[RC_STAMP0 + rl] + QZ_VICTIM_LEN_WEIGHT * rlen;
                        int m = (int)(v ^ 0x80000000u);
                        {
                            int t;
                            t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x111, 0xf, 0xf, false); m = t < m ? t : m;
                            t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x112, 0xf, 0xf, false); m = t < m ? t : m;
                            t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x114, 0xf, 0xf, false); m = t < m ? t : m;
                            t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x118, 0xf, 0xf, false); m = t < m ? t : m;
                        }
                        const uint32_t bestv = rread((uint32_t)m, W - 1, lane) ^ 0x80000000u;
                        const uint32_t who = rballot(v == bestv, lane);
                        dest = who ? (uint32_t)(__ffs((int)who) - 1) : 0u;
                        if (bestv == 0xFFFFFFFFu) dest = left_rec != QZ_NONE ? left_rec : 0u;  // every record was useful just now
                        from = 0u;
                    }
                }
                const uint16_t stamp = (uint16_t)(dest + 1u);
                const uint32_t first = from > 0u ? from : (left_rec != QZ_NONE ? left_at : 0u);
                if (from < n) {
                    wave_sync();  // lane 0 stored walked levels into the descent buffer, all lanes read it below
                    uint32_t* const qe = pe0 + (size_t)dest * CAP;
                    unsigned long long* const qb = pb0 + (size_t)dest * CAP;
                    for (uint32_t i = from + (uint32_t)rl; i < n; i += (uint32_t)W) {
                        uint32_t ed;
                        unsigned long long bl;
                        if (i < LCAP) {
                            ed = we[i];
                            bl = wb[i];
                        } else {
                            ed = gwe[i];
                            bl = gwb[i];
                        }
                        qe[i] = ed;
                        qb[i] = bl;
                        if (i >= first) pool[ed].rid = stamp;
                    }
                }
                if ((uint32_t)rl == dest) {
                    if (from < n) rlen = n;
                    lc[RC_STAMP0 + rl] = clock;
                } else if ((used >> rl) & 1u) {
                    lc[RC_STAMP0 + rl] = clock;
                }
                st_rec_last = dest;
                st_rec_clock = clock;
            }
            QZ_RS(5)  // 5: record commit
        } else {
            // a root that is not expanded yet IS the leaf: its memo probe all the same
            if (E.memo.small) {
                const uint32_t mlo = ((uint32_t)p1 & 0xFFu) | (((uint32_t)p2 & 0xFFu) << 8), mhi = (uint32_t)cur_pl;
                const uint64_t h = memo_hash(E.root_hb[bb], E.root_vb[bb], (uint64_t)mlo | ((uint64_t)mhi << 32));
                const uint32_t* const B = E.memo.small + (size_t)((uint32_t)h & E.memo.small_mask) * (QZ_MEMO_S_WAYS * QZ_MEMO_S_DW);
                pr_d0 = B[rl];
                pr_d1 = B[rl + 16];
                pr_d2 = B[rl + 32];
                pr_d3 = B[rl + 48];
                pr_d4 = B[rl + 64];
                pr_d5 = B[rl + 80];
                pr_d6 = B[rl + 96];
                pr_d7 = B[rl + 112];
            }
        }
        if (nonfinite) count(RC_NONFINITE, 1u);
        count(RC_SCANNED, scanned);
        if (rl == 0 && plen > lc[RC_MAXDEPTH]) lc[RC_MAXDEPTH] = plen;
        if (plen >= 256u && rl == 0) {  // telemetry of the deepest lines (select_core's counters)
            atomicAdd(&E.counters[QZ_C_DEEP_DESCENTS], 1ull);
            atomicAdd(&E.counters[QZ_C_DEEP_LEVELS], (unsigned long long)plen);
        }
        // ---- drop_if_too_deep: the reference's RecursionError
        if (E.max_depth > 0 && plen > (uint32_t)E.max_depth) {
            QZ_KARG_P(uint8_t, status); QZ_KARG_P(unsigned long long, drop_log); QZ_KARG_P(uint32_t, ply); QZ_KARG_P(uint64_t, root_meta);
            if (rl == 0) {
                E.status[bb] = QZ_ABORTED;
                atomicAdd(&E.counters[QZ_C_ABORT_DEPTH], 1ull);
                log_dropped_game(E, bb, QZ_C_ABORT_DEPTH);
            }
            store_board(false);
            have = false;
            continue;
        }
        st_mvalid = plen < LCAP ? plen : LCAP;
        wave_sync();  // the descent buffer (lane 0 / other lanes) before the backup reads it
        // ================================================================ the leaf
        if (gdone) {  // mcts.py:119-126: a finished game is backed up at once (the reference does not rotate players on a terminal move)
            const int win = p2 < 9 ? 2 : (p1 > 71 ? 1 : 0);
            const uint32_t term = win == cur_pl ? 1u : 2u;
            backup(terminal_value(E, term), pedge, plen, term, true);
            st_done++;
            wave_sync();
            continue;
        }
        QZ_RS(6)  // 6: counters, the terminal leaves' backups
        // the memo: entry e of the bucket lives in dwords 32 e .. 32 e + 31 = registers 2 e, 2 e + 1; its key in lanes 0..5 of register 2 e
        const uint32_t mlo = ((uint32_t)p1 & 0xFFu) | (((uint32_t)p2 & 0xFFu) << 8), mhi = (uint32_t)cur_pl | (st_epoch << 16);
        int way = -1;
        if (E.memo.small) {
            const uint64_t rhb = E.root_hb[bb], rvb = E.root_vb[bb];
            const uint32_t kd = rl == 0 ? (uint32_t)rhb : (rl == 1 ? (uint32_t)(rhb >> 32) : (rl == 2 ? (uint32_t)rvb : (rl == 3 ? (uint32_t)(rvb >> 32) : (rl == 4 ? mlo : mhi))));
            const uint32_t e0 = rballot(rl >= 6 || pr_d0 == kd, lane), e1 = rballot(rl >= 6 || pr_d2 == kd, lane), e2 = rballot(rl >= 6 || pr_d4 == kd, lane),
                           e3 = rballot(rl >= 6 || pr_d6 == kd, lane);
            way = e0 == RMASK ? 0 : (e1 == RMASK ? 1 : (e2 == RMASK ? 2 : (e3 == RMASK ? 3 : -1)));
        }
        QZ_RS(7)  // 7: the probe's wait + key compare
        QZ_RS_N(10, 1)  // 10: iterations that reached the probe
        if (way >= 0) {
            count(RC_HITS, 1u);
            const uint32_t lo = way == 0 ? pr_d0 : (way == 1 ? pr_d2 : (way == 2 ? pr_d4 : pr_d6));   // dwords 0..15 of the entry: key, v, legal pawn bits, p[0..7]
            const uint32_t hi = way == 0 ? pr_d1 : (way == 1 ? pr_d3 : (way == 2 ? pr_d5 : pr_d7));   // dwords 16..31: p[8..11] in lanes 0..3
            const float v = __uint_as_float(rread(lo, 6, lane));
            const uint32_t bits = rread(lo, 7, lane) & 0xFFFu;
            const uint32_t plo = rread(lo, 8 + (rl & 7), lane), phi = rread(hi, rl & 3, lane);
            const float prior = __uint_as_float(rl < 8 ? plo : phi);   // the prior of pawn code rl (rl < 12)
            QZ_RS(13)  // 13: the hit's payload out of the bucket registers
            const unsigned long long blk = expand(pedge, bits, prior);
            QZ_RS(14)  // 14: expansion
            note(plen, blk, true);
            QZ_RS(15)  // 15: note
            backup((double)v, pedge, plen, 0u, true);
            QZ_RS(16)  // 16: backup
            st_done++;
            wave_sync();
            continue;
        }
        // ---- a leaf for the network: into the miss list, the board waits for the next launch and the row takes another board
        {
            bool waiting = true;
            uint32_t sl = 0u;
            QZ_KARG_P(int, miss_count); QZ_KARG_P(uint64_t, miss_hb); QZ_KARG_P(uint64_t, miss_vb); QZ_KARG_P(uint64_t, miss_meta); QZ_KARG_P(uint32_t, pend_slot);
            QZ_KARG_P(uint32_t, leaf_pedge); QZ_KARG_P(uint32_t, path_len);
            if (rl == 0) sl = (uint32_t)atomicAdd(E.miss_count + par, 1);
            sl = rread(sl, 0, lane);
            if (sl >= (uint32_t)E.n_boards) {  // a stale counter (must not happen): the board forgets this descent and repeats it in its next launch
                if (rl == 0) atomicAdd(&E.counters[QZ_C_MISS_OVERFLOW], 1ull);
                waiting = false;
            } else {
                if (rl == 0) {
                    E.miss_hb[sl] = E.root_hb[bb];
                    E.miss_vb[sl] = E.root_vb[bb];
                    E.miss_meta[sl] = (uint64_t)(((uint32_t)p1 & 0xFFu) | (((uint32_t)p2 & 0xFFu) << 8)) | ((uint64_t)(uint32_t)cur_pl << 32);
                    E.pend_slot[bb] = sl;
                    E.leaf_pedge[bb] = pedge;
                    E.path_len[bb] = plen;
                }
                const uint32_t nm = plen < LCAP ? plen : LCAP;
                for (uint32_t i = (uint32_t)rl; i < nm; i += (uint32_t)W) gwe[i] = we[i];  // the path, for the backup of the launch that gets the answer
                count(RC_EVALS, 1u);
            }
            store_board(waiting);
            have = false;
        }
    }
#ifdef QZ_ROWS_STAMPS
    QZ_RS(0)
    if (lane == 0) for (int k = 0; k < 20; k++) if (k != 12) atomicAdd(&g_rows_stamps[k], rs_acc[k]);
    if (lane == 0) atomicAdd(&g_rows_stamps[12], 1ull);
#endif
}
#undef st_rootN
#undef st_root_ne
#undef st_root_eoff
#undef st_nn
#undef st_neu
#undef st_np
#undef st_rec_last
#undef st_rec_clock
#undef st_done
#undef st_iters
#undef st_half
#undef st_t_it
#undef st_mvalid
#undef st_slot0
#undef st_epoch
#undef st_pool_edges
#undef st_n_list
#undef st_first_free
#undef st_t0
#undef st_rpos
#undef st_li
#undef pt
#undef we
#undef wb
#undef lc
#undef ptab_g
#undef pe0
#undef pb0
#undef gwe
#undef gwb
// Three builds of the loop: four wavefronts per SIMD (128 registers: sixteen boards per SIMD), three (160), two (256)
// (EngineDev must stay the FIRST parameter: rows_boards fetches its fields from offset 0 of the kernel-argument segment, QZ_KARG_*)
template <int WEU>
__global__ void k_rows(EngineDev E, int max_iters, unsigned int budget, int par);
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rows<4>(EngineDev E, int max_iters, unsigned int budget, int par) {
    rows_boards(E, max_iters, budget, par);
}
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3), amdgpu_num_vgpr(160))) void k_rows<3>(EngineDev E, int max_iters, unsigned int budget, int par) {
    rows_boards(E, max_iters, budget, par);
}
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rows<2>(EngineDev E, int max_iters, unsigned int budget, int par) {
    rows_boards(E, max_iters, budget, par);
}

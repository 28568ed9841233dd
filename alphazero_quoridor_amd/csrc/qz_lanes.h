// qz_lanes.h -- k_lanes: the asynchronous loop (MCTS._playout, mcts.py:103-127) with ONE LANE PER BOARD, for the boards on
// which NEITHER player has a wall left -- the regime a reference-faithful game spends 99 % of its plies in (20 walls are
// placed in the first few hundred plies of games that last tens of thousands: DESIGN 6).  Included by qz_kernels.hip.
//
// Why a second mapping.  k_advance gives a board a whole wavefront: 64 lanes for the levels of a replay round, of which a
// late-game descent (mean depth 18, nodes of two to six children) uses a third, and the chip holds 8 wavefronts per SIMD =
// 8,192 boards at a time whatever the board count.  On a board without walls every node has at most six children (four
// steps, or three steps + the straight jump and its two diagonals: quoridor.py:272-353), every leaf lives in the memo's
// small table, an expansion writes at most six records and the scratch board is two pawn tiles: the whole playout is plain
// per-lane code.  Here a wavefront carries 64 boards, every board of the engine is resident at once (13,312 boards = 208
// wavefronts; registers and LDS are not the limit any more), and the instruction stream of a level is shared by 64 descents.
// What it gives up is the replay: a descent takes one dependent trip to memory per level.  Two things make up for it:
//   * every phase of a playout costs its lane ONE trip per iteration of the wavefront's loop and all lanes' trips of an
//     iteration are issued together, whatever phase each lane is in (descent level / memo probe / backup of levels the
//     next descent will not pass): one wait per iteration, no lane ever waits for another lane's phase;
//   * the backup of playout i is FOLDED INTO the descent of playout i + 1 (update_recursive, mcts.py:44-62, top-down): the
//     descent re-reads exactly the records the backup has to rewrite -- at every level of the old path the old edge is one of
//     the children the selection loads -- so the lane updates (N, Q) of that child in registers, stores it, and selects with
//     the updated values; where the new path leaves the old one, the levels of the old path below that point (a different
//     subtree: nothing the descent will read) are flushed four per trip.  Per board the arithmetic and its order are the
//     reference's: an edge's (N, Q) is rewritten before anything reads it again, with the same float64 expression.
// Parity: tests/test_gpu_lanes.py and tests/test_gpu_async*.py run their regimes on this kernel as well (select_opts bit 4) -- root
// visits, float64 Q, float32 P bit-equal with the reference fixtures / oracle.OracleMCTS.
// MEASURED (profiles/round6/SUMMARY.md 1): 88 M playouts/s against k_advance's 344 M at 13,312 boards.  Every iteration of the
// wavefront's loop executes the code of EVERY phase some lane is in (~2,000 instructions for one step of each lane), so a lane's
// playout takes ~112 us against 9 us for a wavefront of k_advance: the prototype of the other mapping VERDICT r5 asked to cost out,
// kept as the parity partner of k_rows (same boards, a third formulation), off by default.
#pragma once

constexpr int LN_PT = 32;      // page-table entries per lane held in LDS (trees beyond 65,536 edge records read the table in memory)
constexpr int LN_LCAP = 192;   // levels of a lane's current path held in LDS (deeper ones live in the board's descent buffer in memory)
enum { LN_IDLE = 0, LN_START = 1, LN_DESC = 2, LN_PROBE = 3, LN_FLUSH = 4 };

// a board k_lanes plays: both players out of walls (meta bits 16..31 = walls of player 1 / 2)
__device__ __forceinline__ bool lanes_eligible(const uint64_t meta) { return ((meta >> 16) & 0xFFFFull) == 0ull; }
__device__ __forceinline__ uint32_t ln_fold(const uint32_t w, const uint32_t c) {
    const uint64_t p = (uint64_t)w * (uint64_t)c;
    return (uint32_t)p ^ (uint32_t)(p >> 32);
}
typedef uint32_t ln_u32x3 __attribute__((ext_vector_type(3)));

__global__ __launch_bounds__(64) void k_lanes(EngineDev E, const int max_iters, const unsigned int budget, const int par) {
    __shared__ uint32_t s_pt[LN_PT * 64];
    __shared__ uint32_t s_path[LN_LCAP * 64];
    const int lane = lane_id();
    const int b_ = (int)blockIdx.x * 64 + lane;
    bool on = false;
    uint64_t rmeta = 0ull;
    if (b_ < E.n_boards) {
        rmeta = E.root_meta[b_];
        on = E.status[b_] == QZ_PLAYING && lanes_eligible(rmeta) && E.reroot_pend[b_] == 0u && !(E.release[b_] & 2u);
    }
    if (__ballot(on) == 0ull) return;
    const int b = on ? b_ : 0;
    lds_u32* const pt = (lds_u32*)s_pt + lane;      // entry pg at pt[pg * 64]: one bank per lane
    lds_u32* const lp = (lds_u32*)s_path + lane;    // level i at lp[i * 64]
    Edge* const pool = E.edge_pool;
    const uint32_t half = E.tree_half[b];
    uint32_t* const ptab_g = E.tree_ptab + tree_slot(E, b, half) * QZ_TREE_PT;
    uint32_t* const gpath = E.path_edges + ((size_t)b * (QZ_PATH_RECS + 1) + QZ_PATH_RECS) * QZ_PATH_CAP;  // the board's descent buffer
    const uint64_t rhb = E.root_hb[b], rvb = E.root_vb[b];
    uint32_t rootN = E.root_N[b], root_ne = E.root_ne[b], root_eoff = E.root_eoff[b];
    uint32_t nn = E.n_nodes[b], neu = E.n_edges[b], np = E.tree_npages[tree_slot(E, b, half)];
    uint32_t done = E.pl_done[b];
    const uint32_t slot0 = on ? E.pend_slot[b] : QZ_NONE;
    const uint32_t epoch = *E.memo.epoch;
    const int rp1 = (int)(int8_t)(rmeta & 0xFF), rp2 = (int)(int8_t)((rmeta >> 8) & 0xFF), rcur = (int)((rmeta >> 32) & 0xFF);
    for (uint32_t pg = 0u; pg < (uint32_t)LN_PT; pg++) {
        if (__ballot(on && pg < np) == 0ull) break;
        if (on && pg < np) pt[pg * 64u] = ptab_g[pg];
    }
    // the walls' share of the memo's bucket hash (memo_hash): the same for every leaf of the board
    const uint32_t hx = ln_fold((uint32_t)rhb, 0x9E3779B1u) ^ ln_fold((uint32_t)(rhb >> 32), 0x85EBCA77u) ^ ln_fold((uint32_t)rvb, 0xC2B2AE3Du) ^
                        ln_fold((uint32_t)(rvb >> 32), 0x27D4EB2Fu);
    uint32_t c_playouts = 0u, c_terminal = 0u, c_overflow = 0u, c_nonfinite = 0u, c_maxdepth = 0u, c_hits = 0u, c_evals = 0u, c_levels = 0u,
             c_scanned = 0u, c_expanded = 0u;

    auto phys = [&](const uint32_t e) -> uint32_t {
        const uint32_t pg = e >> QZ_PAGE_SHIFT;
        uint32_t page;
        if (pg < (uint32_t)LN_PT) page = pt[pg * 64u];
        else page = __hip_atomic_load(ptab_g + (pg < (uint32_t)QZ_TREE_PT ? pg : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (page << QZ_PAGE_SHIFT) | (e & (QZ_PAGE_EDGES - 1u));
    };
    auto path_get = [&](const uint32_t i) -> uint32_t { return i < (uint32_t)LN_LCAP ? lp[i * 64u] : gpath[i < (uint32_t)QZ_PATH_CAP ? i : 0u]; };
    auto path_put = [&](const uint32_t i, const uint32_t e) {
        if (i < (uint32_t)LN_LCAP) lp[i * 64u] = e;
        else if (i < (uint32_t)QZ_PATH_CAP) gpath[i] = e;
    };
    // node.update_recursive's step for ONE edge (mcts.py:48-53)
    auto upd = [&](uint32_t& ql, uint32_t& qh, uint32_t& n, const double val) {
        n += 1u;
        double Q = __hiloint2double((int)qh, (int)ql);
        Q += 1.0 * (val - Q) / (double)n;
        ql = (uint32_t)__double2loint(Q);
        qh = (uint32_t)__double2hiint(Q);
    };
    // TreeNode.expand (mcts.py:27-35) under physical edge pe (QZ_NONE: the root): the legal pawn codes in ascending order =
    // actions() order (quoridor.py:146-147), priors pp[a]
    auto expand = [&](const uint32_t pe, const uint32_t bits12, const float (&pp)[12]) {
        const uint32_t k = (uint32_t)__popc(bits12);
        if (k == 0u) return;
        uint32_t off = QZ_NONE;
        if (E.node_cap <= 0 || nn < (uint32_t)E.node_cap) {  // tree_alloc, one lane
            uint32_t o = neu;
            if ((o & (QZ_PAGE_EDGES - 1u)) + k > QZ_PAGE_EDGES) o = (o + QZ_PAGE_EDGES - 1u) & ~(QZ_PAGE_EDGES - 1u);
            const uint32_t pg = o >> QZ_PAGE_SHIFT;
            bool ok = pg < (uint32_t)QZ_TREE_PT && o + k <= (uint32_t)E.edge_cap;
            if (ok && pg >= np) {
                const int old = atomicSub(E.pool_words + QZ_P_TREE_TOP, 1);
                if (old <= 0) {
                    atomicAdd(E.pool_words + QZ_P_TREE_TOP, 1);  // empty: undo
                    ok = false;
                } else {
                    atomicMin(E.pool_words + QZ_P_TREE_LOW, old - 1);
                    const uint32_t page = E.free_tree[old - 1];
                    ptab_g[pg] = page;
                    if (pg < (uint32_t)LN_PT) pt[pg * 64u] = page;
                    else __threadfence();
                    np = pg + 1u;
                }
            }
            if (ok) {
                off = o;
                neu = o + k;
            }
        }
        if (off == QZ_NONE) {
            c_overflow++;
            return;
        }
        const uint32_t nb = phys(off);
        uint32_t r = 0u;
#pragma unroll
        for (int a = 0; a < 12; a++) {
            if ((bits12 >> a) & 1u) {
                uint4* const q = reinterpret_cast<uint4*>(&pool[nb + r]);
                q[0] = make_uint4(0u, 0u, 0u, __float_as_uint(pp[a]));   // Q = 0.0 | N = 0 | P
                q[1] = make_uint4(0u, (uint32_t)a, pe, 0xFFFFFFFFu);     // coff = 0 | act, cne = 0, rid = 0 | pedge | spare
                r++;
            }
        }
        if (pe != QZ_NONE) {
            pool[pe].coff = off;
            pool[pe].cne = (uint8_t)k;
        } else {
            root_eoff = off;
            root_ne = k;
        }
        nn += 1u;
        c_expanded += k;
    };

    uint32_t ph = on ? LN_START : LN_IDLE;
    // The last playout's path still owes its edges of levels fl_i .. old_len - 1 an update with +-old_val (old_len == 0: nothing owed).
    // While `fused`, the running descent IS the old path so far and pays level by level; a descent that leaves the old path (or a
    // launch that ends) flushes the rest, four levels per trip, and then goes on with `after_flush`.
    enum { AF_IDLE = 0, AF_DESC = 1, AF_LEAF = 2 };
    uint32_t old_len = 0u, fl_i = 0u, after_flush = AF_IDLE;
    double old_val = 0.0;
    bool waiting = false, fused = false;
    uint32_t iters = 0u;
    uint32_t base = 0u, plen = 0u, pedge = QZ_NONE, scanned = 0u;
    int ne = 0, p1 = rp1, p2 = rp2, cur = rcur;
    bool nonfinite = false;
    double sq = 0.0;
    // a playout is over: its leaf's value is known, the tree above it is updated when the next descent passes (or at the flush)
    auto complete = [&](const double value, const uint32_t term) {
        rootN += 1u;  // the root is updated too
        done++;
        c_playouts++;
        c_levels += plen;
        c_scanned += scanned;
        if (term != 0u) c_terminal++;
        if (nonfinite) c_nonfinite++;
        c_maxdepth = plen > c_maxdepth ? plen : c_maxdepth;
        if (plen >= 256u) {  // telemetry of the deepest lines (k_select's counters)
            atomicAdd(&E.counters[QZ_C_DEEP_DESCENTS], 1ull);
            atomicAdd(&E.counters[QZ_C_DEEP_COLD], 1ull);
            atomicAdd(&E.counters[QZ_C_DEEP_LEVELS], (unsigned long long)plen);
        }
        if (plen > (uint32_t)QZ_PATH_CAP) {  // deeper than the descent buffer: walk the parent links now (backup_leaf's fallback)
            double val = -value;
            uint32_t pe = pedge;
            while (pe != QZ_NONE) {
                uint32_t N = pool[pe].N + 1u;
                double Q = pool[pe].Q;
                Q += 1.0 * (val - Q) / (double)N;
                pool[pe].N = N;
                pool[pe].Q = Q;
                val = -val;
                pe = pool[pe].pedge;
            }
            old_len = 0u;
        } else {
            old_len = plen;
            old_val = value;
        }
        ph = LN_START;
    };
    // the descent stands on a leaf (TreeNode.is_leaf()) and owes nothing: a finished game is backed up at once (mcts.py:119-126),
    // any other leaf asks the memo
    auto at_leaf = [&]() {
        const int win = p2 < 9 ? 2 : (p1 > 71 ? 1 : 0);  // has_a_winner(): player 2 first (quoridor.py:196-201)
        if (win != 0) {
            const uint32_t term = win == cur ? 1u : 2u;
            complete(terminal_value(E, term), term);
        } else {
            ph = LN_PROBE;
        }
    };

    // ---- the evaluation this board was waiting for (the leaf of an earlier launch): expansion now, its backup is owed like any other
    if (on && slot0 != QZ_NONE) {
        const uint32_t m0 = E.miss_mask[(size_t)slot0 * 5];
        const float* const prow = E.miss_p + (size_t)slot0 * QZ_N_ACT;
        float pp[12];
#pragma unroll
        for (int a = 0; a < 12; a++) pp[a] = prow[a];
        const double value = (double)E.miss_v[slot0];
        pedge = E.leaf_pedge[b];
        plen = E.path_len[b];
        const uint32_t nm = plen < (uint32_t)LN_LCAP ? plen : (uint32_t)LN_LCAP;
        for (uint32_t i = 0u; i < nm; i++) lp[i * 64u] = gpath[i];
        expand(pedge, m0 & 0xFFFu, pp);
        complete(value, 0u);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_it = t0;

    for (;;) {
        if (__ballot(ph != LN_IDLE) == 0ull) break;
        // ================================================================ START: no memory access
        if (ph == LN_START) {
            bool stop = done >= (uint32_t)E.n_playout || iters >= (uint32_t)max_iters;
            if (!stop) {
                // no new playout once the budget is spent -- or would be overrun by a playout as long as this board's last one
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                const unsigned int last = (unsigned int)(now - t_it);
                t_it = now;
                if (iters > 0u && (unsigned int)(now - t0) + last > budget) stop = true;
            }
            if (stop) {
                if (old_len > 0u) {  // what the last playout owes is paid before the launch ends (the move reads the root's children)
                    ph = LN_FLUSH;
                    fl_i = 0u;
                    after_flush = AF_IDLE;
                } else {
                    ph = LN_IDLE;
                }
            } else {
                iters++;
                plen = 0u;
                pedge = QZ_NONE;
                scanned = 0u;
                nonfinite = false;
                p1 = rp1;
                p2 = rp2;
                cur = rcur;
                if (root_ne == 0u) {  // a root that is not expanded yet IS the leaf (old_len is 0: nothing went through it)
                    ph = LN_PROBE;
                } else {
                    base = phys(root_eoff);
                    ne = (int)root_ne;
                    sq = sqrt_count(rootN);  // np.sqrt(self._parent._n_visits), float64
                    fused = old_len > 0u;
                    ph = LN_DESC;
                }
            }
        }
        const bool any_desc = __ballot(ph == LN_DESC) != 0ull, any_flush = __ballot(ph == LN_FLUSH) != 0ull, any_probe = __ballot(ph == LN_PROBE) != 0ull;
        // ================================================================ all of the iteration's loads, issued together
        uint4 ca[6];
        uint2 cm[6];
        uint32_t kold = 0xFFFFFFFFu;
        const bool in_desc = ph == LN_DESC, in_flush = ph == LN_FLUSH, in_probe = ph == LN_PROBE;  // (phases entered further down wait for the next iteration)
        if (any_desc) {
            if (in_desc) {
                const uint32_t lc = (uint32_t)ne - 1u;
#pragma unroll
                for (uint32_t j = 0u; j < 6u; j++) {
                    const uint32_t* const q = reinterpret_cast<const uint32_t*>(&pool[base + (j < lc ? j : lc)]);
                    ca[j] = *reinterpret_cast<const uint4*>(q);        // Q, N, P
                    cm[j] = *reinterpret_cast<const uint2*>(q + 4);    // coff, act | cne << 8 | rid << 16
                }
                if (fused) kold = path_get(plen) - base;
            }
        }
        uint32_t fe[4];
        ln_u32x3 fv[4];
        if (any_flush) {
            if (in_flush) {
#pragma unroll
                for (uint32_t t = 0u; t < 4u; t++) {
                    const uint32_t i = fl_i + t;
                    fe[t] = i < old_len ? path_get(i) : 0u;
                    fv[t] = *reinterpret_cast<const ln_u32x3*>(&pool[fe[t]]);
                }
            }
        }
        uint4 pw[4][5];
        uint32_t pmeta_lo = 0u, pmeta_hi = 0u;
        if (any_probe) {
            if (in_probe) {
                pmeta_lo = ((uint32_t)p1 & 0xFFu) | (((uint32_t)p2 & 0xFFu) << 8);  // pack_meta: no walls left
                pmeta_hi = (uint32_t)cur;
                uint32_t x = hx ^ ln_fold(pmeta_lo, 0x165667B1u) ^ ln_fold(pmeta_hi, 0xD6E8FEB9u);
                x ^= x >> 15;
                x *= 0x2C1B3C6Du;
                x ^= x >> 13;
                if (E.memo.small) {
                    const uint4* const B = reinterpret_cast<const uint4*>(E.memo.small + (size_t)(x & E.memo.small_mask) * (QZ_MEMO_S_WAYS * QZ_MEMO_S_DW));
#pragma unroll
                    for (int w = 0; w < 4; w++) {
#pragma unroll
                        for (int d = 0; d < 5; d++) pw[w][d] = B[w * (QZ_MEMO_S_DW / 4) + d];
                    }
                }
            }
        }
        // ================================================================ DESC: one level of MCTS._playout's descent (mcts.py:107-113)
        if (any_desc) {
            if (in_desc) {
                if (fused) {
                    // the old path's edge of this level is one of the children: update_recursive's step for it, before the selection reads it
                    uint32_t ql = ca[0].x, qh = ca[0].y, n = ca[0].z;
#pragma unroll
                    for (uint32_t j = 1u; j < 6u; j++) {
                        if (kold == j) {
                            ql = ca[j].x;
                            qh = ca[j].y;
                            n = ca[j].z;
                        }
                    }
                    if (kold >= 6u) {  // (a node of more than six children: never on a board without walls; kept general)
                        const uint32_t* const q = reinterpret_cast<const uint32_t*>(&pool[base + kold]);
                        ql = q[0];
                        qh = q[1];
                        n = q[2];
                    }
                    upd(ql, qh, n, ((old_len - 1u - plen) & 1u) ? old_val : -old_val);
                    ln_u32x3 nv;
                    nv.x = ql;
                    nv.y = qh;
                    nv.z = n;
                    *reinterpret_cast<ln_u32x3*>(&pool[base + kold]) = nv;
                    const uint32_t lc = (uint32_t)ne - 1u;
#pragma unroll
                    for (uint32_t j = 0u; j < 6u; j++) {
                        if ((j < lc ? j : lc) == kold) {
                            ca[j].x = ql;
                            ca[j].y = qh;
                            ca[j].z = n;
                        }
                    }
                }
#define LN_PUCT(c) (__hiloint2double((int)(c).y, (int)(c).x) + (double)(E.c_puct * __uint_as_float((c).w)) * sq / (double)(1u + (c).z))
                double best = LN_PUCT(ca[0]);   // first maximum, like max() over the children dict (mcts.py:42)
                uint32_t k = 0u, wN = ca[0].z, wcoff = cm[0].x, wmisc = cm[0].y;
#pragma unroll
                for (uint32_t j = 1u; j < 6u; j++) {  // (a lane whose node has fewer children sees its last child again: equal, never greater)
                    const double v = LN_PUCT(ca[j]);
                    if (v > best) {
                        best = v;
                        k = j;
                        wN = ca[j].z;
                        wcoff = cm[j].x;
                        wmisc = cm[j].y;
                    }
                }
                if (ne > 6) {
                    for (uint32_t j = 6u; j < (uint32_t)ne; j++) {
                        const uint32_t* const q = reinterpret_cast<const uint32_t*>(&pool[base + j]);
                        const uint4 c = *reinterpret_cast<const uint4*>(q);
                        const double v = LN_PUCT(c);
                        if (v > best) {
                            best = v;
                            k = j;
                            wN = c.z;
                            wcoff = q[4];
                            wmisc = q[5];
                        }
                    }
                }
#undef LN_PUCT
                nonfinite = nonfinite || !(best == best);
                scanned += (uint32_t)ne;
                const uint32_t e = base + k;
                const uint32_t lvl = plen;
                path_put(lvl, e);
                plen = lvl + 1u;
                pedge = e;
                {   // game.step(action), mcts.py:113: a pawn move (quoridor.py:217-243), then rotate unless the game is over
                    const uint32_t a = wmisc & 0xFFu;
                    const uint32_t K0 = 27u | (9u << 6) | (19u << 12) | (17u << 18) | (36u << 24), K1 = 0u | (20u << 6) | (16u << 12) | (28u << 18) | (26u << 24),
                                   K2 = 10u | (8u << 6);
                    const uint32_t aa = a < 12u ? a : 0u;
                    const uint32_t tb = aa < 5u ? K0 : (aa < 10u ? K1 : K2), sh = 6u * (aa < 5u ? aa : (aa < 10u ? aa - 5u : aa - 10u));
                    const int dl = (int)((tb >> sh) & 63u) - 18;
                    if (cur == 1) p1 += dl;
                    else p2 += dl;
                    if (!(p2 < 9 || p1 > 71)) cur = 3 - cur;
                }
                if (fused) {
                    if (k != kold) {  // the new path leaves the old one: the old path's levels below owe their update, and nobody passes there now
                        fused = false;
                        if (lvl + 1u < old_len) fl_i = lvl + 1u;
                        else old_len = 0u;
                    } else if (lvl + 1u >= old_len) {  // the old path's last level is paid
                        fused = false;
                        old_len = 0u;
                    }
                }
                const uint32_t cne = (wmisc >> 8) & 0xFFu;
                if (E.max_depth > 0 && plen > (uint32_t)E.max_depth) {
                    // the reference's RecursionError (drop_if_too_deep): the game is dropped, k_round_tail restarts the slot
                    E.status[b] = QZ_ABORTED;
                    atomicAdd(&E.counters[QZ_C_ABORT_DEPTH], 1ull);
                    log_dropped_game(E, b, QZ_C_ABORT_DEPTH);
                    old_len = 0u;
                    ph = LN_IDLE;
                } else if (cne == 0u) {  // TreeNode.is_leaf()
                    if (old_len > 0u) {  // (left the old path on this very level: the flush first)
                        ph = LN_FLUSH;
                        after_flush = AF_LEAF;
                    } else {
                        at_leaf();
                    }
                } else if (plen > (uint32_t)QZ_TREE_PT * QZ_PAGE_EDGES) {  // deeper than a tree has edges: corrupted storage.  Never hang the GPU
                    atomicAdd(&E.counters[QZ_C_RUNAWAY], 1ull);
                    old_len = 0u;
                    ph = LN_IDLE;
                } else {
                    base = phys(wcoff);
                    ne = (int)cne;
                    sq = sqrt_count(wN);
                    if (old_len > 0u && !fused) {  // (the descent goes on when the flush is done)
                        ph = LN_FLUSH;
                        after_flush = AF_DESC;
                    }
                }
            }
        }
        // ================================================================ FLUSH: four levels of the old path per trip
        if (any_flush) {
            if (in_flush) {
#pragma unroll
                for (uint32_t t = 0u; t < 4u; t++) {
                    const uint32_t i = fl_i + t;
                    if (i < old_len) {
                        uint32_t ql = fv[t].x, qh = fv[t].y, n = fv[t].z;
                        upd(ql, qh, n, ((old_len - 1u - i) & 1u) ? old_val : -old_val);
                        ln_u32x3 nv;
                        nv.x = ql;
                        nv.y = qh;
                        nv.z = n;
                        *reinterpret_cast<ln_u32x3*>(&pool[fe[t]]) = nv;
                    }
                }
                fl_i += 4u;
                if (fl_i >= old_len) {
                    old_len = 0u;
                    if (after_flush == AF_DESC) ph = LN_DESC;
                    else if (after_flush == AF_LEAF) at_leaf();
                    else ph = LN_IDLE;
                }
            }
        }
        // ================================================================ PROBE: the leaf's evaluation from the memo, or the miss list
        if (any_probe) {
            bool miss = false;
            if (in_probe) {
                int way = -1;
                if (E.memo.small) {
                    const uint32_t khi = pmeta_hi | (epoch << 16);
#pragma unroll
                    for (int w = 3; w >= 0; w--) {  // (the first matching entry, like memo_probe_finish)
                        if (pw[w][0].x == (uint32_t)rhb && pw[w][0].y == (uint32_t)(rhb >> 32) && pw[w][0].z == (uint32_t)rvb && pw[w][0].w == (uint32_t)(rvb >> 32) &&
                            pw[w][1].x == pmeta_lo && pw[w][1].y == khi)
                            way = w;
                    }
                }
                if (way >= 0) {
                    uint4 h1 = pw[0][1], h2 = pw[0][2], h3 = pw[0][3], h4 = pw[0][4];
#pragma unroll
                    for (int w = 1; w < 4; w++) {
                        if (way == w) {
                            h1 = pw[w][1];
                            h2 = pw[w][2];
                            h3 = pw[w][3];
                            h4 = pw[w][4];
                        }
                    }
                    const float pp[12] = {__uint_as_float(h2.x), __uint_as_float(h2.y), __uint_as_float(h2.z), __uint_as_float(h2.w),
                                          __uint_as_float(h3.x), __uint_as_float(h3.y), __uint_as_float(h3.z), __uint_as_float(h3.w),
                                          __uint_as_float(h4.x), __uint_as_float(h4.y), __uint_as_float(h4.z), __uint_as_float(h4.w)};
                    c_hits++;
                    expand(pedge, h1.w & 0xFFFu, pp);
                    complete((double)__uint_as_float(h1.z), 0u);
                } else {
                    miss = true;
                }
            }
            // ---- leaves for the network: ONE atomic for the wavefront's misses of this iteration
            const uint64_t mm = __ballot(miss);
            if (mm != 0ull) {
                const int leader = __ffsll((unsigned long long)mm) - 1;
                uint32_t s0 = 0u;
                if (lane == leader) s0 = (uint32_t)atomicAdd(E.miss_count + par, (int)__popcll(mm));
                s0 = rdl(s0, leader);
                if (miss) {
                    const uint32_t s = s0 + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                    if (s >= (uint32_t)E.n_boards) {  // a stale counter (must not happen): the board forgets this descent and repeats it
                        atomicAdd(&E.counters[QZ_C_MISS_OVERFLOW], 1ull);
                    } else {
                        E.miss_hb[s] = rhb;
                        E.miss_vb[s] = rvb;
                        E.miss_meta[s] = (uint64_t)pmeta_lo | ((uint64_t)pmeta_hi << 32);
                        E.pend_slot[b] = s;
                        E.leaf_pedge[b] = pedge;
                        E.path_len[b] = plen;
                        const uint32_t nm = plen < (uint32_t)LN_LCAP ? plen : (uint32_t)LN_LCAP;
                        for (uint32_t i = 0u; i < nm; i++) gpath[i] = lp[i * 64u];  // the path, for the backup of the launch that gets the answer
                        c_evals++;
                        waiting = true;
                    }
                    c_scanned += scanned;   // (the descent's reads were made; its levels are counted with the backup, like k_advance)
                    c_maxdepth = plen > c_maxdepth ? plen : c_maxdepth;
                    if (nonfinite) c_nonfinite++;
                    ph = LN_IDLE;
                }
            }
        }
    }

    // ---- the board's state and counters back to memory
    if (on) {
        E.root_N[b] = rootN;
        E.root_ne[b] = root_ne;
        E.root_eoff[b] = root_eoff;
        E.n_nodes[b] = nn;
        E.n_edges[b] = neu;
        E.tree_npages[tree_slot(E, b, half)] = np;
        E.pl_done[b] = done;
        if (!waiting && slot0 != QZ_NONE) E.pend_slot[b] = QZ_NONE;
        E.bc_playouts[b] += c_playouts;
        E.bc_terminal[b] += c_terminal;
        E.bc_overflow[b] += c_overflow;
        E.bc_nonfinite[b] += c_nonfinite;
        if (c_maxdepth > E.bc_maxdepth[b]) E.bc_maxdepth[b] = c_maxdepth;
        E.bc_memo_hits[b] += c_hits;
        E.bc_evals[b] += c_evals;
        E.bc_levels[b] += (unsigned long long)c_levels;
        E.bc_scanned[b] += (unsigned long long)c_scanned;
        E.bc_expanded[b] += (unsigned long long)c_expanded;
    }
}

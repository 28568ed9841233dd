// qz_movegen_pool.h -- the pooled formulation of Quoridor.actions() + state() for a TILE of
// boards (one 256-thread workgroup = NB boards).  Instead of giving every board a whole
// wavefront for every phase, each phase maps lanes to whatever it has many of:
//
//   P0  lane = board              context: blocked sets, static slot tests, pawn moves,
//                                 jump plans, encoder bitmap                      (NB lanes)
//   P1  lane = (board, player)    one concrete path per player, ORDERED           (2*NB lanes)
//   P2  lane = (board, slot)      which (candidate wall, player) pairs need a
//                                 reachability re-check -> pooled work list       (64*NB tasks)
//   P3  lane = work item          flood with the candidate wall, early exit on the
//                                 goal row OR on the intact tail of the base path  (~20*NB items)
//   P4  lane = board              legal sets = static & ~blocked -> 140-bit mask
//   P5  lane = 16 B of output     26x9x9 planes, 16-byte coalesced stores
//
// On the device P0+P1 are one launch (k_pool_paths, lane = (board, player), every lane busy)
// that leaves a PoolBoard + two PathTab records per board in an HBM scratch area, and P2..P5
// are a second launch (k_pool_tiles, one workgroup per tile of boards).  The phase bodies
// below are plain per-lane functions over those records; tests/hostcheck runs the very same
// functions lane by lane on the CPU to check them against the oracle.
#pragma once
#include "qz_rules.h"
#include "qz_path_rows.h"

namespace qz {

constexpr int POOL_MAX_LAYERS = 40;   // edges of a base path the tables can hold; longer paths fall back
constexpr int POOL_PATH_CAP = 30;     // tiles of an ordered path (>= POOL_MAX_LAYERS + 1)

struct PoolBoard {          // per board, written in P0/P1, read by P2..P5
    Board b;
    Blk base;
    uint64_t sh, sv;
    uint32_t pawn;
    uint32_t flags;         // bit0 mover has walls, bit1 terminal (skip), bit2 both players connected
    JumpPlan plan[2];       // plan[p-1]: jumps around player p's opponent
    PathEdges pe[2];
    int len[2];
    int lastjump[2];        // reverse position of the jump edge closest to the goal, -1 if none
    int farjump[2];         // ... of the jump edge closest to the start, -1 if none
    BB tiles[2];            // every tile of the base path, the pawn's included
    uint32_t blocked[4];    // H lo, H hi, V lo, V hi: slots whose wall would cut somebody off
    uint64_t need[4];       // slots whose wall touches a base path: (H,p1) (H,p2) (V,p1) (V,p2)
};

// What P3 needs to know about a player's base path to stop a flood early (positions count
// edges from the GOAL end, see find_path_tables()):
//   srcpos[t] = k  if the path's k-th edge from the end leaves tile t (255 otherwise; a shortest
//                  path visits a tile once, so this is well defined)
//   suffix[k]      = tiles behind that edge: from any of them the goal stays reachable as long
//                    as no edge closer to the goal is removed
struct PathTab {
    uint8_t srcpos[84];
    BB suffix[POOL_MAX_LAYERS + 1];
};

// ---- P0 ---------------------------------------------------------------------------------
QZ_HD void enc_build(const Board& b, uint32_t* enc, uint32_t& hot) {
    int pm = b.cur == 1 ? b.p1 : b.p2, po = b.cur == 1 ? b.p2 : b.p1;
    if (pm < 0) pm += 81;
    if (po < 0) po += 81;
    BB pl[5] = {spread8(~(b.hb | b.vb)), spread8(b.vb), spread8(b.hb), dest_bit(pm), dest_bit(po)};
    for (int i = 0; i < 13; i++) enc[i] = 0u;
    for (int k = 0; k < 5; k++) {
        int pos = 81 * k, w = pos >> 5, off = pos & 31;
        uint32_t a0 = pl[k].w0, a1 = pl[k].w1, a2 = pl[k].w2;  // 81 bits
        enc[w] |= a0 << off;
        if (off) {
            enc[w + 1] |= (a0 >> (32 - off)) | (a1 << off);
            enc[w + 2] |= (a1 >> (32 - off)) | (a2 << off);
            if (w + 3 < 13) enc[w + 3] |= a2 >> (32 - off);
        } else {
            enc[w + 1] |= a1;
            enc[w + 2] |= a2;
        }
    }
    int wm = b.cur == 1 ? b.w1 : b.w2, wo = b.cur == 1 ? b.w2 : b.w1;
    int im = wm - 1, io = wo - 1;  // Python index -1 -> last plane (quoridor.py:79-80)
    if (im < 0) im += 10;
    if (io < 0) io += 10;
    hot = (uint32_t)(5 + im) | ((uint32_t)(15 + io) << 8) | ((uint32_t)(b.cur == 2 ? 25 : 255) << 16);
}

QZ_HD void pool_p0(PoolBoard& c, const Board& b, bool terminal, bool want_moves) {
    c.b = b;
    c.flags = terminal ? 2u : 0u;
    for (int i = 0; i < 4; i++) c.blocked[i] = 0u;
    c.len[0] = c.len[1] = 0;
    c.pawn = 0u;
    c.sh = c.sv = 0ull;
    c.pe[0].found = c.pe[1].found = false;
    if (terminal || !want_moves) return;
    c.base = blk_or(blocked_from(spread8(b.hb), spread8(b.vb)), blocked_borders());
    c.sh = static_ok_h(b.hb, b.vb);
    c.sv = static_ok_v(b.hb, b.vb);
    int loc = b.cur == 1 ? b.p1 : b.p2, opp = b.cur == 1 ? b.p2 : b.p1;
    c.pawn = pawn_actions_tab(b.hb, b.vb, loc, opp, b.cur);
    if ((b.cur == 1 ? b.w1 : b.w2) > 0) c.flags |= 1u;  // quoridor.py:149-150
    c.plan[0] = make_jump_plan(b.hb, b.vb, b.p2);  // player 1's opponent
    c.plan[1] = make_jump_plan(b.hb, b.vb, b.p1);
}

// ---- P0 + P1 as ONE lane task (device launch 1: lane = (board, player)) -------------------
// Both lanes of a board derive the board context redundantly (cheap, and it keeps every lane
// busy); lane p == 1 stores the shared part of the record, each lane stores its own path.
// Three steps so that a kernel can put its own path finder in the middle (k_wave_rules runs the
// search on nine lanes per player, qz_path_rows.h): pool_k1_pre -> search -> pool_k1_post.
struct K1Pre {
    Blk base;
    bool walls;  // the mover has walls left: blocked sets, slots, jump plans and paths are needed
};
QZ_HD K1Pre pool_k1_pre(const Board& b, bool terminal, bool want_moves, int p, PoolBoard& out, Graph& g) {
    const bool live = !terminal && want_moves;
    K1Pre k;
    k.walls = live && ((b.cur == 1 ? b.w1 : b.w2) > 0);
    k.base.n = k.base.s = k.base.e = k.base.w = bb_zero();
    // a mover without walls only has pawn moves: no blocked sets, wall slots, jump plans or paths
    // (95 % of the leaf boards of a 400-playout self-play run, benchmarks/insitu_leaf_stats.py)
    if (k.walls) k.base = blk_or(blocked_from(spread8(b.hb), spread8(b.vb)), blocked_borders());
    if (p == 1) {
        out.b = b;
        out.flags = (terminal ? 2u : 0u) | (k.walls ? 1u : 0u);
        for (int i = 0; i < 4; i++) out.blocked[i] = 0u;
        out.base = k.base;
        out.sh = k.walls ? static_ok_h(b.hb, b.vb) : 0ull;
        out.sv = k.walls ? static_ok_v(b.hb, b.vb) : 0ull;
        int loc = b.cur == 1 ? b.p1 : b.p2, opp = b.cur == 1 ? b.p2 : b.p1;
        out.pawn = live ? pawn_actions_tab(b.hb, b.vb, loc, opp, b.cur) : 0u;
    }
    if (k.walls) {
        JumpPlan plan = make_jump_plan(b.hb, b.vb, side_opp(b, p));
        out.plan[p - 1] = plan;
        g = make_graph_plan(k.base, plan, -1, false);
    }
    return k;
}
QZ_HD void pool_need_masks(const Board& b, int p, const K1Pre& k, const PathEdges& pe, int detour_mode, const CutMasks* cuts, uint64_t& nh_out,
                           uint64_t& nv_out);
// `op`: the base path of player p (found == false, len == 0 where there was nothing to search)
// `cuts`: path_cut_masks(op.e) if the caller has it already (k_wave_rules gets it from the search), else nullptr
QZ_HD void pool_k1_post(const Board& b, int p, PoolBoard& out, const K1Pre& k, const OrderedPath& op, int lj, int fj, int detour_mode,
                        const CutMasks* cuts = nullptr) {
    const PathEdges& pe = op.e;
    const int len = op.len;
    out.pe[p - 1] = pe;
    out.len[p - 1] = len;
    out.lastjump[p - 1] = lj;
    out.farjump[p - 1] = fj;
    out.tiles[p - 1] = len > 0 ? bb_or(op.last, bb_bit(side_start(b, p))) : bb_zero();
    uint64_t nh, nv;
    pool_need_masks(b, p, k, pe, detour_mode, cuts, nh, nv);
    out.need[p - 1] = nh;
    out.need[2 + p - 1] = nv;
}
// which wall slots need a flood for player p: the candidates that remove an edge of p's base path (or, if the path jumps,
// that sit next to the opponent), minus those a group detour clears
QZ_HD void pool_need_masks(const Board& b, int p, const K1Pre& k, const PathEdges& pe, int detour_mode, const CutMasks* cuts, uint64_t& nh_out,
                           uint64_t& nv_out) {
    uint64_t nh = 0, nv = 0;
    if (pe.found) {
        const CutMasks cm = cuts ? *cuts : path_cut_masks(pe);
        uint64_t near = pe.jump ? near_opp_mask(side_opp(b, p)) : 0ull;
        nh = static_ok_h(b.hb, b.vb) & (cm.h | near);
        nv = static_ok_v(b.hb, b.vb) & (cm.v | near);
        // One flood instead of one per candidate: take away EVERY edge that any of those candidates
        // would remove (blocked sets are a union over walls, so this is blocked_from of the candidate
        // set) and all jump edges.  If the goal is still reachable by simple moves, that detour
        // survives each single candidate, so none of them can cut this player off.  If not, the
        // horizontal and the vertical candidates are tried as two smaller groups (detour_mode 2).
        // Floods per board on the synthetic sets: 20 -> 10 (one group) -> 4 (three groups).
        if (detour_mode >= 1 && (nh | nv) != 0ull) {
            const int start = side_start(b, p), opp = side_opp(b, p);
            const BB goal = side_goal(p);
            Graph g2 = make_graph_nojump(blk_or(k.base, blocked_from(spread8(nh), spread8(nv))), opp);
            if (flood_to(g2, bb_bit(start), goal)) {
                nh = nv = 0ull;
            } else if (detour_mode >= 2 && nh != 0ull && nv != 0ull) {
                g2 = make_graph_nojump(blk_or(k.base, blocked_from(spread8(nh), bb_zero())), opp);
                if (flood_to(g2, bb_bit(start), goal)) nh = 0ull;
                g2 = make_graph_nojump(blk_or(k.base, blocked_from(bb_zero(), spread8(nv))), opp);
                if (flood_to(g2, bb_bit(start), goal)) nv = 0ull;
            }
        }
    }
    nh_out = nh;
    nv_out = nv;
}
// FINDER 0: one search per lane on three-word sets (find_path_tables); 1: the nine-rows formulation in
// its array form (find_path_rows; what the host check runs in place of the SIMT form of k_wave_rules)
template <int FINDER = 0>
QZ_HD void pool_k1(const Board& b, bool terminal, bool want_moves, int p, PoolBoard& out, PathTab& tab, int detour_mode = 0) {
    Graph g;
    const K1Pre k = pool_k1_pre(b, terminal, want_moves, p, out, g);
    OrderedPath op;
    op.e.pn = op.e.ps = op.e.pe = op.e.pw = bb_zero();
    op.e.jump = false;
    op.e.found = false;
    op.len = 0;
    op.last = bb_zero();
    int lj = -1, fj = -1;
    if (k.walls) {
        if (FINDER == 0) op = find_path_tables(g, side_start(b, p), side_goal(p), POOL_MAX_LAYERS + 1, tab, lj, fj);
        else op = find_path_rows(g, side_start(b, p), side_goal(p), POOL_MAX_LAYERS + 1, tab, lj, fj);
    }
    pool_k1_post(b, p, out, k, op, lj, fj, detour_mode);
}

// ---- the hand-off between the two launches of the pooled pipeline ---------------------------------------------------
// Round 3 wrote a PoolBoard and two PathTab records per board into the scratch area (~1.4 KB written, ~0.6 KB read back:
// 1.30 x the op's algorithmic traffic).  What the second launch cannot cheaply recompute from the 24-byte board is only:
// the two base paths -- as the bare tile sequence, goal end first -- the need masks (group detours already applied), the
// jump plans and the pawn moves: 184 bytes.  Blocked sets and static slot tests are recomputed from the board; path edge
// sets, path tiles, jump positions and the srcpos table are rebuilt from the sequences (pool_hand_rebuild_path); a suffix set
// is the OR of a prefix of the sequence, taken when a flood needs it (pool_p3_seq).
struct PathSeq {
    uint8_t len;                        // edges; 0: no path (or nothing to search); 255: found but longer than the tables
    uint8_t goal;                       // the goal-row tile the path ends on
    uint8_t src[POOL_MAX_LAYERS + 2];   // src[k]: the tile edge k leaves, k counting from the GOAL end (src[len - 1] = the pawn's tile)
};
struct PoolHand {
    uint32_t pawn;
    uint32_t flags;        // PoolBoard::flags
    uint64_t need[4];
    JumpPlan plan[2];
    PathSeq seq[2];
};
static_assert(sizeof(PathSeq) == 44 && sizeof(PoolHand) == 184, "hand-off record layout");

// launch 1, lane = (board, player): pool_k1 with the hand-off record as its only output
QZ_HD void pool_k1_hand(const Board& b, bool terminal, int p, PoolHand& out, int detour_mode) {
    // (pool_k1_pre / pool_k1_post without a PoolBoard in between: a record indexed by the player would live in scratch memory)
    const bool live = !terminal;
    K1Pre k;
    k.walls = live && ((b.cur == 1 ? b.w1 : b.w2) > 0);
    k.base.n = k.base.s = k.base.e = k.base.w = bb_zero();
    if (k.walls) k.base = blk_or(blocked_from(spread8(b.hb), spread8(b.vb)), blocked_borders());
    if (p == 1) {
        const int loc = b.cur == 1 ? b.p1 : b.p2, opp = b.cur == 1 ? b.p2 : b.p1;
        out.pawn = live ? pawn_actions_tab(b.hb, b.vb, loc, opp, b.cur) : 0u;
        out.flags = (terminal ? 2u : 0u) | (k.walls ? 1u : 0u);
    }
    PathSeq& sq = out.seq[p - 1];
    OrderedPath op;
    op.e.pn = op.e.ps = op.e.pe = op.e.pw = bb_zero();
    op.e.jump = false;
    op.e.found = false;
    op.len = 0;
    op.last = bb_zero();
    int lj = -1, fj = -1;
    if (k.walls) {
        const JumpPlan plan = make_jump_plan(b.hb, b.vb, side_opp(b, p));
        out.plan[p - 1] = plan;
        const Graph g = make_graph_plan(k.base, plan, -1, false);
        op = find_path_walk(
            g, side_start(b, p), side_goal(p), POOL_MAX_LAYERS + 1, lj, fj, []() {},
            [&](int kk, int s, int t, const BB&) {
                if (kk == 0) sq.goal = (uint8_t)t;
                sq.src[kk] = (uint8_t)s;
            });
    }
    sq.len = !op.e.found ? 0 : (op.len < 0 ? 255 : (uint8_t)op.len);
    uint64_t nh, nv;
    pool_need_masks(b, p, k, op.e, detour_mode, nullptr, nh, nv);
    out.need[p - 1] = nh;
    out.need[2 + p - 1] = nv;
}
// launch 2, lane = board: the board's context from the board itself + the record
QZ_HD void pool_hand_rebuild_board(PoolBoard& c, const PoolHand& h, const Board& b) {
    c.b = b;
    c.flags = h.flags;
    c.pawn = h.pawn;
    for (int i = 0; i < 4; i++) c.blocked[i] = 0u;
    for (int i = 0; i < 4; i++) c.need[i] = h.need[i];
    const bool walls = (h.flags & 3u) == 1u;
    c.base.n = c.base.s = c.base.e = c.base.w = bb_zero();
    if (walls) c.base = blk_or(blocked_from(spread8(b.hb), spread8(b.vb)), blocked_borders());
    c.sh = walls ? static_ok_h(b.hb, b.vb) : 0ull;
    c.sv = walls ? static_ok_v(b.hb, b.vb) : 0ull;
    c.plan[0] = h.plan[0];
    c.plan[1] = h.plan[1];
}
// launch 2, lane = (board, player): edge sets, path tiles, jump positions and the srcpos table (84 bytes, pre-filled with 255
// by the caller) from the tile sequence
QZ_HD void pool_hand_rebuild_path(PoolBoard& c, int p, const PathSeq& sq, uint8_t* srcpos) {
    PathEdges e;
    e.pn = e.ps = e.pe = e.pw = bb_zero();
    e.jump = false;
    e.found = sq.len != 0;
    int len = sq.len == 255 ? -1 : (int)sq.len, lj = -1, fj = -1;
    BB tiles = bb_zero();
    if (len < 0) {  // found but longer than the tables: every candidate gets a full re-check (find_path_walk's conservative answer)
        e.pn = e.ps = e.pe = e.pw = bb_not(bb_zero());
        e.jump = true;
    }
    int t = sq.goal;
    for (int k = 0; k < len; k++) {
        const int s = sq.src[k], d = t - s;
        if (d == 9) e.pn = bb_or(e.pn, bb_bit(s));
        else if (d == -9) e.ps = bb_or(e.ps, bb_bit(s));
        else if (d == 1) e.pe = bb_or(e.pe, bb_bit(s));
        else if (d == -1) e.pw = bb_or(e.pw, bb_bit(s));
        else {
            e.jump = true;
            if (lj < 0) lj = k;
            fj = k;
        }
        tiles = bb_or(tiles, bb_bit(t));
        srcpos[s] = (uint8_t)k;
        t = s;
    }
    if (len > 0) tiles = bb_or(tiles, bb_bit(side_start(c.b, p)));
    c.pe[p - 1] = e;
    c.len[p - 1] = len;
    c.lastjump[p - 1] = lj;
    c.farjump[p - 1] = fj;
    c.tiles[p - 1] = tiles;
}
// PathTab::suffix[k] from the sequence: the goal tile and the sources of the edges closer to the goal than edge k
QZ_HD BB pool_seq_suffix(const PathSeq& sq, int k) {
    BB acc = bb_bit(sq.goal);
    for (int j = 0; j < k; j++) acc = bb_or(acc, bb_bit(sq.src[j]));
    return acc;
}

// ---- P2 ---------------------------------------------------------------------------------
// returns a 4-bit need mask for slot ix: bit0 (H,p1) bit1 (H,p2) bit2 (V,p1) bit3 (V,p2)
QZ_HD uint32_t pool_p2_ref(const PoolBoard& c, int ix) {
    if ((c.flags & 3u) != 1u) return 0u;
    if (!(c.pe[0].found && c.pe[1].found)) return 0u;  // handled in P4: nothing is legal
    bool stH = (c.sh >> ix) & 1ull, stV = (c.sv >> ix) & 1ull;
    uint32_t m = 0;
    bool n1 = near_opp(ix, c.b.p2), n2 = near_opp(ix, c.b.p1);
    if (stH) {
        Blk d = candidate_delta_fast(ix, true);
        if (cuts(d, c.pe[0]) || (c.pe[0].jump && n1)) m |= 1u;
        if (cuts(d, c.pe[1]) || (c.pe[1].jump && n2)) m |= 2u;
    }
    if (stV) {
        Blk d = candidate_delta_fast(ix, false);
        if (cuts(d, c.pe[0]) || (c.pe[0].jump && n1)) m |= 4u;
        if (cuts(d, c.pe[1]) || (c.pe[1].jump && n2)) m |= 8u;
    }
    return m;
}
// the same answer from the per-board need masks the path lanes left behind
QZ_HD uint32_t pool_p2(const PoolBoard& c, int ix) {
    if ((c.flags & 3u) != 1u) return 0u;
    if (!(c.pe[0].found && c.pe[1].found)) return 0u;  // handled in P4: nothing is legal
    return (uint32_t)((c.need[0] >> ix) & 1ull) | ((uint32_t)((c.need[1] >> ix) & 1ull) << 1) |
           ((uint32_t)((c.need[2] >> ix) & 1ull) << 2) | ((uint32_t)((c.need[3] >> ix) & 1ull) << 3);
}
QZ_HD uint32_t pool_item(int board, int ix, bool horizontal, int p) {
    return ((uint32_t)board << 8) | (uint32_t)ix | (horizontal ? 0x40u : 0u) | (p == 2 ? 0x80u : 0u);
}

// ---- P3 ---------------------------------------------------------------------------------
// true if player p can still reach its goal with the candidate wall added
template <typename SuffixFn>
QZ_HD bool pool_p3_with(const PoolBoard& c, uint32_t item, const uint8_t* srcpos, SuffixFn suffix) {
    int ix = (int)(item & 63u);
    bool hz = (item & 0x40u) != 0u;
    int p = (item & 0x80u) ? 2 : 1;
    Blk d = candidate_delta_fast(ix, hz);
    BB target = side_goal(p);
    BB from = bb_bit(side_start(c.b, p));
    if (c.len[p - 1] > 0) {
        const PathEdges& e = c.pe[p - 1];
        // source tiles of the path edges this candidate removes
        BB hit = bb_or(bb_or(bb_and(d.n, e.pn), bb_and(d.s, e.ps)), bb_or(bb_and(d.e, e.pe), bb_and(d.w, e.pw)));
        // `best`: the removed edge closest to the goal -- everything behind it is still connected to
        // the goal.  `worst`: the removed edge closest to the pawn -- everything in front of it is
        // still connected to the pawn.  A wall next to the opponent may also change the jump
        // edges, so those count as removed.  The flood then runs from the front part to the back
        // part of the path: a detour around the wall instead of the whole way from the pawn.
        const bool jumps = e.jump && near_opp(ix, side_opp(c.b, p)) && c.lastjump[p - 1] >= 0;
        int best = jumps ? c.lastjump[p - 1] : 255;
        int worst = jumps ? c.farjump[p - 1] : -1;
        for (int guard = 0; guard < 8 && bb_any(hit); guard++) {
            int t = bb_lowest(hit);
            hit = bb_andn(hit, bb_bit(t));
            int pos = srcpos[t];
            best = pos < best ? pos : best;
            worst = pos > worst ? pos : worst;
        }
        if (best == 255) return true;  // nothing on the path is touched
        const BB behind_best = suffix(best);
        target = bb_or(target, behind_best);
        // tiles of the path in front of edge `worst` = all path tiles minus (its destination and
        // everything behind it); one removed edge is the common case: no second table read
        const BB behind_worst = worst == best ? behind_best : suffix(worst);
        from = bb_andn(c.tiles[p - 1], behind_worst);
    }
    Graph g = make_graph_plan(blk_or(c.base, d), c.plan[p - 1], ix, hz);
    return flood_to(g, from, target);
}

QZ_HD bool pool_p3(const PoolBoard& c, uint32_t item, const uint8_t* srcpos, const BB* suffix) {
    return pool_p3_with(c, item, srcpos, [&](int k) { return suffix[k]; });
}
QZ_HD bool pool_p3(const PoolBoard& c, uint32_t item, const PathTab& tab) { return pool_p3(c, item, tab.srcpos, tab.suffix); }
// the pooled pipeline's form: srcpos rebuilt by pool_hand_rebuild_path, suffix sets from the tile sequence
QZ_HD bool pool_p3_seq(const PoolBoard& c, uint32_t item, const uint8_t* srcpos, const PathSeq& sq) {
    return pool_p3_with(c, item, srcpos, [&](int k) { return pool_seq_suffix(sq, k); });
}

// ---- P4 ---------------------------------------------------------------------------------
QZ_HD void pool_p4(const PoolBoard& c, uint32_t mask5[5]) {
    uint64_t lh = 0, lv = 0;
    if ((c.flags & 3u) == 1u && c.pe[0].found && c.pe[1].found) {
        uint64_t bh = (uint64_t)c.blocked[0] | ((uint64_t)c.blocked[1] << 32);
        uint64_t bv = (uint64_t)c.blocked[2] | ((uint64_t)c.blocked[3] << 32);
        lh = c.sh & ~bh;
        lv = c.sv & ~bv;
    }
    mask5[0] = c.pawn | (uint32_t)(lh << 12);
    mask5[1] = (uint32_t)(lh >> 20);
    mask5[2] = (uint32_t)(lh >> 52) | (uint32_t)(lv << 12);
    mask5[3] = (uint32_t)(lv >> 20);
    mask5[4] = (uint32_t)(lv >> 52);
}

// ---- P5 prologue: word k (bits 32k..32k+31) of the board's 2,106-bit state() bitmap ----------
constexpr int POOL_BM_WORDS = 66;
constexpr int QZ_POOL_PLANES_N = 2106;  // 26 * 81 floats of state() per board
constexpr int POOL_STREAM_WORDS(int nb) { return (nb * QZ_POOL_PLANES_N + 31) / 32; }
struct EncCtx {          // what the encoder needs to know about one board
    uint32_t enc[13];    // planes 0..4 as a 405-bit string
    uint32_t hot;        // all-ones planes among 5..25, one per byte (255 = none)
    uint32_t terminal;   // terminal leaf: all-zero planes
};
QZ_HD void enc_ctx_build(EncCtx& c, const Board& b, bool terminal) {
    enc_build(b, c.enc, c.hot);
    c.terminal = terminal ? 1u : 0u;
}
QZ_HD uint32_t pool_bitmap_word(const EncCtx& c, int k) {
    if (c.terminal) return 0u;  // terminal leaf: all-zero planes
    uint32_t val = k < 13 ? c.enc[k] : 0u;
    const int lo_w = 32 * k, hi_w = lo_w + 32;
    for (int j = 0; j < 3; j++) {
        int h = (int)((c.hot >> (8 * j)) & 0xFFu);
        if (h > 25) continue;
        int lo = 81 * h, hi = lo + 81;
        lo = lo > lo_w ? lo : lo_w;
        hi = hi < hi_w ? hi : hi_w;
        if (hi > lo) {
            int n = hi - lo;
            uint32_t m = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
            val |= m << (lo - lo_w);
        }
    }
    return val;
}
// the four output floats starting at element idx (even, <= 2104) of a board, as a 4-bit value;
// `bm` = this board's bitmap, `next0` = word 0 of the next board's bitmap (for idx == 2104)
QZ_HD uint32_t pool_bitmap_nibble(const uint32_t* bm, uint32_t next0, int idx) {
    int w = idx >> 5, sh = idx & 31;
    uint64_t two = (uint64_t)bm[w] | ((uint64_t)(w + 1 < POOL_BM_WORDS ? bm[w + 1] : 0u) << 32);
    uint32_t nib = (uint32_t)(two >> sh) & 15u;
    if (idx == 2104) nib = (nib & 3u) | ((next0 & 3u) << 2);
    return nib;
}

// The encoder groups write a tile of boards as ONE bit stream (2,106 bits per board, boards back
// to back, no padding): output float4 q of the tile is then bits 4q..4q+3 of the stream -- a
// nibble that never straddles a word, at a lane-invariant shift when q advances by 256.
// word w of that stream, from the tile's word-aligned bitmaps (`bm`, POOL_BM_WORDS per board,
// followed by one all-zero pad board; bits past 2,106 of a board's last word are zero)
QZ_HD uint32_t pool_stream_word(const uint32_t* bm, int w) {
    const int bit0 = 32 * w;
    const int bd = bit0 / QZ_POOL_PLANES_N, o = bit0 - bd * QZ_POOL_PLANES_N;
    const int k = o >> 5, sh = o & 31;
    const uint32_t* b = bm + bd * POOL_BM_WORDS;
    const uint32_t lo = b[k], hi = (k + 1 < POOL_BM_WORDS) ? b[k + 1] : 0u;
    uint32_t v = (uint32_t)((((uint64_t)hi << 32) | (uint64_t)lo) >> sh);
    const int rem = QZ_POOL_PLANES_N - o;  // bits this board still has from `o` on (>= 1)
    if (rem < 32) v |= b[POOL_BM_WORDS] << rem;  // the rest comes from word 0 of the next board
    return v;
}
QZ_HD uint32_t pool_stream_nibble(const uint32_t* st, int q) { return (st[q >> 3] >> ((q & 7) << 2)) & 15u; }

// ---- P5 (reference form, one element) ---------------------------------------------------
QZ_HD float pool_plane_value(const EncCtx& c, int idx) {
    if (c.terminal) return 0.0f;
    if (idx < 405) return (float)((c.enc[idx >> 5] >> (idx & 31)) & 1u);
    uint32_t plane = (uint32_t)idx / 81u;
    uint32_t h = c.hot;
    bool on = plane == (h & 0xFFu) || plane == ((h >> 8) & 0xFFu) || plane == ((h >> 16) & 0xFFu);
    return on ? 1.0f : 0.0f;
}

}  // namespace qz

// hostcheck.cpp -- TEST-ONLY g++ build of alphazero_quoridor_amd/csrc/qz_rules.h.
//
// Emulates, lane by lane, exactly what the HIP movegen kernel does with the shared
// per-lane functions, so the bitboard formulation (closed-form corners, linear blocked
// sets, path-cut pruning) can be compared with the oracle on a machine without a GPU.
// It is never loaded by the product (alphazero_quoridor_amd/ does not know it exists).
#include <cstdint>
#include <cstring>

#include "../../alphazero_quoridor_amd/csrc/qz_rules.h"
#include "../../alphazero_quoridor_amd/csrc/qz_movegen_pool.h"
#include <vector>

using namespace qz;

extern "C" {

int hc_corner(uint64_t hb, uint64_t vb, int t, int which) {
    int a = corner(hb, vb, t, which), b = corner_tab(hb, vb, t, which);  // closed form == constant table
    if (a != b || corner_ref(t, which) != corner_ref_tab(t, which)) return -99;
    return a;
}

uint32_t hc_pawn_actions(uint64_t hb, uint64_t vb, int loc, int opp, int player) {
    uint32_t a = pawn_actions(hb, vb, loc, opp, player), b = pawn_actions_tab(hb, vb, loc, opp, player);
    return a == b ? a : 0xFFFFFFFFu;
}

// reachability of player p (1|2) with pawns at p1/p2 on walls hb/vb
int hc_reach(uint64_t hb, uint64_t vb, int p1, int p2, int p) {
    Board b = opening();
    b.hb = hb;
    b.vb = vb;
    b.p1 = p1;
    b.p2 = p2;
    MoveCtx c = make_ctx(b);
    Graph g = make_graph(c.base, hb, vb, side_opp(b, p));
    bool f = flood(g, side_start(b, p), side_goal(p));
    BB layers[96];
    PathEdges pe = base_path(c, p, layers);
    if (pe.found != f) return -1;  // the two floods must agree
    return f ? 1 : 0;
}

// mode 0: the kernel's algorithm (path-cut pruning); mode 1: brute force (flood every candidate)
static void movegen_one(const Board& b, uint32_t* mask5, int mode, int64_t* floods) {
    MoveCtx c = make_ctx(b);
    uint64_t lh = 0, lv = 0;
    if (c.walls) {
        BB layers[96];
        PathEdges path[3];
        path[1] = base_path(c, 1, layers);
        path[2] = base_path(c, 2, layers);
        for (int cand = 0; cand < 128; cand++) {  // "lanes"
            int ix = cand & 63;
            bool hz = cand < 64;
            bool st = ((hz ? c.sh : c.sv) >> ix) & 1ull;
            if (!st) continue;
            bool ok = true;
            if (!path[1].found || !path[2].found) {
                ok = false;
            } else {
                Blk d = candidate_delta(ix, hz);
                for (int p = 1; p <= 2 && ok; p++) {
                    bool need = mode == 1 ? true : needs_check(c, path[p], p, ix, d);
                    if (need) {
                        if (floods) (*floods)++;
                        ok = candidate_reaches(c, p, ix, hz, d);
                    }
                }
            }
            if (ok) {
                if (hz) lh |= 1ull << ix;
                else lv |= 1ull << ix;
            }
        }
    }
    // 140-bit mask: [pawn 12][H 64][V 64]
    mask5[0] = c.pawn | (uint32_t)(lh << 12);
    mask5[1] = (uint32_t)(lh >> 20);
    mask5[2] = (uint32_t)(lh >> 52) | (uint32_t)(lv << 12);
    mask5[3] = (uint32_t)(lv >> 20);
    mask5[4] = (uint32_t)(lv >> 52);
}

void hc_movegen(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n, uint32_t* mask5,
                int mode, int64_t* floods) {
    if (floods) *floods = 0;
    for (int i = 0; i < n; i++) movegen_one(unpack(hb[i], vb[i], meta[i]), mask5 + 5 * (long)i, mode, floods);
}

void hc_encode(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n, float* planes) {
    for (int i = 0; i < n; i++) {
        Board b = unpack(hb[i], vb[i], meta[i]);
        for (int k = 0; k < 26 * 81; k++) planes[(long)i * 2106 + k] = plane_value(b, k);
    }
}

void hc_step(uint64_t* hb, uint64_t* vb, uint64_t* meta, const uint8_t* action, int n, uint8_t* done,
             uint8_t* winner) {
    for (int i = 0; i < n; i++) {
        Board b = unpack(hb[i], vb[i], meta[i]);
        done[i] = apply_action(b, action[i]) ? 1 : 0;
        winner[i] = (uint8_t)winner_of(b);
        hb[i] = b.hb;
        vb[i] = b.vb;
        meta[i] = pack_meta(b);
    }
}

// ordered action list from a mask via order_index (the expand kernel's slot computation)
int hc_ordered(const uint32_t* mask5, int* out) {
    uint32_t pawn = mask5[0] & 0xFFFu;
    uint64_t lh = ((uint64_t)mask5[0] >> 12) | ((uint64_t)mask5[1] << 20) | ((uint64_t)(mask5[2] & 0xFFFu) << 52);
    uint64_t lv = ((uint64_t)mask5[2] >> 12) | ((uint64_t)mask5[3] << 20) | ((uint64_t)(mask5[4] & 0xFFFu) << 52);
    int n = 0;
    for (int a = 0; a < 140; a++) {
        bool on = (mask5[a >> 5] >> (a & 31)) & 1u;
        if (!on) continue;
        out[order_index(pawn, lh, lv, a)] = a;
        n++;
    }
    return n;
}
}

// blocked_rows() (qz_path_rows.h) against the three-word blocked sets: number of mismatching (board, row, set) triples
extern "C" long hc_blocked_rows_mismatches(const uint64_t* hb, const uint64_t* vb, int n) {
    long bad = 0;
    for (int i = 0; i < n; i++) {
        Blk base = blk_or(blocked_from(spread8(hb[i]), spread8(vb[i])), blocked_borders());
        for (int r = 0; r < 9; r++) {
            uint32_t bn, bs, be, bw;
            blocked_rows(hb[i], vb[i], r, bn, bs, be, bw);
            bad += bn != bb_row(base.n, r);
            bad += bs != bb_row(base.s, r);
            bad += be != bb_row(base.e, r);
            bad += bw != bb_row(base.w, r);
        }
    }
    return bad;
}

// jump_plan_corner() / plan_jump_rows() (qz_path_rows.h: what k_wave_rules computes on parallel lanes) against
// make_jump_plan() / plan_jumps(): number of mismatches over both players of n boards
extern "C" long hc_plan_rows_mismatches(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n) {
    long bad = 0;
    for (int i = 0; i < n; i++) {
        Board b = unpack(hb[i], vb[i], meta[i]);
        if (winner_of(b) != 0) continue;
        for (int p = 1; p <= 2; p++) {
            const int O = side_opp(b, p);
            JumpPlan plan = make_jump_plan(b.hb, b.vb, O);
            int8_t cv[12];
            for (int k = 0; k < 12; k++) {
                int ref, val;
                jump_plan_corner(b.hb, b.vb, O, k, ref, val);
                bad += ref != plan.ref[k];
                bad += val != plan.val[k];
                cv[k] = (int8_t)val;
            }
            Jumps j = plan_jumps(plan, -1, false);
            for (int r = 0; r < 9; r++) {
                uint32_t jb[4], jd[4];
                plan_jump_rows(cv, O, r, jb, jd);
                for (int k = 0; k < 4; k++) {
                    const uint32_t eb = (j.a[k] >= 0 && j.a[k] <= 80 && j.a[k] / 9 == r) ? (1u << (j.a[k] % 9)) : 0u;
                    const uint32_t ed = j.a[k] >= 0 ? bb_row(j.d[k], r) : 0u;
                    bad += jb[k] != eb;
                    bad += jd[k] != ed;
                }
            }
        }
    }
    return bad;
}
extern "C" long hc_cut_row_mismatches() { return g_cut_row_mismatches; }
static long g_p2_mismatch = 0;
extern "C" long hc_p2_mismatches() { return g_p2_mismatch; }
// ---- the pooled kernel's phases (qz_movegen_pool.h), executed lane by lane for tiles of nb boards
static int g_try_detour = 0;
extern "C" void hc_set_detour(int on) { g_try_detour = on; }
static int g_finder = 0;  // 0: find_path_tables, 1: find_path_rows (the nine-rows formulation of k_wave_rules)
extern "C" void hc_set_finder(int f) { g_finder = f; }
static int g_handoff = 0;  // 1: launch 1 -> launch 2 through the 184-byte PoolHand record (what the device's pooled pipeline does)
extern "C" void hc_set_handoff(int on) { g_handoff = on; }
static long g_handoff_mismatch = 0;  // fields of the rebuilt PoolBoard / srcpos / suffix sets that differ from pool_k1's own
extern "C" long hc_handoff_mismatches() { return g_handoff_mismatch; }
static bool bb_same(const BB& a, const BB& b) { return a.w0 == b.w0 && a.w1 == b.w1 && a.w2 == b.w2; }
static void pool_tile(const Board* boards, int nb, uint32_t* mask5, float* planes, int64_t* floods, int64_t* flood_iters) {
    std::vector<PoolBoard> ctx(nb);
    std::vector<PathTab> tabs((size_t)nb * 2);
    std::vector<PoolHand> hand(nb);
    for (int i = 0; i < nb; i++)  // launch 1: lane = (board, player)
        for (int p = 2; p >= 1; p--) {
            if (g_finder == 0) pool_k1<0>(boards[i], false, true, p, ctx[i], tabs[(size_t)i * 2 + p - 1], g_try_detour);
            else pool_k1<1>(boards[i], false, true, p, ctx[i], tabs[(size_t)i * 2 + p - 1], g_try_detour);
        }
    if (g_handoff) {
        // the same boards through the hand-off record; what launch 2 rebuilds from it must be what pool_k1 left in its
        // PoolBoard / PathTab (compared field by field), and the rest of the tile then runs on the REBUILT records
        std::vector<PoolBoard> rb(nb);
        std::vector<uint8_t> sp((size_t)nb * 2 * 84, 255);
        for (int i = 0; i < nb; i++) {
            memset(&hand[i], 0xCD, sizeof(PoolHand));
            for (int p = 2; p >= 1; p--) pool_k1_hand(boards[i], false, p, hand[i], g_try_detour);
            pool_hand_rebuild_board(rb[i], hand[i], boards[i]);
            for (int p = 1; p <= 2; p++) pool_hand_rebuild_path(rb[i], p, hand[i].seq[p - 1], &sp[((size_t)i * 2 + p - 1) * 84]);
            const PoolBoard &a = ctx[i], &r = rb[i];
            long bad = 0;
            bad += a.flags != r.flags || a.pawn != r.pawn || a.sh != r.sh || a.sv != r.sv;
            for (int q = 0; q < 4; q++) bad += a.need[q] != r.need[q];
            bad += !bb_same(a.base.n, r.base.n) || !bb_same(a.base.s, r.base.s) || !bb_same(a.base.e, r.base.e) || !bb_same(a.base.w, r.base.w);
            if ((a.flags & 3u) == 1u)
                for (int p = 0; p < 2; p++) {
                    bad += a.len[p] != r.len[p] || a.pe[p].found != r.pe[p].found || a.pe[p].jump != r.pe[p].jump;
                    bad += !bb_same(a.pe[p].pn, r.pe[p].pn) || !bb_same(a.pe[p].ps, r.pe[p].ps) || !bb_same(a.pe[p].pe, r.pe[p].pe) || !bb_same(a.pe[p].pw, r.pe[p].pw);
                    bad += memcmp(&a.plan[p], &r.plan[p], sizeof(JumpPlan)) != 0;
                    if (a.len[p] > 0) {
                        bad += a.lastjump[p] != r.lastjump[p] || a.farjump[p] != r.farjump[p] || !bb_same(a.tiles[p], r.tiles[p]);
                        bad += memcmp(tabs[(size_t)i * 2 + p].srcpos, &sp[((size_t)i * 2 + p) * 84], 81) != 0;
                        for (int k = 0; k < a.len[p]; k++) bad += !bb_same(tabs[(size_t)i * 2 + p].suffix[k], pool_seq_suffix(hand[i].seq[p], k));
                    }
                }
            g_handoff_mismatch += bad;
        }
        std::vector<uint32_t> items;
        for (int i = 0; i < nb; i++)
            for (int ix = 0; ix < 64; ix++) {
                uint32_t m = pool_p2(rb[i], ix);
                if (m & 1u) items.push_back(pool_item(i, ix, true, 1));
                if (m & 2u) items.push_back(pool_item(i, ix, true, 2));
                if (m & 4u) items.push_back(pool_item(i, ix, false, 1));
                if (m & 8u) items.push_back(pool_item(i, ix, false, 2));
            }
        for (uint32_t it : items) {
            int bd = (int)(it >> 8), ix = (int)(it & 63u), p = (it & 0x80u) ? 2 : 1;
            bool hz = (it & 0x40u) != 0u;
            bool ok = pool_p3_seq(rb[bd], it, &sp[((size_t)bd * 2 + p - 1) * 84], hand[bd].seq[p - 1]);
            if (floods) (*floods)++;
            if (!ok) rb[bd].blocked[(hz ? 0 : 2) + (ix >> 5)] |= 1u << (ix & 31);
        }
        for (int i = 0; i < nb; i++) pool_p4(rb[i], mask5 + 5 * (long)i);
        (void)planes;
        return;
    }
    std::vector<uint32_t> items;
    for (int i = 0; i < nb; i++)
        for (int ix = 0; ix < 64; ix++) {
            uint32_t m = pool_p2(ctx[i], ix);
            if (!g_try_detour && m != pool_p2_ref(ctx[i], ix)) g_p2_mismatch++;  // need masks must equal the per-slot tests
            if (m & 1u) items.push_back(pool_item(i, ix, true, 1));
            if (m & 2u) items.push_back(pool_item(i, ix, true, 2));
            if (m & 4u) items.push_back(pool_item(i, ix, false, 1));
            if (m & 8u) items.push_back(pool_item(i, ix, false, 2));
        }
    for (uint32_t it : items) {
        int bd = (int)(it >> 8), ix = (int)(it & 63u), p = (it & 0x80u) ? 2 : 1;
        bool hz = (it & 0x40u) != 0u;
        bool ok = pool_p3(ctx[bd], it, tabs[(size_t)bd * 2 + p - 1]);
        if (floods) (*floods)++;
        if (!ok) ctx[bd].blocked[(hz ? 0 : 2) + (ix >> 5)] |= 1u << (ix & 31);
    }
    (void)flood_iters;
    for (int i = 0; i < nb; i++) pool_p4(ctx[i], mask5 + 5 * (long)i);
    if (planes) {  // P5 through the per-board bitmap, 16 bytes (4 floats) at a time like the kernel
        std::vector<uint32_t> bm((size_t)(nb + 1) * POOL_BM_WORDS, 0u);
        for (int i = 0; i < nb; i++) {
            EncCtx ec;
            enc_ctx_build(ec, boards[i], false);
            for (int k = 0; k < POOL_BM_WORDS; k++) bm[(size_t)i * POOL_BM_WORDS + k] = pool_bitmap_word(ec, k);
        }
        int nf = nb * 2106;
        std::vector<uint32_t> st((size_t)POOL_STREAM_WORDS(nb) + 1, 0u);
        for (int w = 0; w < POOL_STREAM_WORDS(nb); w++) st[w] = pool_stream_word(bm.data(), w);
        for (int q = 0; q < (nf >> 2); q++) {
            uint32_t nib = pool_stream_nibble(st.data(), q);
            for (int j = 0; j < 4; j++) planes[4 * q + j] = (float)((nib >> j) & 1u);
        }
        if (nf & 3) {
            EncCtx ec;
            enc_ctx_build(ec, boards[nb - 1], false);
            planes[nf - 2] = pool_plane_value(ec, 2104);
            planes[nf - 1] = pool_plane_value(ec, 2105);
        }
    }
}

extern "C" void hc_movegen_pool(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n, int nb_tile, uint32_t* mask5,
                                float* planes, int64_t* floods) {
    if (floods) *floods = 0;
    std::vector<Board> bs(nb_tile);
    for (int b0 = 0; b0 < n; b0 += nb_tile) {
        int nb = n - b0 < nb_tile ? n - b0 : nb_tile;
        for (int i = 0; i < nb; i++) bs[i] = unpack(hb[b0 + i], vb[b0 + i], meta[b0 + i]);
        pool_tile(bs.data(), nb, mask5 + 5 * (long)b0, planes ? planes + (long)b0 * 2106 : nullptr, floods, nullptr);
    }
}

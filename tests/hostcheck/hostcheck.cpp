// hostcheck.cpp -- TEST-ONLY g++ build of alphazero_quoridor_amd/csrc/qz_rules.h.
//
// Emulates, lane by lane, exactly what the HIP movegen kernel does with the shared
// per-lane functions, so the bitboard formulation (closed-form corners, linear blocked
// sets, path-cut pruning) can be compared with the oracle on a machine without a GPU.
// It is never loaded by the product (alphazero_quoridor_amd/ does not know it exists).
#include <cstdint>
#include <cstring>

#include "../../alphazero_quoridor_amd/csrc/qz_rules.h"

using namespace qz;

extern "C" {

int hc_corner(uint64_t hb, uint64_t vb, int t, int which) { return corner(hb, vb, t, which); }

uint32_t hc_pawn_actions(uint64_t hb, uint64_t vb, int loc, int opp, int player) {
    return pawn_actions(hb, vb, loc, opp, player);
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

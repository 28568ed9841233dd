// qz_rules.h -- Quoridor rules on 81-bit bitboards (device + host-check build).
//
// Everything here is per-lane / wave-uniform arithmetic with no memory traffic; the
// wave-level orchestration (ballots, LDS work lists, stores) lives in qz_kernels.hip.
// The same header is compiled by g++ into tests/hostcheck (QZ_HD empty) so that the
// bitboard formulation can be checked against the oracle without a GPU; that build is
// test-only and is never loaded by the product.
//
// Semantics restated from the reference (file:line = cryer/AlphaZero_Quoridor):
//   corner()            quoridor.py:356-418  _get_intersections, incl. the row-0 NE overwrite
//   build_moves()       quoridor.py:287-293  n/s/e/w wall tests for all 81 tiles at once
//   jump_dests()        quoridor.py:301-351  jump branches around the opponent pawn
//   pawn_actions()      quoridor.py:272-353  the mover's ordered pawn codes (as a 12-bit mask)
//   flood()             quoridor.py:479-528  _bfs_to_goal == directed reachability
//   apply_action()      quoridor.py:159-186, 217-269  step / rotate_players
//   plane_value()       quoridor.py:58-131   state()
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define QZ_HD __host__ __device__ __forceinline__
#else
#define QZ_HD inline
#endif

// "does any lane of my wavefront see x?" -- lets per-lane loops run with ONE wave-uniform exit
// instead of per-lane divergent breaks (which cost dozens of exec-mask scalar instructions per
// trip on CDNA).  On the host build a "wave" is the single lane being emulated.
#if defined(__HIP_DEVICE_COMPILE__)
#define QZ_WAVE_ANY(x) (__ballot((x)) != 0ull)
#else
#define QZ_WAVE_ANY(x) (x)
#endif

namespace qz {

// ----------------------------------------------------------------------------- boards
struct Board {
    uint64_t hb, vb;  // horizontal / vertical wall bits, bit ix = intersection ix
    int p1, p2;       // pawn tiles (may be off-board on terminal boards)
    int w1, w2;       // walls remaining
    int cur;          // 1 | 2
};

QZ_HD uint64_t pack_meta(const Board& b) {
    return (uint64_t)(uint8_t)(int8_t)b.p1 | ((uint64_t)(uint8_t)(int8_t)b.p2 << 8) |
           ((uint64_t)(uint8_t)b.w1 << 16) | ((uint64_t)(uint8_t)b.w2 << 24) |
           ((uint64_t)(uint8_t)b.cur << 32);
}
QZ_HD Board unpack(uint64_t hb, uint64_t vb, uint64_t meta) {
    Board b;
    b.hb = hb;
    b.vb = vb;
    b.p1 = (int)(int8_t)(meta & 0xFF);
    b.p2 = (int)(int8_t)((meta >> 8) & 0xFF);
    b.w1 = (int)((meta >> 16) & 0xFF);
    b.w2 = (int)((meta >> 24) & 0xFF);
    b.cur = (int)((meta >> 32) & 0xFF);
    return b;
}
QZ_HD Board opening() {  // quoridor.py:26-56
    Board b;
    b.hb = 0;
    b.vb = 0;
    b.p1 = 4;
    b.p2 = 76;
    b.w1 = 10;
    b.w2 = 10;
    b.cur = 1;
    return b;
}
// quoridor.py:193-202 (player 2 first)
QZ_HD int winner_of(const Board& b) { return b.p2 < 9 ? 2 : (b.p1 > 71 ? 1 : 0); }

QZ_HD int action_delta(int a) {  // quoridor.py:217-241
    // N S E W NN SS EE WW NE NW SE SW
    const int d[12] = {9, -9, 1, -1, 18, -18, 2, -2, 10, 8, -8, -10};
    return d[a];
}

// quoridor.py:159-186: move / place, then rotate unless the game is over. returns done.
QZ_HD bool apply_action(Board& b, int a) {
    if (a < 12) {
        if (b.cur == 1) b.p1 += action_delta(a);
        else b.p2 += action_delta(a);
    } else {
        int w = a - 12;
        if (w < 64) b.hb |= 1ull << w;
        else b.vb |= 1ull << (w - 64);
        if (b.cur == 1) b.w1 -= 1;
        else b.w2 -= 1;
    }
    if (winner_of(b) != 0) return true;  // quoridor.py:176-179: no rotation
    b.cur = 3 - b.cur;
    return false;
}

// ----------------------------------------------------------------------------- 81-bit sets
struct BB {
    uint32_t w0, w1, w2;  // tiles 0-31, 32-63, 64-80
};
QZ_HD BB bb_zero() { return BB{0u, 0u, 0u}; }
QZ_HD BB bb_or(BB a, BB b) { return BB{a.w0 | b.w0, a.w1 | b.w1, a.w2 | b.w2}; }
QZ_HD BB bb_and(BB a, BB b) { return BB{a.w0 & b.w0, a.w1 & b.w1, a.w2 & b.w2}; }
QZ_HD BB bb_andn(BB a, BB b) { return BB{a.w0 & ~b.w0, a.w1 & ~b.w1, a.w2 & ~b.w2}; }
QZ_HD BB bb_not(BB a) { return BB{~a.w0, ~a.w1, ~a.w2 & 0x1FFFFu}; }
QZ_HD bool bb_any(BB a) { return (a.w0 | a.w1 | a.w2) != 0u; }
QZ_HD bool bb_eq(BB a, BB b) { return ((a.w0 ^ b.w0) | (a.w1 ^ b.w1) | (a.w2 ^ b.w2)) == 0u; }
QZ_HD BB bb_bit(int t) {  // t in 0..80 (anything else -> empty)
    BB r = bb_zero();
    uint32_t m = 1u << (t & 31);
    int w = t >> 5;
    r.w0 = (w == 0) ? m : 0u;
    r.w1 = (w == 1) ? m : 0u;
    r.w2 = (w == 2 && t <= 80) ? m : 0u;
    return r;
}
QZ_HD bool bb_test(BB a, int t) {  // t in 0..80
    uint32_t w = (t < 32) ? a.w0 : ((t < 64) ? a.w1 : a.w2);
    return (w >> (t & 31)) & 1u;
}
template <int K>
QZ_HD BB bb_shl(BB a) {  // towards higher tiles; bits past tile 80 are dropped
    BB r;
    r.w0 = a.w0 << K;
    r.w1 = (a.w1 << K) | (a.w0 >> (32 - K));
    r.w2 = ((a.w2 << K) | (a.w1 >> (32 - K))) & 0x1FFFFu;
    return r;
}
template <int K>
QZ_HD BB bb_shr(BB a) {
    BB r;
    r.w0 = (a.w0 >> K) | (a.w1 << (32 - K));
    r.w1 = (a.w1 >> K) | (a.w2 << (32 - K));
    r.w2 = a.w2 >> K;
    return r;
}
QZ_HD int bb_lowest(BB a) {  // index of lowest set bit; a must be non-empty
#if defined(__HIP_DEVICE_COMPILE__)
    if (a.w0) return __ffs((int)a.w0) - 1;
    if (a.w1) return 32 + __ffs((int)a.w1) - 1;
    return 64 + __ffs((int)a.w2) - 1;
#else
    if (a.w0) return __builtin_ctz(a.w0);
    if (a.w1) return 32 + __builtin_ctz(a.w1);
    return 64 + __builtin_ctz(a.w2);
#endif
}

// constant sets
QZ_HD BB ROW0() { return BB{0x1FFu, 0u, 0u}; }                      // tiles 0..8
QZ_HD BB ROW8() { return BB{0u, 0u, 0x1FF00u}; }                    // tiles 72..80
QZ_HD BB ROW0_C1_7() { return BB{0xFEu, 0u, 0u}; }                  // tiles 1..7
QZ_HD BB COL0() { return BB{0x08040201u, 0x80402010u, 0x00000100u}; }  // tiles 0,9,..,72
QZ_HD BB COL8() { return BB{0x04020100u, 0x40201008u, 0x00010080u}; }  // tiles 8,17,..,80

// 8x8 intersection bits -> 81-bit set with bit (9r+c) = I(r,c), r,c < 8
QZ_HD BB spread8(uint64_t x) {
    BB b = bb_zero();
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 8; r++) {
        uint32_t row = (uint32_t)(x >> (8 * r)) & 0xFFu;
        int pos = 9 * r, w = pos >> 5, off = pos & 31;
        uint32_t lo = row << off;
        uint32_t hi = (off > 24) ? (row >> (32 - off)) : 0u;
        if (w == 0) {
            b.w0 |= lo;
            b.w1 |= hi;
        } else if (w == 1) {
            b.w1 |= lo;
            b.w2 |= hi;
        } else {
            b.w2 |= lo;
        }
    }
    return b;
}

// ----------------------------------------------------------------------------- corners
// quoridor.py:356-418 in closed form. which: 0 NW, 1 NE, 2 SE, 3 SW.
QZ_HD int inter_val(uint64_t hb, uint64_t vb, int r, int c) {
    int ix = 8 * r + c;
    return (int)((hb >> ix) & 1ull) - (int)((vb >> ix) & 1ull);
}
QZ_HD int corner(uint64_t hb, uint64_t vb, int t, int which) {
    int r = t / 9, c = t - 9 * r;
    switch (which) {
        case 0:  // NW
            if (c == 0) return -1;
            if (r == 8) return 1;
            return inter_val(hb, vb, r, c - 1);
        case 1:  // NE
            if (r == 8) return 1;
            if (r == 0) return c == 0 ? inter_val(hb, vb, 0, 0) : inter_val(hb, vb, 0, c - 1);  // :388,:392
            if (c == 8) return -1;
            return inter_val(hb, vb, r, c);
        case 2:  // SE
            if (c == 8) return -1;
            if (r == 0) return 1;
            return inter_val(hb, vb, r - 1, c);
        default:  // SW
            if (r == 0) return 1;
            if (c == 0) return -1;
            return inter_val(hb, vb, r - 1, c - 1);
    }
}

// corner_ref() for all 81 tiles x 4 corners as a constant table (generated from corner_ref();
// tests/hostcheck verifies the two agree).  Index 4*t + which.
#if defined(__HIPCC__)
#define QZ_TABLE_QUAL __device__ __constant__
#else
#define QZ_TABLE_QUAL static const
#endif
#if defined(__HIPCC__)
QZ_TABLE_QUAL __attribute__((aligned(4))) uint8_t g_corner_ref_dev[324] = {
    65, 0, 64, 64, 0, 0, 64, 64, 1, 1, 64, 64, 2, 2, 64, 64, 3, 3, 64, 64, 4, 4, 64, 64,
    5, 5, 64, 64, 6, 6, 64, 64, 7, 7, 65, 64, 65, 8, 0, 65, 8, 9, 1, 0, 9, 10, 2, 1,
    10, 11, 3, 2, 11, 12, 4, 3, 12, 13, 5, 4, 13, 14, 6, 5, 14, 15, 7, 6, 15, 65, 65, 7,
    65, 16, 8, 65, 16, 17, 9, 8, 17, 18, 10, 9, 18, 19, 11, 10, 19, 20, 12, 11, 20, 21, 13, 12,
    21, 22, 14, 13, 22, 23, 15, 14, 23, 65, 65, 15, 65, 24, 16, 65, 24, 25, 17, 16, 25, 26, 18, 17,
    26, 27, 19, 18, 27, 28, 20, 19, 28, 29, 21, 20, 29, 30, 22, 21, 30, 31, 23, 22, 31, 65, 65, 23,
    65, 32, 24, 65, 32, 33, 25, 24, 33, 34, 26, 25, 34, 35, 27, 26, 35, 36, 28, 27, 36, 37, 29, 28,
    37, 38, 30, 29, 38, 39, 31, 30, 39, 65, 65, 31, 65, 40, 32, 65, 40, 41, 33, 32, 41, 42, 34, 33,
    42, 43, 35, 34, 43, 44, 36, 35, 44, 45, 37, 36, 45, 46, 38, 37, 46, 47, 39, 38, 47, 65, 65, 39,
    65, 48, 40, 65, 48, 49, 41, 40, 49, 50, 42, 41, 50, 51, 43, 42, 51, 52, 44, 43, 52, 53, 45, 44,
    53, 54, 46, 45, 54, 55, 47, 46, 55, 65, 65, 47, 65, 56, 48, 65, 56, 57, 49, 48, 57, 58, 50, 49,
    58, 59, 51, 50, 59, 60, 52, 51, 60, 61, 53, 52, 61, 62, 54, 53, 62, 63, 55, 54, 63, 65, 65, 55,
    65, 64, 56, 65, 64, 64, 57, 56, 64, 64, 58, 57, 64, 64, 59, 58, 64, 64, 60, 59, 64, 64, 61, 60,
    64, 64, 62, 61, 64, 64, 63, 62, 64, 64, 65, 63,
};
#endif
static const uint8_t g_corner_ref_host[324] = {
    65, 0, 64, 64, 0, 0, 64, 64, 1, 1, 64, 64, 2, 2, 64, 64, 3, 3, 64, 64, 4, 4, 64, 64,
    5, 5, 64, 64, 6, 6, 64, 64, 7, 7, 65, 64, 65, 8, 0, 65, 8, 9, 1, 0, 9, 10, 2, 1,
    10, 11, 3, 2, 11, 12, 4, 3, 12, 13, 5, 4, 13, 14, 6, 5, 14, 15, 7, 6, 15, 65, 65, 7,
    65, 16, 8, 65, 16, 17, 9, 8, 17, 18, 10, 9, 18, 19, 11, 10, 19, 20, 12, 11, 20, 21, 13, 12,
    21, 22, 14, 13, 22, 23, 15, 14, 23, 65, 65, 15, 65, 24, 16, 65, 24, 25, 17, 16, 25, 26, 18, 17,
    26, 27, 19, 18, 27, 28, 20, 19, 28, 29, 21, 20, 29, 30, 22, 21, 30, 31, 23, 22, 31, 65, 65, 23,
    65, 32, 24, 65, 32, 33, 25, 24, 33, 34, 26, 25, 34, 35, 27, 26, 35, 36, 28, 27, 36, 37, 29, 28,
    37, 38, 30, 29, 38, 39, 31, 30, 39, 65, 65, 31, 65, 40, 32, 65, 40, 41, 33, 32, 41, 42, 34, 33,
    42, 43, 35, 34, 43, 44, 36, 35, 44, 45, 37, 36, 45, 46, 38, 37, 46, 47, 39, 38, 47, 65, 65, 39,
    65, 48, 40, 65, 48, 49, 41, 40, 49, 50, 42, 41, 50, 51, 43, 42, 51, 52, 44, 43, 52, 53, 45, 44,
    53, 54, 46, 45, 54, 55, 47, 46, 55, 65, 65, 47, 65, 56, 48, 65, 56, 57, 49, 48, 57, 58, 50, 49,
    58, 59, 51, 50, 59, 60, 52, 51, 60, 61, 53, 52, 61, 62, 54, 53, 62, 63, 55, 54, 63, 65, 65, 55,
    65, 64, 56, 65, 64, 64, 57, 56, 64, 64, 58, 57, 64, 64, 59, 58, 64, 64, 60, 59, 64, 64, 61, 60,
    64, 64, 62, 61, 64, 64, 63, 62, 64, 64, 65, 63,
};
QZ_HD int corner_ref_tab(int t, int which) {
#if defined(__HIP_DEVICE_COMPILE__)
    return g_corner_ref_dev[4 * t + which];
#else
    return g_corner_ref_host[4 * t + which];
#endif
}
// the four corner references of tile t in one word: NW | NE << 8 | SE << 16 | SW << 24 (t in 0..80).  ONE load where
// four byte loads used to wait for each other: on the GPU every table look-up is a round trip of the dependent chain
QZ_HD uint32_t corner_ref4(int t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return reinterpret_cast<const uint32_t*>(g_corner_ref_dev)[t];
#else
    return (uint32_t)g_corner_ref_host[4 * t] | ((uint32_t)g_corner_ref_host[4 * t + 1] << 8) |
           ((uint32_t)g_corner_ref_host[4 * t + 2] << 16) | ((uint32_t)g_corner_ref_host[4 * t + 3] << 24);
#endif
}
// branch-free ref_value
QZ_HD int ref_value_fast(uint64_t hb, uint64_t vb, int ref) {
    int sh = ref & 63;
    int h = (int)((hb >> sh) & 1ull), v = (int)((vb >> sh) & 1ull);
    return ref < 64 ? h - v : (ref == 64 ? 1 : -1);
}
QZ_HD int corner_tab(uint64_t hb, uint64_t vb, int t, int which) { return ref_value_fast(hb, vb, corner_ref_tab(t, which)); }

// ----------------------------------------------------------------------------- move sets
// blocked-by-wall sets for the four simple moves, for all tiles at once.  Linear (OR of
// shifts) in the spread wall sets, so a candidate wall contributes an additive delta.
struct Blk {
    BB n, s, e, w;
};
QZ_HD Blk blocked_from(BB Hs, BB Vs) {
    Blk k;
    BB h1 = bb_shl<1>(Hs);
    // N: rows 1..7 use I(r,c-1)|I(r,c); row 0 uses I(0,c-1) only, plus I(0,0) for tile 0
    k.n = bb_or(bb_andn(bb_or(Hs, h1), ROW0()), bb_and(bb_or(h1, bb_and(Hs, BB{1u, 0u, 0u})), ROW0()));
    // S: I(r-1,c) | I(r-1,c-1)
    k.s = bb_or(bb_shl<9>(Hs), bb_shl<10>(Hs));
    // E: I(r,c) | I(r-1,c); row 0, cols 1..7 use I(0,c-1) instead (the :392 overwrite)
    BB v1 = bb_shl<1>(Vs);
    k.e = bb_or(bb_andn(bb_or(Vs, bb_shl<9>(Vs)), ROW0_C1_7()), bb_and(v1, ROW0_C1_7()));
    // W: I(r,c-1) | I(r-1,c-1)
    k.w = bb_or(v1, bb_shl<10>(Vs));
    return k;
}
// border sentinels: N from row 8, S from row 0, E from col 8, W from col 0 never pass the
// wall test (quoridor.py:365-392 sentinels +1/-1)
QZ_HD Blk blocked_borders() {
    Blk k;
    k.n = ROW8();
    k.s = ROW0();
    k.e = COL8();
    k.w = COL0();
    return k;
}
QZ_HD Blk blk_or(Blk a, Blk b) { return Blk{bb_or(a.n, b.n), bb_or(a.s, b.s), bb_or(a.e, b.e), bb_or(a.w, b.w)}; }

struct Jumps {
    int a[4];   // tiles adjacent to the opponent: O-9, O+9, O-1, O+1 (or -1 when unusable)
    BB d[4];    // on-board jump destinations from a[k]
};
QZ_HD BB dest_bit(int t) { return (t >= 0 && t <= 80) ? bb_bit(t) : bb_zero(); }
// quoridor.py:301-351 for the four tiles next to the opponent O (walls hb/vb).
QZ_HD Jumps jump_dests(uint64_t hb, uint64_t vb, int O) {
    const int H = 1, V = -1;
    Jumps j;
    int onw = corner(hb, vb, O, 0), one = corner(hb, vb, O, 1), ose = corner(hb, vb, O, 2),
        osw = corner(hb, vb, O, 3);
    for (int k = 0; k < 4; k++) {
        j.a[k] = -1;
        j.d[k] = bb_zero();
    }
    {  // opponent north of A = O-9   (:301-314)
        int A = O - 9;
        if (A >= 0) {
            int xnw = corner(hb, vb, A, 0), xne = corner(hb, vb, A, 1);
            if (xne != H && xnw != H) {
                j.a[0] = A;
                BB d = bb_zero();
                if (onw != H && one != H) d = bb_or(d, dest_bit(O + 9));
                if (one != V && xne != V) d = bb_or(d, dest_bit(A + 10));
                if (onw != V && xnw != V) d = bb_or(d, dest_bit(A + 8));
                j.d[0] = d;
            }
        }
    }
    {  // opponent south of A = O+9   (:317-327)
        int A = O + 9;
        if (A <= 80) {
            int xse = corner(hb, vb, A, 2), xsw = corner(hb, vb, A, 3);
            if (xse != H && xsw != H) {
                j.a[1] = A;
                BB d = bb_zero();
                if (osw != H && ose != H) d = bb_or(d, dest_bit(O - 9));
                if (ose != V && xse != V) d = bb_or(d, dest_bit(A - 8));
                if (osw != V && xsw != V) d = bb_or(d, dest_bit(A - 10));
                j.d[1] = d;
            }
        }
    }
    {  // opponent east of A = O-1    (:330-339)
        int A = O - 1;
        if (A >= 0) {
            int xse = corner(hb, vb, A, 2), xne = corner(hb, vb, A, 1);
            if (xse != V && xne != V) {
                j.a[2] = A;
                BB d = bb_zero();
                if (ose != V && one != V) d = bb_or(d, dest_bit(O + 1));
                if (one != H) d = bb_or(d, dest_bit(A + 10));
                if (ose != H) d = bb_or(d, dest_bit(A - 8));
                j.d[2] = d;
            }
        }
    }
    {  // opponent west of A = O+1    (:342-351)
        int A = O + 1;
        if (A <= 80) {
            int xsw = corner(hb, vb, A, 3), xnw = corner(hb, vb, A, 0);
            if (xsw != V && xnw != V) {
                j.a[3] = A;
                BB d = bb_zero();
                if (onw != V && osw != V) d = bb_or(d, dest_bit(O - 1));
                if (onw != H) d = bb_or(d, dest_bit(A + 8));
                if (osw != H) d = bb_or(d, dest_bit(A - 10));
                j.d[3] = d;
            }
        }
    }
    return j;
}

// quoridor.py:272-353 for the side to move, as a 12-bit mask (bit = action code).  The
// reference's emission order is ascending code order in every branch.
QZ_HD uint32_t pawn_actions(uint64_t hb, uint64_t vb, int loc, int opp, int player) {
    const int H = 1, V = -1;
    int xnw = corner(hb, vb, loc, 0), xne = corner(hb, vb, loc, 1), xse = corner(hb, vb, loc, 2),
        xsw = corner(hb, vb, loc, 3);
    bool on = loc == opp - 9, os = loc == opp + 9, oe = loc == opp - 1, ow = loc == opp + 1;
    int row = loc / 9;
    uint32_t m = 0;
    bool n = xnw != H && xne != H && !on;
    bool s = xsw != H && xse != H && !os;
    bool e = xne != V && xse != V && !oe;
    bool w = xnw != V && xsw != V && !ow;
    if (n || (player == 1 && row == 8)) m |= 1u << 0;
    if (s || (player == 2 && row == 0)) m |= 1u << 1;
    if (e) m |= 1u << 2;
    if (w) m |= 1u << 3;
    if (on && xne != H && xnw != H) {
        int onw = corner(hb, vb, opp, 0), one = corner(hb, vb, opp, 1);
        if ((onw != H && one != H) || (row == 7 && player == 1)) m |= 1u << 4;
        if (one != V && xne != V) m |= 1u << 8;
        if (onw != V && xnw != V) m |= 1u << 9;
    } else if (os && xse != H && xsw != H) {
        int ose = corner(hb, vb, opp, 2), osw = corner(hb, vb, opp, 3);
        if ((osw != H && ose != H) || (row == 1 && player == 2)) m |= 1u << 5;
        if (ose != V && xse != V) m |= 1u << 10;
        if (osw != V && xsw != V) m |= 1u << 11;
    } else if (oe && xse != V && xne != V) {
        int one = corner(hb, vb, opp, 1), ose = corner(hb, vb, opp, 2);
        if (ose != V && one != V) m |= 1u << 6;
        if (one != H) m |= 1u << 8;
        if (ose != H) m |= 1u << 10;
    } else if (ow && xsw != V && xnw != V) {
        int onw = corner(hb, vb, opp, 0), osw = corner(hb, vb, opp, 3);
        if (onw != V && osw != V) m |= 1u << 7;
        if (onw != H) m |= 1u << 9;
        if (osw != H) m |= 1u << 11;
    }
    return m;
}

QZ_HD uint32_t pawn_actions_tab(uint64_t hb, uint64_t vb, int loc, int opp, int player) {
    const int H = 1, V = -1;
    // both tiles' corner references up front: two independent loads, one round trip
    const uint32_t wl = corner_ref4((unsigned)loc <= 80u ? loc : 0), wo = corner_ref4((unsigned)opp <= 80u ? opp : 0);
    int xnw = ref_value_fast(hb, vb, (int)(wl & 0xFFu)), xne = ref_value_fast(hb, vb, (int)((wl >> 8) & 0xFFu)),
        xse = ref_value_fast(hb, vb, (int)((wl >> 16) & 0xFFu)), xsw = ref_value_fast(hb, vb, (int)(wl >> 24));
    const int onw = ref_value_fast(hb, vb, (int)(wo & 0xFFu)), one = ref_value_fast(hb, vb, (int)((wo >> 8) & 0xFFu)),
              ose = ref_value_fast(hb, vb, (int)((wo >> 16) & 0xFFu)), osw = ref_value_fast(hb, vb, (int)(wo >> 24));
    bool on = loc == opp - 9, os = loc == opp + 9, oe = loc == opp - 1, ow = loc == opp + 1;
    int row = loc / 9;
    uint32_t m = 0;
    bool n = xnw != H && xne != H && !on;
    bool s = xsw != H && xse != H && !os;
    bool e = xne != V && xse != V && !oe;
    bool w = xnw != V && xsw != V && !ow;
    if (n || (player == 1 && row == 8)) m |= 1u << 0;
    if (s || (player == 2 && row == 0)) m |= 1u << 1;
    if (e) m |= 1u << 2;
    if (w) m |= 1u << 3;
    if (on && xne != H && xnw != H) {
        if ((onw != H && one != H) || (row == 7 && player == 1)) m |= 1u << 4;
        if (one != V && xne != V) m |= 1u << 8;
        if (onw != V && xnw != V) m |= 1u << 9;
    } else if (os && xse != H && xsw != H) {
        if ((osw != H && ose != H) || (row == 1 && player == 2)) m |= 1u << 5;
        if (ose != V && xse != V) m |= 1u << 10;
        if (osw != V && xsw != V) m |= 1u << 11;
    } else if (oe && xse != V && xne != V) {
        if (ose != V && one != V) m |= 1u << 6;
        if (one != H) m |= 1u << 8;
        if (ose != H) m |= 1u << 10;
    } else if (ow && xsw != V && xnw != V) {
        if (onw != V && osw != V) m |= 1u << 7;
        if (onw != H) m |= 1u << 9;
        if (osw != H) m |= 1u << 11;
    }
    return m;
}

// ----------------------------------------------------------------------------- reachability
struct Graph {
    BB cn, cs, ce, cw;  // tiles from which N/S/E/W passes the wall test
    BB notO;            // every tile except the opponent's (simple moves may not enter it)
    Jumps j;
};
QZ_HD Graph make_graph(Blk blocked, uint64_t hb, uint64_t vb, int O) {
    Graph g;
    g.cn = bb_not(blocked.n);
    g.cs = bb_not(blocked.s);
    g.ce = bb_not(blocked.e);
    g.cw = bb_not(blocked.w);
    g.notO = bb_not(dest_bit(O));
    g.j = jump_dests(hb, vb, O);
    return g;
}
// one BFS layer: every tile generated from the set R (quoridor.py:488-516)
QZ_HD BB expand(const Graph& g, BB R) {
    BB nx = bb_shl<9>(bb_and(R, g.cn));
    nx = bb_or(nx, bb_shr<9>(bb_and(R, g.cs)));
    nx = bb_or(nx, bb_shl<1>(bb_and(R, g.ce)));
    nx = bb_or(nx, bb_shr<1>(bb_and(R, g.cw)));
    nx = bb_and(nx, g.notO);
    for (int k = 0; k < 4; k++) {
        if (g.j.a[k] >= 0 && bb_test(R, g.j.a[k])) nx = bb_or(nx, g.j.d[k]);
    }
    return nx;
}
// same, with the four jump sources pre-merged into `jsrc` (tested once per layer)
QZ_HD BB expand_j(const Graph& g, BB jsrc, BB R) {
    BB nx = bb_shl<9>(bb_and(R, g.cn));
    nx = bb_or(nx, bb_shr<9>(bb_and(R, g.cs)));
    nx = bb_or(nx, bb_shl<1>(bb_and(R, g.ce)));
    nx = bb_or(nx, bb_shr<1>(bb_and(R, g.cw)));
    nx = bb_and(nx, g.notO);
    if (bb_any(bb_and(R, jsrc))) {
        for (int k = 0; k < 4; k++)
            if (g.j.a[k] >= 0 && bb_test(R, g.j.a[k])) nx = bb_or(nx, g.j.d[k]);
    }
    return nx;
}
QZ_HD BB jump_sources(const Graph& g) {
    BB jsrc = bb_zero();
    for (int k = 0; k < 4; k++)
        if (g.j.a[k] >= 0) jsrc = bb_or(jsrc, bb_bit(g.j.a[k]));
    return jsrc;
}
// quoridor.py:479-528: can `start` reach any tile of `goal`?  (goal test on generation;
// FIFO order is unobservable, so a layered flood gives the same answer)
QZ_HD bool flood(const Graph& g, int start, BB goal) {
    BB R = bb_bit(start);
    for (int it = 0; it < 96; it++) {
        BB nx = expand(g, R);
        if (bb_any(bb_and(nx, goal))) return true;
        BB R2 = bb_or(R, nx);
        if (bb_eq(R2, R)) return false;
        R = R2;
    }
    return false;
}

// edges of one concrete start->goal path, grouped by move type (tiles the move leaves from)
struct PathEdges {
    BB pn, ps, pe, pw;
    bool jump;   // the path uses at least one jump edge
    bool found;
};
// Layered flood that keeps every layer in `layers` (caller storage, >= 82 entries), then
// walks one path back from the goal.  Used once per player per board; candidates that
// remove none of these edges cannot disconnect the player (adding a wall only removes
// edges: SURVEY Appendix A.5).
QZ_HD PathEdges find_path(const Graph& g, int start, BB goal, BB* layers) {
    PathEdges p;
    p.pn = p.ps = p.pe = p.pw = bb_zero();
    p.jump = false;
    p.found = false;
    BB R = bb_bit(start);
    layers[0] = R;
    int L = 0;
    BB hit = bb_zero();
    for (int it = 0; it < 81; it++) {
        BB nx = expand(g, R);
        hit = bb_and(nx, goal);
        BB R2 = bb_or(R, nx);
        if (bb_any(hit)) {
            L = it + 1;
            layers[L] = R2;
            p.found = true;
            break;
        }
        if (bb_eq(R2, R)) return p;
        R = R2;
        layers[it + 1] = R;
    }
    if (!p.found) return p;
    int t = bb_lowest(hit);
    for (int i = L; i >= 1; i--) {
        BB prev = layers[i - 1];
        if (bb_test(prev, t)) continue;  // already reached earlier: same tile, shallower layer
        int s;
        if (t >= 9 && bb_test(prev, t - 9) && bb_test(g.cn, t - 9) && bb_test(g.notO, t)) {
            s = t - 9;
            p.pn = bb_or(p.pn, bb_bit(s));
        } else if (t <= 71 && bb_test(prev, t + 9) && bb_test(g.cs, t + 9) && bb_test(g.notO, t)) {
            s = t + 9;
            p.ps = bb_or(p.ps, bb_bit(s));
        } else if (t >= 1 && bb_test(prev, t - 1) && bb_test(g.ce, t - 1) && bb_test(g.notO, t)) {
            s = t - 1;
            p.pe = bb_or(p.pe, bb_bit(s));
        } else if (t <= 79 && bb_test(prev, t + 1) && bb_test(g.cw, t + 1) && bb_test(g.notO, t)) {
            s = t + 1;
            p.pw = bb_or(p.pw, bb_bit(s));
        } else {
            s = -1;
            for (int k = 0; k < 4; k++) {
                if (s < 0 && g.j.a[k] >= 0 && bb_test(prev, g.j.a[k]) && bb_test(g.j.d[k], t)) s = g.j.a[k];
            }
            p.jump = true;
            if (s < 0) {  // cannot happen; fail safe = re-check every candidate
                p.pn = p.ps = p.pe = p.pw = bb_not(bb_zero());
                return p;
            }
        }
        t = s;
    }
    return p;
}

// ----------------------------------------------------------------------------- candidates
// static part of _validate_horizontal/_vertical (quoridor.py:432-444, 448-459): slot empty
// and no same-orientation neighbour overlap, for all 64 slots at once
QZ_HD uint64_t static_ok_h(uint64_t hb, uint64_t vb) {
    const uint64_t C0 = 0x0101010101010101ull, C7 = 0x8080808080808080ull;
    return ~(hb | vb) & ~((hb << 1) & ~C0) & ~((hb >> 1) & ~C7);
}
QZ_HD uint64_t static_ok_v(uint64_t hb, uint64_t vb) { return ~(hb | vb) & ~(vb << 8) & ~(vb >> 8); }

// what a candidate wall at intersection ix adds to the blocked sets
QZ_HD Blk candidate_delta(int ix, bool horizontal) {
    int b = 9 * (ix >> 3) + (ix & 7);
    BB bit = bb_bit(b);
    return horizontal ? blocked_from(bit, bb_zero()) : blocked_from(bb_zero(), bit);
}
QZ_HD bool cuts(const Blk& d, const PathEdges& p) {
    BB x = bb_or(bb_or(bb_and(d.n, p.pn), bb_and(d.s, p.ps)), bb_or(bb_and(d.e, p.pe), bb_and(d.w, p.pw)));
    return bb_any(x);
}
// intersections whose value is read by the jump logic around opponent tile O
QZ_HD bool near_opp(int ix, int O) {
    int ir = ix >> 3, ic = ix & 7, r = O / 9, c = O - 9 * r;
    return ir >= r - 2 && ir <= r + 1 && ic >= c - 2 && ic <= c + 1;
}

// 81-bit tile set -> 64-bit slot set: bit (8r+c) <- tile (r,c) for r,c < 8 (row 8 / column 8 dropped)
QZ_HD uint64_t compress8(BB t) {
    uint64_t out = 0;
    const uint32_t w[4] = {t.w0, t.w1, t.w2, 0u};
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 8; r++) {
        int pos = 9 * r, i = pos >> 5, off = pos & 31;
        uint32_t v = w[i] >> off;
        if (off > 24) v |= w[i + 1] << (32 - off);
        out |= (uint64_t)(v & 0xFFu) << (8 * r);
    }
    return out;
}
// All wall slots whose wall removes at least one edge of a path, per orientation: the inverse
// image of candidate_delta_fast() over the path's edge sets (== cuts() for all 128 candidates).
struct CutMasks {
    uint64_t h, v;
};
QZ_HD CutMasks path_cut_masks(const PathEdges& e) {
    const uint64_t ROW0 = 0xFFull;
    CutMasks m;
    uint64_t a = compress8(e.pn), b = compress8(bb_shr<1>(e.pn));
    uint64_t cn = ((a | b) & ~ROW0) | ((b | (a & 1ull)) & ROW0);
    uint64_t cs = compress8(bb_shr<9>(e.ps)) | compress8(bb_shr<10>(e.ps));
    m.h = cn | cs;
    uint64_t e0 = compress8(e.pe), e1 = compress8(bb_shr<1>(e.pe)), e9 = compress8(bb_shr<9>(e.pe));
    uint64_t ce = ((e0 | e9) & ~ROW0) | ((e9 | (e1 & 0x7Full) | (e0 & 1ull)) & ROW0);
    uint64_t cw = compress8(bb_shr<1>(e.pw)) | compress8(bb_shr<10>(e.pw));
    m.v = ce | cw;
    return m;
}
// near_opp() for all 64 slots at once
QZ_HD uint64_t near_opp_mask(int O) {
    int r = O / 9, c = O - 9 * r;
    int c0 = c - 2 < 0 ? 0 : c - 2, c1 = c + 1 > 7 ? 7 : c + 1;
    int r0 = r - 2 < 0 ? 0 : r - 2, r1 = r + 1 > 7 ? 7 : r + 1;
    if (c0 > c1 || r0 > r1) return 0ull;
    uint64_t cols = ((1ull << (c1 - c0 + 1)) - 1ull) << c0;
    uint64_t m = 0;
    for (int rr = r0; rr <= r1; rr++) m |= cols << (8 * rr);
    return m;
}

// position of an action inside the reference's ordered actions() list, given the legal
// sets (pawn12, legalH, legalV): quoridor.py:146-157 + 423-428 (pawn codes ascending, then
// walls interleaved h(ix), v(ix))
QZ_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
QZ_HD int order_index(uint32_t pawn12, uint64_t lh, uint64_t lv, int a) {
    if (a < 12) return popc64(pawn12 & ((1u << a) - 1u));
    int np = popc64(pawn12);
    if (a < 76) {
        int ix = a - 12;
        uint64_t below = (ix == 0) ? 0ull : (~0ull >> (64 - ix));
        return np + popc64(lh & below) + popc64(lv & below);
    }
    int ix = a - 76;
    uint64_t below = (ix == 0) ? 0ull : (~0ull >> (64 - ix));
    uint64_t beq = below | (1ull << ix);
    return np + popc64(lh & beq) + popc64(lv & below);
}

// ----------------------------------------------------------------------------- per-board context
// Wave-uniform data of one actions() call (quoridor.py:138-157).
struct MoveCtx {
    Board b;
    Blk base;         // blocked sets of the walls on the board (borders included)
    uint64_t sh, sv;  // slots passing the static tests of _validate_horizontal/_vertical
    uint32_t pawn;    // the mover's pawn codes
    bool walls;       // mover has walls left (quoridor.py:149-150)
};
QZ_HD MoveCtx make_ctx(const Board& b) {
    MoveCtx c;
    c.b = b;
    c.base = blk_or(blocked_from(spread8(b.hb), spread8(b.vb)), blocked_borders());
    c.sh = static_ok_h(b.hb, b.vb);
    c.sv = static_ok_v(b.hb, b.vb);
    int loc = b.cur == 1 ? b.p1 : b.p2, opp = b.cur == 1 ? b.p2 : b.p1;
    c.pawn = pawn_actions(b.hb, b.vb, loc, opp, b.cur);
    c.walls = (b.cur == 1 ? b.w1 : b.w2) > 0;
    return c;
}
// player p's search problem in _blocks_path (quoridor.py:463-477): P1 -> row 8, P2 -> row 0,
// the other pawn fixed as obstacle / jump pivot
QZ_HD int side_start(const Board& b, int p) { return p == 1 ? b.p1 : b.p2; }
QZ_HD int side_opp(const Board& b, int p) { return p == 1 ? b.p2 : b.p1; }
QZ_HD BB side_goal(int p) { return p == 1 ? ROW8() : ROW0(); }

// one concrete path of player p on the current walls (no candidate)
QZ_HD PathEdges base_path(const MoveCtx& c, int p, BB* layers) {
    Graph g = make_graph(c.base, c.b.hb, c.b.vb, side_opp(c.b, p));
    return find_path(g, side_start(c.b, p), side_goal(p), layers);
}
// must player p's reachability be re-checked for this candidate?
QZ_HD bool needs_check(const MoveCtx& c, const PathEdges& path, int p, int ix, const Blk& delta) {
    if (!path.found) return false;  // p is already cut off: every candidate "blocks", no flood needed
    return cuts(delta, path) || (path.jump && near_opp(ix, side_opp(c.b, p)));
}
// _bfs_to_goal for player p with the candidate wall added (quoridor.py:470-475)
QZ_HD bool candidate_reaches(const MoveCtx& c, int p, int ix, bool horizontal, const Blk& delta) {
    uint64_t hb = c.b.hb | (horizontal ? (1ull << ix) : 0ull);
    uint64_t vb = c.b.vb | (horizontal ? 0ull : (1ull << ix));
    Graph g = make_graph(blk_or(c.base, delta), hb, vb, side_opp(c.b, p));
    return flood(g, side_start(c.b, p), side_goal(p));
}

// ----------------------------------------------------------------------------- v2 building blocks
// (used by the pooled kernel: lane = board / (board,player) / (board,slot) / work item)

// A corner of a tile as a REFERENCE: 0..63 = intersection index, 64 = constant +1, 65 = constant -1
// (same case analysis as corner(); quoridor.py:356-418)
QZ_HD int corner_ref(int t, int which) {
    int r = t / 9, c = t - 9 * r;
    switch (which) {
        case 0:
            if (c == 0) return 65;
            if (r == 8) return 64;
            return 8 * r + c - 1;
        case 1:
            if (r == 8) return 64;
            if (r == 0) return c == 0 ? 0 : c - 1;
            if (c == 8) return 65;
            return 8 * r + c;
        case 2:
            if (c == 8) return 65;
            if (r == 0) return 64;
            return 8 * (r - 1) + c;
        default:
            if (r == 0) return 64;
            if (c == 0) return 65;
            return 8 * (r - 1) + c - 1;
    }
}
QZ_HD int ref_value(uint64_t hb, uint64_t vb, int ref) {
    if (ref >= 64) return ref == 64 ? 1 : -1;
    return (int)((hb >> ref) & 1ull) - (int)((vb >> ref) & 1ull);
}
// Everything about the jumps around opponent tile O that does not depend on a candidate wall:
// the 12 corners the jump rules read (as references + their values on the current walls).
//   0..3  O.NW O.NE O.SE O.SW | 4,5 A0.NW A0.NE | 6,7 A1.SE A1.SW | 8,9 A2.SE A2.NE | 10,11 A3.SW A3.NW
// with A0 = O-9, A1 = O+9, A2 = O-1, A3 = O+1 (quoridor.py:301-351)
struct JumpPlan {
    int O;
    int8_t ref[12];
    int8_t val[12];
};
QZ_HD JumpPlan make_jump_plan(uint64_t hb, uint64_t vb, int O) {
    JumpPlan p;
    p.O = O;
    // the five tiles' corner references up front (clamped index, validity applied afterwards): five independent loads, one
    // round trip of the dependent chain instead of twelve
    const int tile5[5] = {O, O - 9, O + 9, O - 1, O + 1};
    uint32_t w5[5];
    bool ok5[5];
    for (int k = 0; k < 5; k++) {
        ok5[k] = tile5[k] >= 0 && tile5[k] <= 80;
        w5[k] = corner_ref4(ok5[k] ? tile5[k] : 0);
    }
    const int grp[12] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4};
    const int which[12] = {0, 1, 2, 3, 0, 1, 2, 3, 2, 1, 3, 0};
    for (int i = 0; i < 12; i++) {
        int rf = ok5[grp[i]] ? (int)((w5[grp[i]] >> (8 * which[i])) & 0xFFu) : 64;
        p.ref[i] = (int8_t)rf;
        p.val[i] = (int8_t)ref_value_fast(hb, vb, rf);
    }
    return p;
}
// jump_dests() for the current walls plus an optional candidate wall (cix < 0: none).  A
// candidate only changes corners that refer to its own (empty) slot.
QZ_HD BB bb_if(bool c, BB x) {  // c ? x : empty, without a branch
    const uint32_t m = c ? 0xFFFFFFFFu : 0u;
    return BB{x.w0 & m, x.w1 & m, x.w2 & m};
}
// Branch-free: this runs once per flood item, on divergent lanes -- four `if` blocks cost more in exec-mask
// bookkeeping than the selects they save (measured: ~250 -> ~170 instructions per call).
QZ_HD Jumps plan_jumps(const JumpPlan& p, int cix, bool horizontal) {
    const int H = 1, V = -1;
    int cv[12];
    int cval = horizontal ? 1 : -1;
    for (int i = 0; i < 12; i++) cv[i] = (cix >= 0 && p.ref[i] == cix) ? cval : p.val[i];
    const int O = p.O;
    const BB dN = dest_bit(O + 9), dS = dest_bit(O - 9), dE = dest_bit(O + 1), dW = dest_bit(O - 1);
    const int onw = cv[0], one = cv[1], ose = cv[2], osw = cv[3];
    Jumps j;
    // :301-314  (A0 = O-9: xnw = cv[4], xne = cv[5])
    const bool ok0 = O - 9 >= 0 && cv[5] != H && cv[4] != H;
    j.a[0] = ok0 ? O - 9 : -1;
    j.d[0] = bb_if(ok0, bb_or(bb_or(bb_if(onw != H && one != H, dN), bb_if(one != V && cv[5] != V, dE)), bb_if(onw != V && cv[4] != V, dW)));
    // :317-327  (A1 = O+9: xse = cv[6], xsw = cv[7])
    const bool ok1 = O + 9 <= 80 && cv[6] != H && cv[7] != H;
    j.a[1] = ok1 ? O + 9 : -1;
    j.d[1] = bb_if(ok1, bb_or(bb_or(bb_if(osw != H && ose != H, dS), bb_if(ose != V && cv[6] != V, dE)), bb_if(osw != V && cv[7] != V, dW)));
    // :330-339  (A2 = O-1: xse = cv[8], xne = cv[9])
    const bool ok2 = O - 1 >= 0 && cv[8] != V && cv[9] != V;
    j.a[2] = ok2 ? O - 1 : -1;
    j.d[2] = bb_if(ok2, bb_or(bb_or(bb_if(ose != V && one != V, dE), bb_if(one != H, dN)), bb_if(ose != H, dS)));
    // :342-351  (A3 = O+1: xsw = cv[10], xnw = cv[11])
    const bool ok3 = O + 1 <= 80 && cv[10] != V && cv[11] != V;
    j.a[3] = ok3 ? O + 1 : -1;
    j.d[3] = bb_if(ok3, bb_or(bb_or(bb_if(onw != V && osw != V, dW), bb_if(onw != H, dN)), bb_if(osw != H, dS)));
    return j;
}
QZ_HD Graph make_graph_plan(Blk blocked, const JumpPlan& plan, int cix, bool horizontal) {
    Graph g;
    g.cn = bb_not(blocked.n);
    g.cs = bb_not(blocked.s);
    g.ce = bb_not(blocked.e);
    g.cw = bb_not(blocked.w);
    g.notO = bb_not(dest_bit(plan.O));
    g.j = plan_jumps(plan, cix, horizontal);
    return g;
}
// simple moves only (no jump edges); the opponent's tile stays an obstacle
QZ_HD Graph make_graph_nojump(Blk blocked, int O) {
    Graph g;
    g.cn = bb_not(blocked.n);
    g.cs = bb_not(blocked.s);
    g.ce = bb_not(blocked.e);
    g.cw = bb_not(blocked.w);
    g.notO = bb_not(dest_bit(O));
    for (int k = 0; k < 4; k++) {
        g.j.a[k] = -1;
        g.j.d[k] = bb_zero();
    }
    return g;
}
// what a candidate wall adds to the blocked sets, written out bit by bit (== candidate_delta)
QZ_HD Blk candidate_delta_fast(int ix, bool horizontal) {
    int r = ix >> 3, c = ix & 7, b = 9 * r + c;
    Blk k;
    k.n = k.s = k.e = k.w = bb_zero();
    if (horizontal) {
        // N blocked from (r,c) and (r,c+1); on row 0 only from (0,c+1), plus tile 0 for slot 0
        if (r > 0) k.n = bb_or(bb_bit(b), bb_bit(b + 1));
        else k.n = bb_or(bb_bit(b + 1), c == 0 ? bb_bit(0) : bb_zero());
        k.s = bb_or(bb_bit(b + 9), bb_bit(b + 10));
    } else {
        // E blocked from (r,c) and (r+1,c); on row 0: tile (0,c+1) for c<=6 instead of (0,c), tile 0 for slot 0
        if (r > 0) k.e = bb_or(bb_bit(b), bb_bit(b + 9));
        else k.e = bb_or(bb_bit(b + 9), bb_or(c <= 6 ? bb_bit(b + 1) : bb_zero(), c == 0 ? bb_bit(0) : bb_zero()));
        k.w = bb_or(bb_bit(b + 1), bb_bit(b + 10));
    }
    return k;
}
// the flood with two extra early exits:
//  * `also`: tiles from which the goal is known to stay reachable whatever this candidate
//    does (the part of the player's base path behind the last edge the candidate removes);
//  * jump sources are tested only when the reached set touches one of them.
//  * `from`: any set of tiles known to be connected to the player's pawn under this candidate
//    (the pawn tile itself, or the part of the base path in front of the first edge the
//    candidate removes): the flood then only has to find a way AROUND the wall.
QZ_HD bool flood_to(const Graph& g, BB from, BB goal_or_safe) {
    BB R = from;
    if (bb_any(bb_and(R, goal_or_safe))) return true;
    const BB jsrc = jump_sources(g);
    // two layers per trip: the exit tests (and, on the GPU, the exec-mask bookkeeping of a
    // divergent loop) cost about as much as a layer, so test every other layer
    for (int it = 0; it < 48; it++) {
        BB n1 = expand_j(g, jsrc, R);
        BB R1 = bb_or(R, n1);
        BB n2 = expand_j(g, jsrc, R1);
        BB R2 = bb_or(R1, n2);
        if (bb_any(bb_and(bb_or(n1, n2), goal_or_safe))) return true;
        if (bb_eq(R2, R)) return false;
        R = R2;
    }
    return false;
}
QZ_HD bool flood_to(const Graph& g, int start, BB goal_or_safe) { return flood_to(g, bb_bit(start), goal_or_safe); }

// One concrete start->goal path WITH its order, for the pooled kernel.
//   layers: caller storage, entry i at layers[i*stride], at least max_layers+1 entries
//   tiles[0..len]: the path's tiles (tiles[0] = start), kinds[i]: move from tiles[i] to
//   tiles[i+1]: 0 N, 1 S, 2 E, 3 W, 4 jump.
// If the goal is further than max_layers the path is reported as found with len = -1 and the
// caller must treat every candidate as cutting it (exact, just slower).
struct OrderedPath {
    PathEdges e;
    int len;  // number of edges, 0 if !found, -1 if found but longer than the layer store
    BB last;  // find_path_tables: tab.suffix[len - 1] (every tile of the path but the start), so that nobody reads it back
};
QZ_HD OrderedPath find_path_ordered(const Graph& g, int start, BB goal, BB* layers, int stride, int max_layers,
                                    uint8_t* tiles, uint8_t* kinds, int tstride) {
    OrderedPath p;
    p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_zero();
    p.e.jump = false;
    p.e.found = false;
    p.len = 0;
    BB R = bb_bit(start);
    layers[0] = R;
    int L = 0;
    BB hit = bb_zero();
    const BB jsrc = jump_sources(g);
    for (int it = 0; it < 81; it++) {
        BB nx = expand_j(g, jsrc, R);
        hit = bb_and(nx, goal);
        BB R2 = bb_or(R, nx);
        if (bb_any(hit)) {
            L = it + 1;
            p.e.found = true;
            break;
        }
        if (bb_eq(R2, R)) return p;
        R = R2;
        if (it + 1 <= max_layers) layers[(it + 1) * stride] = R;
    }
    if (!p.e.found) return p;
    if (L > max_layers) {  // layers beyond the store were not kept: conservative answer
        p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_not(bb_zero());
        p.e.jump = true;
        p.len = -1;
        return p;
    }
    // walk back; fill tiles/kinds from the end
    int t = bb_lowest(hit);
    int n = 0;  // edges found so far (stored reversed at the tail, compacted below)
    int pos = L;
    tiles[pos * tstride] = (uint8_t)t;
    for (int i = L; i >= 1; i--) {
        BB prev = layers[(i - 1) * stride];
        if (bb_test(prev, t)) continue;
        int s, kind;
        if (t >= 9 && bb_test(prev, t - 9) && bb_test(g.cn, t - 9)) {
            s = t - 9;
            kind = 0;
            p.e.pn = bb_or(p.e.pn, bb_bit(s));
        } else if (t <= 71 && bb_test(prev, t + 9) && bb_test(g.cs, t + 9)) {
            s = t + 9;
            kind = 1;
            p.e.ps = bb_or(p.e.ps, bb_bit(s));
        } else if (t >= 1 && bb_test(prev, t - 1) && bb_test(g.ce, t - 1)) {
            s = t - 1;
            kind = 2;
            p.e.pe = bb_or(p.e.pe, bb_bit(s));
        } else if (t <= 79 && bb_test(prev, t + 1) && bb_test(g.cw, t + 1)) {
            s = t + 1;
            kind = 3;
            p.e.pw = bb_or(p.e.pw, bb_bit(s));
        } else {
            s = -1;
            for (int k = 0; k < 4; k++)
                if (s < 0 && g.j.a[k] >= 0 && bb_test(prev, g.j.a[k]) && bb_test(g.j.d[k], t)) s = g.j.a[k];
            kind = 4;
            p.e.jump = true;
            if (s < 0) {
                p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_not(bb_zero());
                p.len = -1;
                return p;
            }
        }
        pos--;
        tiles[pos * tstride] = (uint8_t)s;
        kinds[pos * tstride] = (uint8_t)kind;
        n++;
        t = s;
    }
    // the path occupies tiles[pos..L], kinds[pos..L-1]; shift it to index 0
    for (int i = 0; i <= n; i++) tiles[i * tstride] = tiles[(pos + i) * tstride];
    for (int i = 0; i < n; i++) kinds[i * tstride] = kinds[(pos + i) * tstride];
    p.len = n;
    return p;
}
// One concrete shortest start->goal path and the two lookup tables P3 needs, with NO layer
// store: the flood remembers, per tile, which KIND of move discovered it first ("came-from"
// sets, one per move kind, fixed priority N,S,E,W,jump0..3), so the walk back from the goal
// is a chain of set-membership tests.  Tables are indexed by REVERSE position (edge 0 = the
// edge that enters the goal row, edge 1 the one before it, ...):
//   tab.srcpos[t] = k   if the path's k-th edge from the end leaves tile t (255 otherwise)
//   tab.suffix[k]       = tiles behind that edge (its destination and everything after it)
// first_jump_r = reverse position of the jump edge closest to the goal (-1: none).
// len = number of edges (0 if !found).  tab.suffix must hold max_edges entries; a longer path
// is reported with len = -1 and all-ones path sets (every candidate then gets re-checked).
// find_path_walk: the search + the walk back; `on_found()` is called once a path exists (before the walk), `on_edge(k, s, t,
// acc)` for every edge, k counting from the GOAL end: the edge leaves tile s for tile t, acc = t and every tile behind it.
// find_path_tables writes them into lookup tables; the pooled pipeline's hand-off keeps the bare tile sequence (PathSeq).
template <typename OnFound, typename OnEdge>
QZ_HD OrderedPath find_path_walk(const Graph& g, int start, BB goal, int max_edges, int& first_jump_r, int& far_jump_r, OnFound on_found,
                                 OnEdge on_edge) {
    OrderedPath p;
    p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_zero();
    p.e.jump = false;
    p.e.found = false;
    p.len = 0;
    p.last = bb_zero();
    first_jump_r = -1;
    far_jump_r = -1;  // reverse position of the jump edge closest to the START
    BB R = bb_bit(start);
    BB fN = bb_zero(), fS = bb_zero(), fE = bb_zero(), fW = bb_zero();
    BB fJ[4] = {bb_zero(), bb_zero(), bb_zero(), bb_zero()};
    const BB jsrc = jump_sources(g);
    BB hit = bb_zero();
    for (int it = 0; it < 81; it++) {
        BB aN = bb_and(bb_shl<9>(bb_and(R, g.cn)), g.notO);
        BB aS = bb_and(bb_shr<9>(bb_and(R, g.cs)), g.notO);
        BB aE = bb_and(bb_shl<1>(bb_and(R, g.ce)), g.notO);
        BB aW = bb_and(bb_shr<1>(bb_and(R, g.cw)), g.notO);
        BB nx = bb_or(bb_or(aN, aS), bb_or(aE, aW));
        BB fresh = bb_andn(nx, R);  // first reached by a simple move in this layer
        fN = bb_or(fN, bb_and(aN, fresh));
        fresh = bb_andn(fresh, aN);
        fS = bb_or(fS, bb_and(aS, fresh));
        fresh = bb_andn(fresh, aS);
        fE = bb_or(fE, bb_and(aE, fresh));
        fresh = bb_andn(fresh, aE);
        fW = bb_or(fW, bb_and(aW, fresh));
        if (bb_any(bb_and(R, jsrc))) {
            BB seen = bb_or(R, nx);
            for (int k = 0; k < 4; k++) {
                if (g.j.a[k] >= 0 && bb_test(R, g.j.a[k])) {
                    BB nj = bb_andn(g.j.d[k], seen);
                    fJ[k] = bb_or(fJ[k], nj);
                    seen = bb_or(seen, nj);
                    nx = bb_or(nx, g.j.d[k]);
                }
            }
        }
        hit = bb_and(nx, goal);
        if (bb_any(hit)) {
            p.e.found = true;
            break;
        }
        BB R2 = bb_or(R, nx);
        if (bb_eq(R2, R)) return p;
        R = R2;
    }
    if (!p.e.found) return p;
    on_found();
    int t = bb_lowest(hit);
    BB acc = bb_zero();
    int k = 0;
    for (int guard = 0; guard < 82 && t != start; guard++) {
        int s;
        bool jump = false;
        if (bb_test(fN, t)) {
            s = t - 9;
            p.e.pn = bb_or(p.e.pn, bb_bit(s));
        } else if (bb_test(fS, t)) {
            s = t + 9;
            p.e.ps = bb_or(p.e.ps, bb_bit(s));
        } else if (bb_test(fE, t)) {
            s = t - 1;
            p.e.pe = bb_or(p.e.pe, bb_bit(s));
        } else if (bb_test(fW, t)) {
            s = t + 1;
            p.e.pw = bb_or(p.e.pw, bb_bit(s));
        } else {
            s = -1;
            for (int q = 0; q < 4; q++)
                if (s < 0 && bb_test(fJ[q], t)) s = g.j.a[q];
            jump = true;
            p.e.jump = true;
        }
        if (s < 0 || k >= max_edges) {  // cannot happen / path longer than the tables: conservative answer
            p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_not(bb_zero());
            p.e.jump = true;
            p.len = -1;
            return p;
        }
        acc = bb_or(acc, bb_bit(t));  // t and everything behind it
        on_edge(k, s, t, acc);
        if (jump && first_jump_r < 0) first_jump_r = k;
        if (jump) far_jump_r = k;
        k++;
        t = s;
    }
    p.len = k;
    p.last = acc;
    return p;
}
template <typename Tab>
QZ_HD OrderedPath find_path_tables(const Graph& g, int start, BB goal, int max_edges, Tab& tab, int& first_jump_r,
                                   int& far_jump_r) {
    return find_path_walk(
        g, start, goal, max_edges, first_jump_r, far_jump_r,
        [&]() {
            for (int i = 0; i < 21; i++) reinterpret_cast<uint32_t*>(tab.srcpos)[i] = 0xFFFFFFFFu;
        },
        [&](int k, int s, int, const BB& acc) {
            tab.suffix[k] = acc;
            tab.srcpos[s] = (uint8_t)k;
        });
}

// Tiles of the base path from which the goal stays reachable whatever candidate (ix, horizontal)
// does: everything behind the LAST edge it may remove.  `delta` = candidate_delta(ix, horizontal).
QZ_HD BB safe_suffix(const uint8_t* tiles, const uint8_t* kinds, int tstride, int len, const Blk& delta,
                     bool near_o) {
    BB safe = bb_zero();
    for (int i = len - 1; i >= 0; i--) {
        int s = tiles[i * tstride], kind = kinds[i * tstride];
        bool cut;
        if (kind == 0) cut = bb_test(delta.n, s);
        else if (kind == 1) cut = bb_test(delta.s, s);
        else if (kind == 2) cut = bb_test(delta.e, s);
        else if (kind == 3) cut = bb_test(delta.w, s);
        else cut = near_o;
        safe = bb_or(safe, bb_bit(tiles[(i + 1) * tstride]));
        if (cut) return safe;
    }
    return bb_not(bb_zero());  // nothing cut (the caller does not flood in that case)
}

// ----------------------------------------------------------------------------- encoder
// quoridor.py:58-131: value of element idx (= plane*81 + row*9 + col) of state()
QZ_HD float plane_value(const Board& b, int idx) {
    int plane = idx / 81, cell = idx - 81 * plane;
    if (plane >= 5) {
        int wm = b.cur == 1 ? b.w1 : b.w2, wo = b.cur == 1 ? b.w2 : b.w1;
        int im = wm - 1, io = wo - 1;  // Python index: -1 -> last plane (quoridor.py:79-80)
        if (im < 0) im += 10;
        if (io < 0) io += 10;
        bool on = (plane == 5 + im) || (plane == 15 + io) || (plane == 25 && b.cur == 2);
        return on ? 1.0f : 0.0f;
    }
    if (plane >= 3) {
        int pm = b.cur == 1 ? b.p1 : b.p2, po = b.cur == 1 ? b.p2 : b.p1;
        int p = plane == 3 ? pm : po;
        if (p < 0) p += 81;
        return cell == p ? 1.0f : 0.0f;
    }
    int r = cell / 9, c = cell - 9 * r;
    if (r >= 8 || c >= 8) return 0.0f;  // np.pad(..., (0,1)) quoridor.py:83-102
    int ix = 8 * r + c;
    uint64_t bits = plane == 0 ? ~(b.hb | b.vb) : (plane == 1 ? b.vb : b.hb);
    return (float)((bits >> ix) & 1ull);
}

}  // namespace qz

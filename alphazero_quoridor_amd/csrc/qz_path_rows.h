// qz_path_rows.h -- the base-path search of the rules op with one LANE PER BOARD ROW.
//
// find_path_tables() (qz_rules.h) runs one search per lane on 81-bit sets held in three words:
// ~120-200 instructions per flood layer and per walk-back step, all on one dependent chain, and
// that chain is what a rules kernel waits for.  Here nine neighbouring lanes share
// a search, lane r holding row r of every set as 9 bits: a north / south move is a one-lane
// shift of the whole wavefront (DPP wave_shr / wave_shl; rows 0 and 8 never move outwards, so
// nothing leaks into the neighbouring group), an east / west move a one-bit shift.  Seven
// searches fit a wavefront; two flood layers cost ~88 instructions, two walk-back steps ~64
// (measured in k_wave_rules: ~95 instructions and ~850 cycles per path edge, against ~300
// instructions with one search per lane).  It pays where FEW searches decide a launch's duration
// (k_wave_rules: 4,096 leaf boards, a few hundred of them searching); with a search for every
// lane of the chip the one-lane form needs a third of the issue slots.
//
//   coop_find_path()   the device form (SIMT, all 64 lanes of a wavefront call it together)
//   find_path_rows()   the same algorithm written over arrays of nine rows: the host-check build
//                      runs it in place of find_path_tables() and compares the masks with the
//                      oracle (tests/hostcheck), so the formulation is checked without a GPU
//   blocked_rows(), cut_row(), jump_plan_corner(), plan_jump_rows()
//                      the pieces around the search in the same row form (shared by both forms
//                      and by the host check, which compares them with the three-word functions)
//
// Both produce what find_path_tables() produces -- one concrete shortest path as edge sets, its
// length, the two lookup tables of the flood phase and the jump positions -- but not necessarily
// the SAME shortest path (ties are broken by the came-from priority N, S, E, W, jumps in both;
// the goal tile is the lowest hit in both: in practice they agree).  Path-cut pruning is exact
// for any concrete path, so the legal sets do not depend on the choice.
#pragma once
#include "qz_rules.h"

namespace qz {

// row r (bits 9r .. 9r+8) of an 81-bit set
QZ_HD uint32_t bb_row(BB a, int r) {
    const int pos = 9 * r, w = pos >> 5, off = pos & 31;
    const uint32_t lo = w == 0 ? a.w0 : (w == 1 ? a.w1 : a.w2);
    const uint32_t hi = w == 0 ? a.w1 : (w == 1 ? a.w2 : 0u);
    return (uint32_t)(((((uint64_t)hi) << 32) | (uint64_t)lo) >> off) & 0x1FFu;
}
// word w (0..2) of the 81-bit set whose rows are rw[0..8]
template <typename T>
QZ_HD uint32_t rows_word(const T* rw, int w) {
    if (w == 0) return (uint32_t)rw[0] | ((uint32_t)rw[1] << 9) | ((uint32_t)rw[2] << 18) | ((uint32_t)rw[3] << 27);
    if (w == 1)
        return ((uint32_t)rw[3] >> 5) | ((uint32_t)rw[4] << 4) | ((uint32_t)rw[5] << 13) | ((uint32_t)rw[6] << 22) |
               ((uint32_t)rw[7] << 31);
    return ((uint32_t)rw[7] >> 1) | ((uint32_t)rw[8] << 8);
}
template <typename T>
QZ_HD BB bb_from_rows(const T* rw) { return BB{rows_word(rw, 0), rows_word(rw, 1), rows_word(rw, 2)}; }

// Row r of the four blocked sets (walls + borders) straight from the wall bytes:
// == bb_row(blk_or(blocked_from(spread8(hb), spread8(vb)), blocked_borders()).n|s|e|w, r), incl. the row-0 quirks
// (tests/hostcheck compares the two on every test position).
QZ_HD void blocked_rows(uint64_t hb, uint64_t vb, int r, uint32_t& bn, uint32_t& bs, uint32_t& be, uint32_t& bw) {
    const uint32_t h = r < 8 ? (uint32_t)(hb >> (8 * (r & 7))) & 0xFFu : 0u;
    const uint32_t hm = r > 0 ? (uint32_t)(hb >> (8 * ((r - 1) & 7))) & 0xFFu : 0u;
    const uint32_t v = r < 8 ? (uint32_t)(vb >> (8 * (r & 7))) & 0xFFu : 0u;
    const uint32_t vm = r > 0 ? (uint32_t)(vb >> (8 * ((r - 1) & 7))) & 0xFFu : 0u;
    bn = r == 8 ? 0x1FFu : (r == 0 ? ((h << 1) | (h & 1u)) : (h | (h << 1)));
    bs = r == 0 ? 0x1FFu : (hm | (hm << 1));
    be = (r == 0 ? (((v << 1) & 0xFEu) | (v & 1u)) : (v | vm)) | 0x100u;
    bw = (((v | vm) << 1) | 1u) & 0x1FFu;
}

// Row r (a byte, r < 8) of path_cut_masks(): the slots whose horizontal / vertical wall removes an edge of the path,
// from row r and row r + 1 (`*_up`) of the path's edge sets
QZ_HD void cut_row(int r, uint32_t pn, uint32_t ps_up, uint32_t pe, uint32_t pe_up, uint32_t pw, uint32_t pw_up, uint32_t& h, uint32_t& v) {
    const uint32_t ca = pn & 0xFFu, cb = (pn >> 1) & 0xFFu;
    h = (r == 0 ? (cb | (ca & 1u)) : (ca | cb)) | (ps_up & 0xFFu) | ((ps_up >> 1) & 0xFFu);
    const uint32_t e0 = pe & 0xFFu, e1 = (pe >> 1) & 0xFFu, e9 = pe_up & 0xFFu;
    v = (r == 0 ? (e9 | (e1 & 0x7Fu) | (e0 & 1u)) : (e0 | e9)) | ((pw >> 1) & 0xFFu) | ((pw_up >> 1) & 0xFFu);
}
// make_jump_plan() one corner at a time (i = 0..11): what lane i of a plan's twelve lanes computes in k_wave_rules
QZ_HD void jump_plan_corner(uint64_t hb, uint64_t vb, int O, int i, int& ref, int& val) {
    const int grp = (int)((0x443322110000ull >> (4 * i)) & 15ull);  // {0,0,0,0,1,1,2,2,3,3,4,4}
    const int which = (int)((0x36E4E4u >> (2 * i)) & 3u);             // {0,1,2,3,0,1,2,3,2,1,3,0}
    const int tile = O + (grp == 0 ? 0 : (grp == 1 ? -9 : (grp == 2 ? 9 : (grp == 3 ? -1 : 1))));
    const bool ok = tile >= 0 && tile <= 80;
    const uint32_t w = corner_ref4(ok ? tile : 0);
    ref = ok ? (int)((w >> (8 * which)) & 0xFFu) : 64;
    val = ref_value_fast(hb, vb, ref);
}
// plan_jumps(plan, no candidate) in rows: row r of the four jump sources (jb) and of their destination sets (jd)
QZ_HD void plan_jump_rows(const int8_t* cv, int O, int r, uint32_t jb[4], uint32_t jd[4]) {
    const int H = 1, V = -1;
    const int t0 = 9 * r;
    const int tn = O + 9, ts = O - 9, te = O + 1, tw = O - 1;
    const uint32_t rN = (tn <= 80 && tn >= t0 && tn < t0 + 9) ? (1u << (tn - t0)) : 0u;
    const uint32_t rS = (ts >= 0 && ts >= t0 && ts < t0 + 9) ? (1u << (ts - t0)) : 0u;
    const uint32_t rE = (te <= 80 && te >= t0 && te < t0 + 9) ? (1u << (te - t0)) : 0u;
    const uint32_t rW = (tw >= 0 && tw >= t0 && tw < t0 + 9) ? (1u << (tw - t0)) : 0u;
    const int onw = cv[0], one = cv[1], ose = cv[2], osw = cv[3];
    const bool ok0 = O - 9 >= 0 && cv[5] != H && cv[4] != H;
    const bool ok1 = O + 9 <= 80 && cv[6] != H && cv[7] != H;
    const bool ok2 = O - 1 >= 0 && cv[8] != V && cv[9] != V;
    const bool ok3 = O + 1 <= 80 && cv[10] != V && cv[11] != V;
    jb[0] = ok0 ? rS : 0u;
    jb[1] = ok1 ? rN : 0u;
    jb[2] = ok2 ? rW : 0u;
    jb[3] = ok3 ? rE : 0u;
    jd[0] = ok0 ? (((onw != H && one != H) ? rN : 0u) | ((one != V && cv[5] != V) ? rE : 0u) | ((onw != V && cv[4] != V) ? rW : 0u)) : 0u;
    jd[1] = ok1 ? (((osw != H && ose != H) ? rS : 0u) | ((ose != V && cv[6] != V) ? rE : 0u) | ((osw != V && cv[7] != V) ? rW : 0u)) : 0u;
    jd[2] = ok2 ? (((ose != V && one != V) ? rE : 0u) | ((one != H) ? rN : 0u) | ((ose != H) ? rS : 0u)) : 0u;
    jd[3] = ok3 ? (((onw != V && osw != V) ? rW : 0u) | ((onw != H) ? rN : 0u) | ((osw != H) ? rS : 0u)) : 0u;
}
#if !defined(__HIPCC__)
static long g_cut_row_mismatches = 0;  // host check: cut_row() against path_cut_masks() on every path found
#endif

// ---- array form (host check; also the specification of the SIMT form below) -----------------
template <typename Tab>
QZ_HD OrderedPath find_path_rows(const Graph& g, int start, BB goal, int max_edges, Tab& tab, int& first_jump_r, int& far_jump_r) {
    OrderedPath p;
    p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_zero();
    p.e.jump = false;
    p.e.found = false;
    p.len = 0;
    p.last = bb_zero();
    first_jump_r = -1;
    far_jump_r = -1;
    uint32_t cn[9], cs[9], ce[9], cw[9], nO[9], gl[9], jsrc[9], jb[4][9], jd[4][9];
    uint32_t R[9], front[9], fN[9], fS[9], fE[9], fW[9], fJ[4][9], fJany[9];
    for (int r = 0; r < 9; r++) {
        cn[r] = r == 8 ? 0u : bb_row(g.cn, r);
        cs[r] = r == 0 ? 0u : bb_row(g.cs, r);
        ce[r] = bb_row(g.ce, r) & 0xFFu;
        cw[r] = bb_row(g.cw, r) & 0x1FEu;
        nO[r] = bb_row(g.notO, r);
        gl[r] = bb_row(goal, r);
        jsrc[r] = 0u;
        for (int k = 0; k < 4; k++) {
            const int a = g.j.a[k];
            jb[k][r] = (a >= 0 && a <= 80 && a / 9 == r) ? (1u << (a % 9)) : 0u;
            jd[k][r] = a >= 0 ? bb_row(g.j.d[k], r) : 0u;
            jsrc[r] |= jb[k][r];
            fJ[k][r] = 0u;
        }
        R[r] = (start >= 0 && start <= 80 && start / 9 == r) ? (1u << (start % 9)) : 0u;
        front[r] = R[r];
        fN[r] = fS[r] = fE[r] = fW[r] = 0u;
    }
    uint32_t hit[9];
    int L = 0;
    for (int it = 0; it < 81; it++) {
        uint32_t nx[9], nf[9], aN[9], aS[9], aE[9], aW[9];
        for (int r = 0; r < 9; r++) {
            aN[r] = (r > 0 ? (R[r - 1] & cn[r - 1]) : 0u) & nO[r];
            aS[r] = (r < 8 ? (R[r + 1] & cs[r + 1]) : 0u) & nO[r];
            aE[r] = ((R[r] & ce[r]) << 1) & nO[r];
            aW[r] = ((R[r] & cw[r]) >> 1) & nO[r];
            nx[r] = aN[r] | aS[r] | aE[r] | aW[r];
            uint32_t fresh = nx[r] & ~R[r];  // first reached by a simple move in this layer
            fN[r] |= aN[r] & fresh;
            fresh &= ~aN[r];
            fS[r] |= aS[r] & fresh;
            fresh &= ~aS[r];
            fE[r] |= aE[r] & fresh;
            fresh &= ~aE[r];
            fW[r] |= aW[r] & fresh;
        }
        // a jump source fires once, in the layer after it joined the reached set (its destinations
        // are in the reached set from then on, so firing again would add nothing)
        bool anyj = false;
        for (int r = 0; r < 9; r++) anyj = anyj || (front[r] & jsrc[r]) != 0u;
        if (anyj) {
            uint32_t seen[9];
            for (int r = 0; r < 9; r++) seen[r] = R[r] | nx[r];
            for (int k = 0; k < 4; k++) {
                bool fire = false;
                for (int r = 0; r < 9; r++) fire = fire || (front[r] & jb[k][r]) != 0u;
                if (!fire) continue;
                for (int r = 0; r < 9; r++) {
                    const uint32_t nj = jd[k][r] & ~seen[r];
                    fJ[k][r] |= nj;
                    seen[r] |= nj;
                    nx[r] |= jd[k][r];
                }
            }
        }
        bool anyhit = false, anynew = false;
        for (int r = 0; r < 9; r++) {
            hit[r] = nx[r] & gl[r];
            nf[r] = nx[r] & ~R[r];
            anyhit = anyhit || hit[r] != 0u;
            anynew = anynew || nf[r] != 0u;
        }
        if (anyhit) {
            p.e.found = true;
            L = it + 1;
            break;
        }
        if (!anynew) return p;
        for (int r = 0; r < 9; r++) {
            R[r] |= nx[r];
            front[r] = nf[r];
        }
    }
    if (!p.e.found) return p;
    if (L > max_edges) {  // path longer than the tables: conservative answer (every candidate is re-checked)
        p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_not(bb_zero());
        p.e.jump = true;
        p.len = -1;
        return p;
    }
    for (int i = 0; i < 21; i++) reinterpret_cast<uint32_t*>(tab.srcpos)[i] = 0xFFFFFFFFu;
    uint32_t cur[9], acc[9], pn[9], ps[9], pe[9], pw[9];
    {
        bool taken = false;  // the lowest hit tile
        for (int r = 0; r < 9; r++) {
            cur[r] = (!taken && hit[r]) ? (hit[r] & (0u - hit[r])) : 0u;
            taken = taken || hit[r] != 0u;
            acc[r] = pn[r] = ps[r] = pe[r] = pw[r] = 0u;
            fJany[r] = fJ[0][r] | fJ[1][r] | fJ[2][r] | fJ[3][r];
        }
    }
    int k = 0;
    for (int guard = 0; guard < 82; guard++) {
        uint32_t nc[9], sN[9], sS[9], sE[9], sW[9];
        bool anyj = false, anyc = false;
        for (int r = 0; r < 9; r++) {
            sN[r] = r < 8 ? (cur[r + 1] & fN[r + 1]) : 0u;  // reached by a north move: it came from the row below
            sS[r] = r > 0 ? (cur[r - 1] & fS[r - 1]) : 0u;
            sE[r] = (cur[r] & fE[r]) >> 1;
            sW[r] = (cur[r] & fW[r]) << 1;
            nc[r] = sN[r] | sS[r] | sE[r] | sW[r];
            anyj = anyj || (cur[r] & fJany[r]) != 0u;
        }
        if (anyj) {
            for (int q = 0; q < 4; q++) {
                bool via = false;
                for (int r = 0; r < 9; r++) via = via || (cur[r] & fJ[q][r]) != 0u;
                if (!via) continue;
                for (int r = 0; r < 9; r++) nc[r] |= jb[q][r];
                p.e.jump = true;
                if (first_jump_r < 0) first_jump_r = k;
                far_jump_r = k;
                break;
            }
        }
        for (int r = 0; r < 9; r++) anyc = anyc || nc[r] != 0u;
        if (!anyc) break;  // cur is the start tile: nothing discovered it
        for (int r = 0; r < 9; r++) {
            acc[r] |= cur[r];  // this tile and everything behind it
            pn[r] |= sN[r];
            ps[r] |= sS[r];
            pe[r] |= sE[r];
            pw[r] |= sW[r];
            if (nc[r]) {
                int c = 0;
                while (!((nc[r] >> c) & 1u)) c++;
                tab.srcpos[9 * r + c] = (uint8_t)k;
            }
            cur[r] = nc[r];
        }
        tab.suffix[k] = bb_from_rows(acc);
        k++;
    }
    if (k != L) {  // cannot happen: the came-from sets describe shortest paths of exactly L edges
        p.e.pn = p.e.ps = p.e.pe = p.e.pw = bb_not(bb_zero());
        p.e.jump = true;
        p.len = -1;
        return p;
    }
    p.e.pn = bb_from_rows(pn);
    p.e.ps = bb_from_rows(ps);
    p.e.pe = bb_from_rows(pe);
    p.e.pw = bb_from_rows(pw);
    p.len = k;
    p.last = bb_from_rows(acc);
#if !defined(__HIPCC__)
    {
        uint64_t ch = 0, cv = 0;
        for (int r = 0; r < 8; r++) {
            uint32_t h, v;
            cut_row(r, pn[r], ps[r + 1], pe[r], pe[r + 1], pw[r], pw[r + 1], h, v);
            ch |= (uint64_t)h << (8 * r);
            cv |= (uint64_t)v << (8 * r);
        }
        const CutMasks cm = path_cut_masks(p.e);
        if (cm.h != ch || cm.v != cv) g_cut_row_mismatches++;
    }
#endif
    return p;
}

#if defined(__HIPCC__)
// ---- SIMT form ------------------------------------------------------------------------------
constexpr int COOP_GROUPS = 7;      // searches per wavefront (lanes 9g .. 9g+8; lane 63 idles)
constexpr int COOP_MAX_EDGES = 41;  // == POOL_MAX_LAYERS + 1
// value of the lane below / above in the wavefront (0 beyond its ends)
__device__ __forceinline__ uint32_t lane_below(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t lane_above(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true); }

// What a search reads (filled by the lane that built the board context) and leaves behind, in LDS.
struct CoopSearch {
    // in: the board's walls, the opponent's tile (an obstacle), the corner values around it (make_jump_plan: the jump
    // edges follow from them, plan_jump_rows), the pawn's tile, the goal row
    uint64_t hb, vb;
    const JumpPlan* plan;
    int32_t opp, start, goal_row;  // goal_row < 0: no search (the group idles)
    // out
    int32_t found, jump, len, first_jump, far_jump;
    uint8_t cut_h[8], cut_v[8];            // path_cut_masks() of the path found, one byte per slot row
    uint16_t sets[9][9];                   // rows of pn, ps, pe, pw, last; search 0 also leaves the board's blocked sets n, s, e, w
    uint16_t sfx[COOP_MAX_EDGES + 1][10];      // rows of suffix[k]; column 9 takes the stores of lanes outside every search
    uint8_t srcpos[84];  // tiles 0..80; byte 83 takes the stores of lanes that have nothing to say
};
// All 64 lanes of a wavefront call this together (full exec mask); group g = lane / 9 runs the search
// described by s[g] for g < n_searches.  Results are in s[g] when it returns (the caller needs a
// wave-level LDS fence before reading them from other lanes).
__device__ __forceinline__ void coop_find_path(CoopSearch* s, int n_searches) {
    const int lane = (int)(threadIdx.x & 63);
    const int grp = lane / 9, r = lane - 9 * grp;
    const uint32_t gbase = (uint32_t)(9 * grp);
    const bool mine = grp < n_searches && grp < COOP_GROUPS;
    CoopSearch& S = s[mine ? grp : 0];
    const int goal_row = mine ? S.goal_row : -1;
    bool act = goal_row >= 0;
    uint32_t cn = 0, cs = 0, ce = 0, cw = 0, nO = 0, jsrc = 0, jb0 = 0, jb1 = 0, jb2 = 0, jb3 = 0, jd0 = 0, jd1 = 0, jd2 = 0, jd3 = 0, R = 0;
    if (act) {
        uint32_t bn, bs, be, bw;  // the simple-move graph straight from the wall bytes: nothing to wait for
        blocked_rows(S.hb, S.vb, r, bn, bs, be, bw);
        cn = ~bn & 0x1FFu;
        cs = ~bs & 0x1FFu;
        ce = ~be & 0x1FFu;
        cw = ~bw & 0x1FFu;
        if (grp == 0) {  // for the board record (the flood phase works on three-word sets)
            S.sets[5][r] = (uint16_t)bn;
            S.sets[6][r] = (uint16_t)bs;
            S.sets[7][r] = (uint16_t)be;
            S.sets[8][r] = (uint16_t)bw;
        }
        const int t0 = 9 * r, opp = S.opp;
        nO = (opp >= t0 && opp < t0 + 9) ? (~(1u << (opp - t0)) & 0x1FFu) : 0x1FFu;
        const JumpPlan& P = *S.plan;
        int8_t cv[12];
        for (int k = 0; k < 12; k++) cv[k] = P.val[k];
        uint32_t jb[4], jd[4];
        plan_jump_rows(cv, P.O, r, jb, jd);
        jb0 = jb[0];
        jb1 = jb[1];
        jb2 = jb[2];
        jb3 = jb[3];
        jd0 = jd[0];
        jd1 = jd[1];
        jd2 = jd[2];
        jd3 = jd[3];
        jsrc = jb0 | jb1 | jb2 | jb3;
        const int st = S.start;
        R = (st >= t0 && st < t0 + 9) ? (1u << (st - t0)) : 0u;
    }
    const uint32_t gl = r == goal_row ? 0x1FFu : 0u;
    uint32_t front = R, fN = 0, fS = 0, fE = 0, fW = 0, fJ0 = 0, fJ1 = 0, fJ2 = 0, fJ3 = 0, hitrow = 0;
    bool found = false;
    int L = 0;
#define QZ_GANY(pred) ((uint32_t)(__ballot(pred) >> gbase) & 0x1FFu)
    // one flood layer from the reached set Rin whose newest tiles are fin: came-from bookkeeping, returns the generated set
    auto layer = [&](const uint32_t Rin, const uint32_t fin) -> uint32_t {
        const uint32_t aN = lane_below(Rin & cn) & nO;
        const uint32_t aS = lane_above(Rin & cs) & nO;
        const uint32_t aE = ((Rin & ce) << 1) & nO;
        const uint32_t aW = ((Rin & cw) >> 1) & nO;
        uint32_t nx = aN | aS | aE | aW;
        uint32_t fresh = nx & ~Rin;
        fN |= aN & fresh;
        fresh &= ~aN;
        fS |= aS & fresh;
        fresh &= ~aS;
        fE |= aE & fresh;
        fresh &= ~aE;
        fW |= aW & fresh;
        if (__builtin_expect(__ballot((fin & jsrc) != 0u) != 0ull, 0)) {  // wave-uniform, rare: the frontier touches a tile next to the opponent
            uint32_t seen = Rin | nx;
            uint32_t nj;
            if (QZ_GANY((fin & jb0) != 0u)) { nj = jd0 & ~seen; fJ0 |= nj; seen |= nj; nx |= jd0; }
            if (QZ_GANY((fin & jb1) != 0u)) { nj = jd1 & ~seen; fJ1 |= nj; seen |= nj; nx |= jd1; }
            if (QZ_GANY((fin & jb2) != 0u)) { nj = jd2 & ~seen; fJ2 |= nj; seen |= nj; nx |= jd2; }
            if (QZ_GANY((fin & jb3) != 0u)) { nj = jd3 & ~seen; fJ3 |= nj; seen |= nj; nx |= jd3; }
        }
        return nx;
    };
    // Two layers per trip: the ballots and the taken branch of a trip cost about as much as a layer.  If the goal is
    // hit in the first layer of a pair the second one has still run: it only marks tiles that the first did not
    // reach (every tile gets its came-from mark once, when it is first reached), so the walk back is not affected.
    for (int it = 0; it < 41; it++) {
        const uint32_t nx1 = layer(R, front);
        const uint32_t hit1 = nx1 & gl, nf1 = nx1 & ~R, R1 = R | nx1;
        const uint32_t nx2 = layer(R1, nf1);
        const uint32_t hit2 = nx2 & gl, nf2 = nx2 & ~R1;
        const uint32_t gh1 = QZ_GANY(hit1 != 0u), gh2 = QZ_GANY(hit2 != 0u);
        // Branch-free bookkeeping (a divergent `if` costs more scalar exec-mask work than the layers).  A finished
        // group keeps R and front, and everything above is idempotent on unchanged inputs, so its lanes may run on:
        // hitrow stays what it was in the trip that found the goal.  A group that cannot reach its goal (no legal
        // position has one) stops generating tiles and idles until the others are done.
        const bool hitnow = act && (gh1 | gh2) != 0u, cont = act && (gh1 | gh2) == 0u;
        L += act ? (gh1 != 0u ? 1 : 2) : 0;
        found = found || hitnow;
        hitrow = gh1 != 0u ? hit1 : hit2;
        R = cont ? (R1 | nx2) : R;
        front = cont ? nf2 : front;
        act = cont;
        if (__ballot(act && (nf1 | nf2) != 0u) == 0ull) break;  // every group is done or stuck
    }
    const bool toolong = found && L > COOP_MAX_EDGES;
    // the lowest hit tile (hits are confined to the goal row, i.e. to one lane of the group)
    uint32_t cur = (found && !toolong) ? (hitrow & (0u - hitrow)) : 0u;
    if (mine && goal_row >= 0)
        for (int i = r; i < 21; i += 9) reinterpret_cast<uint32_t*>(S.srcpos)[i] = 0xFFFFFFFFu;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t fJany = fJ0 | fJ1 | fJ2 | fJ3;
    uint32_t acc = 0, pn = 0, ps = 0, pe = 0, pw = 0;
    int first_jump = -1, far_jump = -1;
    bool jump = false;
    // one step of the walk back: from the tile `c` to the tile it was discovered from (0 everywhere once a group is
    // back at its start tile: nothing discovered that one).  No branches but the rare jump test.  A group that is
    // done ORs its start tile into acc (harmless: `last` is only used together with the start tile) and writes
    // entries k >= len of sfx (never read); lanes without the new tile aim their srcpos byte at the padding.
    auto step = [&](const uint32_t c, const int k) -> uint32_t {
        const uint32_t sN = lane_above(c & fN);  // reached by a north move: it came from the row below
        const uint32_t sS = lane_below(c & fS);
        const uint32_t sE = (c & fE) >> 1;
        const uint32_t sW = (c & fW) << 1;
        uint32_t nc = sN | sS | sE | sW;
        if (__builtin_expect(__ballot((c & fJany) != 0u) != 0ull, 0)) {  // wave-uniform, rare
            bool via = false;
            const uint32_t v0 = QZ_GANY((c & fJ0) != 0u), v1 = QZ_GANY((c & fJ1) != 0u), v2 = QZ_GANY((c & fJ2) != 0u),
                           v3 = QZ_GANY((c & fJ3) != 0u);
            if (v0) { nc |= jb0; via = true; }
            else if (v1) { nc |= jb1; via = true; }
            else if (v2) { nc |= jb2; via = true; }
            else if (v3) { nc |= jb3; via = true; }
            if (via) {
                jump = true;
                if (first_jump < 0) first_jump = k;
                far_jump = k;
            }
        }
        acc |= c;
        S.sfx[k][mine ? r : 9] = (uint16_t)acc;
        S.srcpos[nc != 0u ? 9 * r + (__ffs((int)nc) - 1) : 83] = (uint8_t)k;
        pn |= sN;
        ps |= sS;
        pe |= sE;
        pw |= sW;
        return nc;
    };
    for (int k = 0; k < COOP_MAX_EDGES; k += 2) {  // two steps per trip
        const uint32_t mid = step(cur, k);
        cur = step(mid, k + 1);
        if (__ballot(cur != 0u) == 0ull) break;  // every group is back at its start tile
    }
#undef QZ_GANY
    // path_cut_masks() in rows: slot (r, c) cuts the path if its wall removes one of the path's edges
    pn = toolong ? 0x1FFu : pn;
    ps = toolong ? 0x1FFu : ps;
    pe = toolong ? 0x1FFu : pe;
    pw = toolong ? 0x1FFu : pw;
    const uint32_t ps_up = lane_above(ps), pe_up = lane_above(pe), pw_up = lane_above(pw);  // row r + 1 (lane 8 is not used)
    uint32_t cut_h, cut_v;
    cut_row(r, pn, ps_up, pe, pe_up, pw, pw_up, cut_h, cut_v);
    if (mine && goal_row >= 0) {
        if (r < 8) {
            S.cut_h[r] = (uint8_t)cut_h;
            S.cut_v[r] = (uint8_t)cut_v;
        }
        S.sets[0][r] = (uint16_t)pn;
        S.sets[1][r] = (uint16_t)ps;
        S.sets[2][r] = (uint16_t)pe;
        S.sets[3][r] = (uint16_t)pw;
        S.sets[4][r] = (uint16_t)(toolong ? 0u : acc);
        if (r == 0) {
            S.found = found ? 1 : 0;
            S.jump = (jump || toolong) ? 1 : 0;
            S.len = !found ? 0 : (toolong ? -1 : L);
            S.first_jump = first_jump;
            S.far_jump = far_jump;
        }
    }
}
#endif  // __HIPCC__

}  // namespace qz

/*
 * qz_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar restatement of cryer/AlphaZero_Quoridor's rules engine (quoridor.py) in
 * the reference's own control flow: per-candidate wall, copy the walls, run two
 * FIFO breadth-first searches that re-derive pawn moves tile by tile.  Nothing
 * here is shared with the HIP path (which uses bitboards + path-cut pruning);
 * that is the point of an oracle.
 *
 * Parity: pinned against tests/golden/ (npz fixtures) (generated from the real reference).
 */
#include "qz_oracle.h"

#include <string.h>

#define H_WALL 1  /* quoridor.py:6   HORIZONTAL = 1  */
#define V_WALL (-1) /* quoridor.py:7 VERTICAL = -1 */

/* Python's // and % (floor semantics) -- tiles may be negative after an
 * off-board SS jump (quoridor.py:229) */
static int py_div(int a, int b) {
    int q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}
static int py_mod(int a, int b) {
    int m = a % b;
    if (m != 0 && ((m < 0) != (b < 0))) m += b;
    return m;
}

/* numpy 1-d indexing: negative indices wrap once, otherwise IndexError */
static int at(const int8_t* inter, int idx, int* err) {
    if (idx < 0) idx += 64;
    if (idx < 0 || idx >= 64) {
        *err = 1;
        return 0;
    }
    return inter[idx];
}

/* quoridor.py:26-56 */
void qzo_reset(qzo_game* g) {
    memset(g, 0, sizeof(*g));
    g->cur = 1;
    g->last = -1;
    g->pos[1] = 4;
    g->pos[2] = 76;
    g->walls[1] = 10;
    g->walls[2] = 10;
}

/* quoridor.py:356-418 -- branch priority N-border, S-border, W-border, E-border.
 * out = {NW, NE, SE, SW}.  Line 388/392: `nw = ne = inter[...]` overwrites NE on
 * the south border (the "row-0 bug"); restated literally. */
int qzo_get_intersections(const int8_t* inter, int t, int out[4]) {
    int err = 0;
    int row = py_div(t, 9);                    /* :358 */
    int n_border = t > 71;                     /* :360 */
    int e_border = py_mod(t, 9) == 8;          /* :361 */
    int s_border = t < 9;                      /* :362 */
    int w_border = py_mod(t, 9) == 0;          /* :363 */
    int nw = 0, ne = 0, se = 0, sw = 0;
    if (n_border) {                            /* :365-378 */
        ne = 1;
        if (w_border) {
            nw = -1;
            sw = -1;
            se = at(inter, (t - 9) - (row - 1), &err);
        } else if (e_border) {
            nw = 1;
            se = -1;
            sw = at(inter, (t - 9) - (row - 1) - 1, &err);
        } else {
            nw = 1;
            sw = at(inter, (t - 9) - (row - 1) - 1, &err);
            se = at(inter, (t - 9) - (row - 1), &err);
        }
    } else if (s_border) {                     /* :379-392 */
        sw = 1;
        if (w_border) {
            nw = -1;
            se = 1;
            ne = at(inter, t - row, &err);
        } else if (e_border) {
            se = -1;
            ne = -1;
            nw = ne = at(inter, t - row - 1, &err); /* :388 overwrites ne */
        } else {
            se = 1;
            ne = at(inter, t - row, &err);          /* :391 (dead store, but can raise) */
            nw = ne = at(inter, t - row - 1, &err); /* :392 overwrites ne */
        }
    } else if (w_border) {                     /* :396-400 */
        nw = -1;
        sw = -1;
        ne = at(inter, t - row, &err);
        se = at(inter, (t - 9) - (row - 1), &err);
    } else if (e_border) {                     /* :402-406 */
        ne = -1;
        se = -1;
        nw = at(inter, t - row - 1, &err);
        sw = at(inter, (t - 9) - (row - 1) - 1, &err);
    } else {                                   /* :409-413 */
        ne = at(inter, t - row, &err);
        nw = at(inter, t - row - 1, &err);
        sw = at(inter, (t - 9) - (row - 1) - 1, &err);
        se = at(inter, (t - 9) - (row - 1), &err);
    }
    out[0] = nw;
    out[1] = ne;
    out[2] = se;
    out[3] = sw;
    return err ? QZO_INDEX_ERROR : QZO_OK;
}

enum { NW = 0, NE = 1, SE = 2, SW = 3 };
enum { A_N = 0, A_S, A_E, A_W, A_NN, A_SS, A_EE, A_WW, A_NE, A_NW, A_SE, A_SW }; /* :39-43 */

/* quoridor.py:272-353 */
int qzo_valid_pawn_actions(const int8_t* inter, int loc, int opp, int player, int* out) {
    int n_out = 0;
    int X[4], O[4];
    int opp_n = loc == opp - 9; /* :278-281 */
    int opp_s = loc == opp + 9;
    int opp_e = loc == opp - 1;
    int opp_w = loc == opp + 1;
    int row = py_div(loc, 9);   /* :283 */
    if (qzo_get_intersections(inter, loc, X) < 0) return QZO_INDEX_ERROR; /* :285 */
    int n = X[NW] != H_WALL && X[NE] != H_WALL && !opp_n; /* :287 */
    int s = X[SW] != H_WALL && X[SE] != H_WALL && !opp_s; /* :289 */
    int e = X[NE] != V_WALL && X[SE] != V_WALL && !opp_e; /* :291 */
    int w = X[NW] != V_WALL && X[SW] != V_WALL && !opp_w; /* :293 */
    if (n || (player == 1 && row == 8)) out[n_out++] = A_N; /* :295 */
    if (s || (player == 2 && row == 0)) out[n_out++] = A_S; /* :297 */
    if (e) out[n_out++] = A_E;                              /* :298 */
    if (w) out[n_out++] = A_W;                              /* :299 */
    if (opp_n && X[NE] != H_WALL && X[NW] != H_WALL) {      /* :301-314 */
        if (qzo_get_intersections(inter, opp, O) < 0) return QZO_INDEX_ERROR;
        if ((O[NW] != H_WALL && O[NE] != H_WALL) || (row == 7 && player == 1)) out[n_out++] = A_NN;
        if (O[NE] != V_WALL && X[NE] != V_WALL) out[n_out++] = A_NE;
        if (O[NW] != V_WALL && X[NW] != V_WALL) out[n_out++] = A_NW;
    } else if (opp_s && X[SE] != H_WALL && X[SW] != H_WALL) { /* :317-327 */
        if (qzo_get_intersections(inter, opp, O) < 0) return QZO_INDEX_ERROR;
        if ((O[SW] != H_WALL && O[SE] != H_WALL) || (row == 1 && player == 2)) out[n_out++] = A_SS;
        if (O[SE] != V_WALL && X[SE] != V_WALL) out[n_out++] = A_SE;
        if (O[SW] != V_WALL && X[SW] != V_WALL) out[n_out++] = A_SW;
    } else if (opp_e && X[SE] != V_WALL && X[NE] != V_WALL) { /* :330-339 */
        if (qzo_get_intersections(inter, opp, O) < 0) return QZO_INDEX_ERROR;
        if (O[SE] != V_WALL && O[NE] != V_WALL) out[n_out++] = A_EE;
        if (O[NE] != H_WALL) out[n_out++] = A_NE;
        if (O[SE] != H_WALL) out[n_out++] = A_SE;
    } else if (opp_w && X[SW] != V_WALL && X[NW] != V_WALL) { /* :342-351 */
        if (qzo_get_intersections(inter, opp, O) < 0) return QZO_INDEX_ERROR;
        if (O[NW] != V_WALL && O[SW] != V_WALL) out[n_out++] = A_WW;
        if (O[NW] != H_WALL) out[n_out++] = A_NW;
        if (O[SW] != H_WALL) out[n_out++] = A_SW;
    }
    return n_out;
}

static const int DELTA[12] = {9, -9, 1, -1, 18, -18, 2, -2, 10, 8, -8, -10}; /* :217-241, :493-516 */

/* quoridor.py:479-528.  FIFO + visited list exactly as written: the start tile is
 * not pre-marked, the goal test happens on generation, rows 9 / -1 are never
 * enqueued. */
int qzo_bfs_to_goal(const int8_t* inter, int target_row, int pos, int opp, int player) {
    enum { OFF = 40, SPAN = 192 };
    unsigned char visited[SPAN];
    int queue[SPAN + 4];
    int head = 0, tail = 0;
    int target_visited = 0;
    memset(visited, 0, sizeof(visited));
    queue[tail++] = pos; /* :483 */
    while (!target_visited && head < tail) { /* :486 */
        int cur = queue[head++];
        int dirs[12];
        int nd = qzo_valid_pawn_actions(inter, cur, opp, player, dirs); /* :488 */
        if (nd < 0) return nd;
        for (int i = 0; i < nd; i++) {
            int np = cur + DELTA[dirs[i]];
            int nrow = py_div(np, 9); /* :520 */
            if (nrow == target_row) { /* :521 */
                target_visited = 1;
            } else if (!visited[np + OFF]) { /* :523 */
                visited[np + OFF] = 1;
                if (nrow != 9 && nrow != -1) queue[tail++] = np; /* :525 */
            }
        }
    }
    return target_visited;
}

/* quoridor.py:463-477 */
int qzo_blocks_path(const qzo_game* g, int ix, int orientation) {
    int8_t j[64];
    memcpy(j, g->inter, 64); /* :470 */
    j[ix] = (int8_t)orientation; /* :471 */
    int v1 = qzo_bfs_to_goal(j, 8, g->pos[1], g->pos[2], 1); /* :474 */
    if (v1 < 0) return v1;
    int v2 = qzo_bfs_to_goal(j, 0, g->pos[2], g->pos[1], 2); /* :475 */
    if (v2 < 0) return v2;
    return !(v1 && v2);
}

/* quoridor.py:432-446 */
int qzo_validate_horizontal(const qzo_game* g, int ix) {
    int column = ix % 8;
    if (g->inter[ix] != 0) return 0;
    if (column != 0 && g->inter[ix - 1] == 1) return 0;
    if (column != 7 && g->inter[ix + 1] == 1) return 0;
    int b = qzo_blocks_path(g, ix, H_WALL);
    if (b < 0) return b;
    return !b;
}

/* quoridor.py:448-461 */
int qzo_validate_vertical(const qzo_game* g, int ix) {
    int row = ix / 8;
    if (g->inter[ix] != 0) return 0;
    if (row != 0 && g->inter[ix - 8] == -1) return 0;
    if (row != 7 && g->inter[ix + 8] == -1) return 0;
    int b = qzo_blocks_path(g, ix, V_WALL);
    if (b < 0) return b;
    return !b;
}

/* quoridor.py:420-430 */
int qzo_valid_wall_actions(const qzo_game* g, int* out) {
    int n = 0;
    for (int ix = 0; ix < 64; ix++) {
        int h = qzo_validate_horizontal(g, ix);
        if (h < 0) return h;
        if (h) out[n++] = ix;
        int v = qzo_validate_vertical(g, ix);
        if (v < 0) return v;
        if (v) out[n++] = ix + 64;
    }
    return n;
}

/* quoridor.py:138-157 */
int qzo_actions(const qzo_game* g, int* out) {
    int player = g->cur;
    int opponent = player == 2 ? 1 : 2; /* :142 */
    int n = qzo_valid_pawn_actions(g->inter, g->pos[player], g->pos[opponent], player, out);
    if (n < 0) return n;
    if ((g->cur == 1 && g->walls[1] > 0) || (g->cur == 2 && g->walls[2] > 0)) { /* :149-150 */
        int w[128];
        int nw = qzo_valid_wall_actions(g, w);
        if (nw < 0) return nw;
        for (int i = 0; i < nw; i++) out[n++] = w[i] + 12; /* :154 */
    }
    return n;
}

int qzo_actions_mask(const qzo_game* g, uint32_t mask5[5]) {
    int acts[QZO_MAX_LEGAL];
    int n = qzo_actions(g, acts);
    memset(mask5, 0, 5 * sizeof(uint32_t));
    if (n < 0) return n;
    for (int i = 0; i < n; i++) mask5[acts[i] >> 5] |= 1u << (acts[i] & 31);
    return n;
}

/* quoridor.py:193-202 -- player 2 is tested first */
int qzo_has_a_winner(const qzo_game* g, int* winner) {
    if (g->pos[2] < 9) {
        *winner = 2;
        return 1;
    }
    if (g->pos[1] > 71) {
        *winner = 1;
        return 1;
    }
    *winner = 0;
    return 0;
}

/* quoridor.py:159-186.  The reference first recomputes self.actions() (:165) and
 * only uses it when safe=True; that call is dropped here (no state change). */
int qzo_step(qzo_game* g, int action) {
    int player = g->cur;
    if (action < 0) return QZO_VALUE_ERROR;
    if (action < 12) { /* :171-172, :217-243 */
        g->pos[player] += DELTA[action];
    } else {           /* :174, :246-257 */
        int a = action - 12;
        if (a >= 128) return QZO_INDEX_ERROR;
        if (a < 64)
            g->inter[a] = 1;
        else
            g->inter[a - 64] = -1;
        if (g->cur == 1)
            g->walls[1] -= 1;
        else
            g->walls[2] -= 1;
    }
    int winner;
    if (qzo_has_a_winner(g, &winner)) return 1; /* :176-179: no rotation on a terminal move */
    if (g->cur == 1) {                          /* :260-269 */
        g->cur = 2;
        g->last = 1;
    } else {
        g->cur = 1;
        g->last = 2;
    }
    return 0;
}

/* quoridor.py:58-131.  Plane order: no_walls, vertical, horizontal, mover pawn,
 * opponent pawn, mover walls one-hot x10, opponent walls one-hot x10, turn plane.
 * walls_remaining-1 is used as a Python index, so 0 remaining -> index -1 -> plane 9
 * (:79-80).  Pawn planes index tiles[pos] the same way (negative wraps, >80 raises;
 * never evaluated on terminal boards by the build). */
void qzo_state(const qzo_game* g, double* planes) {
    memset(planes, 0, sizeof(double) * QZO_STATE_SIZE);
    for (int ix = 0; ix < 64; ix++) { /* :83-102: 8x8 padded to 9x9 on the high side */
        int r = ix / 8, c = ix % 8;
        int cell = r * 9 + c;
        if (g->inter[ix] == 0) planes[0 * 81 + cell] = 1.0;
        if (g->inter[ix] == -1) planes[1 * 81 + cell] = 1.0;
        if (g->inter[ix] == 1) planes[2 * 81 + cell] = 1.0;
    }
    int me = g->cur, other = g->cur == 1 ? 2 : 1; /* :105-129 */
    int pm = g->pos[me], po = g->pos[other];
    if (pm < 0) pm += 81;
    if (po < 0) po += 81;
    if (pm >= 0 && pm < 81) planes[3 * 81 + pm] = 1.0;
    if (po >= 0 && po < 81) planes[4 * 81 + po] = 1.0;
    int im = g->walls[me] - 1, io = g->walls[other] - 1;
    if (im < 0) im += 10;
    if (io < 0) io += 10;
    for (int i = 0; i < 81; i++) {
        planes[(5 + im) * 81 + i] = 1.0;
        planes[(15 + io) * 81 + i] = 1.0;
        if (g->cur == 2) planes[25 * 81 + i] = 1.0;
    }
}

/* ---------------- packed boards ---------------- */

void qzo_pack(const qzo_game* g, qzo_packed* p) {
    memset(p, 0, sizeof(*p));
    for (int ix = 0; ix < 64; ix++) {
        if (g->inter[ix] == 1) p->hbits |= 1ull << ix;
        if (g->inter[ix] == -1) p->vbits |= 1ull << ix;
    }
    p->p1 = (int8_t)g->pos[1];
    p->p2 = (int8_t)g->pos[2];
    p->w1 = (uint8_t)g->walls[1];
    p->w2 = (uint8_t)g->walls[2];
    p->cur = (uint8_t)g->cur;
}

void qzo_unpack(const qzo_packed* p, qzo_game* g) {
    memset(g, 0, sizeof(*g));
    for (int ix = 0; ix < 64; ix++) {
        if ((p->hbits >> ix) & 1) g->inter[ix] = 1;
        if ((p->vbits >> ix) & 1) g->inter[ix] = -1;
    }
    g->pos[1] = p->p1;
    g->pos[2] = p->p2;
    g->walls[1] = p->w1;
    g->walls[2] = p->w2;
    g->cur = p->cur;
    g->last = p->cur == 1 ? 2 : 1;
}

void qzo_movegen_batch(const qzo_packed* b, int n, uint32_t* mask5, int32_t* status) {
    for (int i = 0; i < n; i++) {
        qzo_game g;
        qzo_unpack(&b[i], &g);
        int s = qzo_actions_mask(&g, mask5 + 5 * (long)i);
        if (status) status[i] = s;
    }
}

void qzo_encode_batch_f32(const qzo_packed* b, int n, float* planes) {
    double tmp[QZO_STATE_SIZE];
    for (int i = 0; i < n; i++) {
        qzo_game g;
        qzo_unpack(&b[i], &g);
        qzo_state(&g, tmp);
        float* o = planes + (long)i * QZO_STATE_SIZE;
        for (int k = 0; k < QZO_STATE_SIZE; k++) o[k] = (float)tmp[k];
    }
}

void qzo_step_batch(qzo_packed* b, const uint8_t* action, int n, uint8_t* done, uint8_t* winner) {
    for (int i = 0; i < n; i++) {
        qzo_game g;
        qzo_unpack(&b[i], &g);
        int d = qzo_step(&g, action[i]);
        int w = 0;
        qzo_has_a_winner(&g, &w);
        qzo_pack(&g, &b[i]);
        if (done) done[i] = (uint8_t)(d > 0);
        if (winner) winner[i] = (uint8_t)w;
    }
}

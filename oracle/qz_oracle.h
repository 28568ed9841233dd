/*
 * qz_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the self-play hot path of cryer/AlphaZero_Quoridor
 * (reference: quoridor.py, mcts.py).  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library -- and only as the checker.
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here against
 * tests/golden/ (npz fixtures), which tests/golden/gen_golden.py produced by importing
 * the real Python reference in the build container.
 */
#ifndef QZ_ORACLE_H
#define QZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QZO_N_ACTIONS 140
#define QZO_MAX_LEGAL 140
#define QZO_STATE_SIZE (26 * 81)

/* Mirror of the mutable fields of quoridor.Quoridor (quoridor.py:26-56). */
typedef struct {
    int8_t inter[64]; /* _intersections: +1 horizontal, -1 vertical, 0 empty (quoridor.py:49-53) */
    int32_t pos[3];   /* _positions[1], _positions[2]; may leave 0..80 (off-board jumps) */
    int32_t walls[3]; /* _player{1,2}_walls_remaining (quoridor.py:55-56) */
    int32_t cur;      /* current_player (quoridor.py:27) */
    int32_t last;     /* last_player    (quoridor.py:28) */
} qzo_game;

/* error codes: the reference raises IndexError on out-of-range tile lookups */
#define QZO_OK 0
#define QZO_INDEX_ERROR (-1)
#define QZO_VALUE_ERROR (-2)

void qzo_reset(qzo_game* g);
/* quoridor.py:356-418; out = {NW, NE, SE, SW}; returns QZO_OK / QZO_INDEX_ERROR */
int qzo_get_intersections(const int8_t* inter, int tile, int out[4]);
/* quoridor.py:272-353; ordered codes 0..11 into out (cap 12); returns count or <0 */
int qzo_valid_pawn_actions(const int8_t* inter, int loc, int opp, int player, int* out);
/* quoridor.py:479-528; returns 1/0 or <0 */
int qzo_bfs_to_goal(const int8_t* inter, int target_row, int pos, int opp, int player);
/* quoridor.py:463-477 */
int qzo_blocks_path(const qzo_game* g, int ix, int orientation);
/* quoridor.py:432-446 / 448-461 */
int qzo_validate_horizontal(const qzo_game* g, int ix);
int qzo_validate_vertical(const qzo_game* g, int ix);
/* quoridor.py:420-430; interleaved ix / ix+64, cap 128 */
int qzo_valid_wall_actions(const qzo_game* g, int* out);
/* quoridor.py:138-157; ordered action list, cap 140; returns count or <0 */
int qzo_actions(const qzo_game* g, int* out);
/* 140-bit mask (5 x u32, bit a of word a/32) of qzo_actions; returns count or <0 */
int qzo_actions_mask(const qzo_game* g, uint32_t mask5[5]);
/* quoridor.py:159-186 without the wasted actions() call; returns done (0/1) or <0 */
int qzo_step(qzo_game* g, int action);
/* quoridor.py:193-202; returns game_over, *winner = 1|2|0(None) */
int qzo_has_a_winner(const qzo_game* g, int* winner);
/* quoridor.py:58-131; planes[26*81] float64 */
void qzo_state(const qzo_game* g, double* planes);

/* ---- batched helpers over the packed 24-byte board (see include/qz_abi.h) ---- */
typedef struct {
    uint64_t hbits; /* bit ix set <=> inter[ix] == +1 */
    uint64_t vbits; /* bit ix set <=> inter[ix] == -1 */
    int8_t p1, p2;  /* pawn tiles (signed: off-board wins) */
    uint8_t w1, w2; /* walls remaining */
    uint8_t cur;    /* 1 | 2 */
    uint8_t pad[3];
} qzo_packed;

void qzo_pack(const qzo_game* g, qzo_packed* p);
void qzo_unpack(const qzo_packed* p, qzo_game* g);
/* mask5[n][5]; status[n] = count or <0 */
void qzo_movegen_batch(const qzo_packed* b, int n, uint32_t* mask5, int32_t* status);
void qzo_encode_batch_f32(const qzo_packed* b, int n, float* planes);
void qzo_step_batch(qzo_packed* b, const uint8_t* action, int n, uint8_t* done, uint8_t* winner);

/* ---------------- MCTS (mcts.py) ---------------- */

/* policy callback == policy_value_fn contract (policy_value_net.py:145-164):
 * fills acts/probs (float32, as np.exp(float32) yields) for the legal moves IN
 * actions() ORDER, *value, returns count (>=0) or <0 on error.
 * `legal`/`n_legal` = qzo_actions(g) computed by the caller (the reference's
 * policy calls game.actions() itself: policy_value_net.py:150). */
typedef int (*qzo_policy_fn)(void* ctx, const qzo_game* g, const int* legal, int n_legal,
                             int* acts, float* probs, double* value);

typedef struct qzo_node qzo_node;
typedef struct qzo_mcts qzo_mcts;

qzo_mcts* qzo_mcts_create(qzo_policy_fn fn, void* ctx, double c_puct, int n_playout,
                          int fix_terminal_sign);
void qzo_mcts_destroy(qzo_mcts* m);
/* mcts.py:103-127 on a private copy of g (the caller's deepcopy, mcts.py:136) */
int qzo_mcts_playout(qzo_mcts* m, const qzo_game* g);
/* mcts.py:129-144: runs n_playout playouts then fills acts/visits/probs in child
 * insertion order; returns number of root children or <0 */
int qzo_mcts_get_move_probs(qzo_mcts* m, const qzo_game* g, double temp, int* acts,
                            int* visits, double* probs);
/* mcts.py:146-151 */
void qzo_mcts_update_with_move(qzo_mcts* m, int last_move);
/* introspection for tests */
int qzo_mcts_root_visits(const qzo_mcts* m);
int qzo_mcts_root_children(const qzo_mcts* m, int* acts, int* visits, double* q, float* p);
int qzo_mcts_node_count(const qzo_mcts* m);
int qzo_mcts_max_depth(const qzo_mcts* m);
long qzo_mcts_policy_calls(const qzo_mcts* m);

/* built-in stub policies (same formulas as tests/golden/gen_golden.py) */
int qzo_policy_uniform(void* ctx, const qzo_game* g, const int* legal, int n_legal, int* acts,
                       float* probs, double* value);
int qzo_policy_hash(void* ctx, const qzo_game* g, const int* legal, int n_legal, int* acts,
                    float* probs, double* value);
/* the hash the stub uses; exported so Python can re-state it */
uint32_t qzo_state_hash(const qzo_game* g);

#ifdef __cplusplus
}
#endif
#endif

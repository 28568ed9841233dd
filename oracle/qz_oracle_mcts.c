/*
 * qz_oracle_mcts.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Pointer-based, one-playout-at-a-time restatement of mcts.py (TreeNode, MCTS).
 * Arithmetic types follow the reference as written for PyTorch 0.3 / a policy
 * that returns np.float32 priors and a Python-float value:
 *   P        float32                        (policy_value_net.py:155,162)
 *   c_puct*P float32 product                (mcts.py:69; int * np.float32)
 *   u, Q     float64                        (np.sqrt(int) is float64; Q is a Python float)
 * Parity: pinned against tests/golden/ (mcts npz fixtures) (real reference + stub policies).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "qz_oracle.h"

struct qzo_node {
    qzo_node* parent;     /* mcts.py:20 */
    int n_children;       /* mcts.py:21 -- dict kept as insertion-ordered arrays */
    int* child_act;
    qzo_node** child;
    int n_visits;         /* mcts.py:22 */
    double Q;             /* mcts.py:23 */
    double u;             /* mcts.py:24 */
    float P;              /* mcts.py:25 */
};

struct qzo_mcts {
    qzo_node* root;
    qzo_policy_fn fn;
    void* ctx;
    double c_puct;
    int n_playout;
    int fix_terminal_sign;
    long policy_calls;
};

static qzo_node* node_new(qzo_node* parent, float prior) {
    qzo_node* n = (qzo_node*)calloc(1, sizeof(qzo_node));
    n->parent = parent;
    n->P = prior;
    return n;
}

static void node_free(qzo_node* n) {
    if (!n) return;
    for (int i = 0; i < n->n_children; i++) node_free(n->child[i]);
    free(n->child_act);
    free(n->child);
    free(n);
}

/* mcts.py:27-35 */
static void node_expand(qzo_node* n, const int* acts, const float* probs, int k) {
    if (k <= 0) return;
    if (n->n_children == 0) {
        n->child_act = (int*)malloc(sizeof(int) * k);
        n->child = (qzo_node**)malloc(sizeof(qzo_node*) * k);
    } else {
        n->child_act = (int*)realloc(n->child_act, sizeof(int) * (n->n_children + k));
        n->child = (qzo_node**)realloc(n->child, sizeof(qzo_node*) * (n->n_children + k));
    }
    for (int i = 0; i < k; i++) {
        int dup = 0;
        for (int j = 0; j < n->n_children; j++)
            if (n->child_act[j] == acts[i]) dup = 1; /* :34 `if action not in self._children` */
        if (dup) continue;
        n->child_act[n->n_children] = acts[i];
        n->child[n->n_children] = node_new(n, probs[i]);
        n->n_children++;
    }
}

/* mcts.py:64-70 */
static double node_get_value(qzo_node* n, double c_puct) {
    float cp = (float)c_puct * n->P;                          /* c_puct * self._P : float32 */
    n->u = (double)cp * sqrt((double)n->parent->n_visits) / (double)(1 + n->n_visits);
    return n->Q + n->u;
}

/* mcts.py:37-42 -- Python max(): first maximal element in insertion order */
static int node_select(qzo_node* n, double c_puct) {
    int best = 0;
    double bv = node_get_value(n->child[0], c_puct);
    for (int i = 1; i < n->n_children; i++) {
        double v = node_get_value(n->child[i], c_puct);
        if (v > bv) {
            bv = v;
            best = i;
        }
    }
    return best;
}

/* mcts.py:44-62 */
static void node_update_recursive(qzo_node* n, double leaf_value) {
    if (n->parent) node_update_recursive(n->parent, -leaf_value);
    n->n_visits += 1;
    n->Q += 1.0 * (leaf_value - n->Q) / n->n_visits;
}

qzo_mcts* qzo_mcts_create(qzo_policy_fn fn, void* ctx, double c_puct, int n_playout,
                          int fix_terminal_sign) {
    qzo_mcts* m = (qzo_mcts*)calloc(1, sizeof(qzo_mcts));
    m->root = node_new(NULL, 1.0f); /* mcts.py:97 */
    m->fn = fn;
    m->ctx = ctx;
    m->c_puct = c_puct;
    m->n_playout = n_playout;
    m->fix_terminal_sign = fix_terminal_sign;
    return m;
}

void qzo_mcts_destroy(qzo_mcts* m) {
    if (!m) return;
    node_free(m->root);
    free(m);
}

/* mcts.py:103-127 */
int qzo_mcts_playout(qzo_mcts* m, const qzo_game* g0) {
    qzo_game g = *g0; /* the caller's deepcopy (mcts.py:136) */
    qzo_node* node = m->root;
    while (node->n_children != 0) { /* :108-113 */
        int k = node_select(node, m->c_puct);
        int a = node->child_act[k];
        node = node->child[k];
        int r = qzo_step(&g, a);
        if (r < 0) return r;
    }
    int winner = 0;
    int end = qzo_has_a_winner(&g, &winner); /* :119 */
    double leaf_value = 0.0;
    if (!end) {
        /* :117 -- the reference evaluates the policy on terminal leaves too and
         * crashes there on off-board boards; the oracle only calls
         * it on live leaves, where it is defined. */
        int legal[QZO_MAX_LEGAL], acts[QZO_MAX_LEGAL];
        float probs[QZO_MAX_LEGAL];
        int nl = qzo_actions(&g, legal);
        if (nl < 0) return nl;
        m->policy_calls++;
        int k = m->fn(m->ctx, &g, legal, nl, acts, probs, &leaf_value);
        if (k < 0) return k;
        node_expand(node, acts, probs, k); /* :121-122 */
    } else {
        /* :125 -- step() does not rotate on a terminal move (quoridor.py:176-181), so
         * current_player == winner and this is always +1.0: the winning edge is then
         * backed up as -1 (the "terminal sign bug").  fix_terminal_sign flips it. */
        leaf_value = (winner == g.cur) ? 1.0 : -1.0;
        if (m->fix_terminal_sign) leaf_value = -leaf_value;
    }
    node_update_recursive(node, -leaf_value); /* :127 */
    return 0;
}

/* mcts.py:6-9, 129-144 */
int qzo_mcts_get_move_probs(qzo_mcts* m, const qzo_game* g, double temp, int* acts, int* visits,
                            double* probs) {
    for (int i = 0; i < m->n_playout; i++) {
        int r = qzo_mcts_playout(m, g);
        if (r < 0) return r;
    }
    int k = m->root->n_children;
    if (k == 0) return 0;
    double mx = -INFINITY;
    for (int i = 0; i < k; i++) {
        acts[i] = m->root->child_act[i];
        visits[i] = m->root->child[i]->n_visits;
        probs[i] = 1.0 / temp * log((double)visits[i] + 1e-10); /* :143 */
        if (probs[i] > mx) mx = probs[i];
    }
    double sum = 0.0;
    for (int i = 0; i < k; i++) {
        probs[i] = exp(probs[i] - mx); /* :7 */
        sum += probs[i];
    }
    for (int i = 0; i < k; i++) probs[i] /= sum; /* :8 */
    return k;
}

/* mcts.py:146-151 */
void qzo_mcts_update_with_move(qzo_mcts* m, int last_move) {
    qzo_node* r = m->root;
    for (int i = 0; i < r->n_children; i++) {
        if (r->child_act[i] == last_move) {
            qzo_node* keep = r->child[i];
            r->child[i] = NULL; /* detach before freeing the rest */
            keep->parent = NULL;
            node_free(r);
            m->root = keep;
            return;
        }
    }
    node_free(r);
    m->root = node_new(NULL, 1.0f);
}

int qzo_mcts_root_visits(const qzo_mcts* m) { return m->root->n_visits; }

int qzo_mcts_root_children(const qzo_mcts* m, int* acts, int* visits, double* q, float* p) {
    int k = m->root->n_children;
    for (int i = 0; i < k; i++) {
        if (acts) acts[i] = m->root->child_act[i];
        if (visits) visits[i] = m->root->child[i]->n_visits;
        if (q) q[i] = m->root->child[i]->Q;
        if (p) p[i] = m->root->child[i]->P;
    }
    return k;
}

static void count_rec(const qzo_node* n, int depth, int* count, int* maxd) {
    if (n->n_children) (*count)++; /* expanded nodes only */
    if (depth > *maxd) *maxd = depth;
    for (int i = 0; i < n->n_children; i++)
        if (n->child[i]) count_rec(n->child[i], depth + 1, count, maxd);
}

int qzo_mcts_node_count(const qzo_mcts* m) {
    int c = 0, d = 0;
    count_rec(m->root, 0, &c, &d);
    return c;
}

int qzo_mcts_max_depth(const qzo_mcts* m) {
    int c = 0, d = 0;
    count_rec(m->root, 0, &c, &d);
    return d;
}

long qzo_mcts_policy_calls(const qzo_mcts* m) { return m->policy_calls; }

/* ---------------- stub policies ---------------- */

static uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

uint32_t qzo_state_hash(const qzo_game* g) {
    qzo_packed p;
    qzo_pack(g, &p);
    uint32_t w[6];
    w[0] = (uint32_t)(p.hbits & 0xFFFFFFFFu);
    w[1] = (uint32_t)(p.hbits >> 32);
    w[2] = (uint32_t)(p.vbits & 0xFFFFFFFFu);
    w[3] = (uint32_t)(p.vbits >> 32);
    w[4] = ((uint32_t)(uint8_t)p.p1) | ((uint32_t)(uint8_t)p.p2 << 8) | ((uint32_t)p.w1 << 16) |
           ((uint32_t)p.w2 << 24);
    w[5] = p.cur;
    uint32_t h = 0x9E3779B9u;
    for (int i = 0; i < 6; i++) h = fmix32(h ^ w[i]);
    return h;
}

/* pure_mcts.py:13-16 style: uniform priors, value 0 */
int qzo_policy_uniform(void* ctx, const qzo_game* g, const int* legal, int n_legal, int* acts,
                       float* probs, double* value) {
    (void)ctx;
    (void)g;
    for (int i = 0; i < n_legal; i++) {
        acts[i] = legal[i];
        probs[i] = (float)(1.0 / (double)n_legal);
    }
    *value = 0.0;
    return n_legal;
}

/* deterministic pseudo-random priors/value; every quantity is exactly representable in
 * float32, so Python (reference side) and C agree bit for bit */
int qzo_policy_hash(void* ctx, const qzo_game* g, const int* legal, int n_legal, int* acts,
                    float* probs, double* value) {
    (void)ctx;
    uint32_t h = qzo_state_hash(g);
    for (int i = 0; i < n_legal; i++) {
        uint32_t a = (uint32_t)legal[i];
        uint32_t r = fmix32(h ^ (a * 0x9E3779B1u + 0x7F4A7C15u));
        acts[i] = legal[i];
        probs[i] = (float)((r >> 8) + 1u) * (1.0f / 536870912.0f); /* ((r>>8)+1) * 2^-29 */
    }
    uint32_t r2 = fmix32(h ^ 0xA511E9B3u);
    *value = (double)((int32_t)(r2 >> 8) - 8388608) / 16777216.0; /* [-0.5, 0.5) */
    return n_legal;
}

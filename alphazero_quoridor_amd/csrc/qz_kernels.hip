// qz_kernels.hip -- CDNA4 (gfx950) kernels of the Quoridor self-play engine.
//
// Tree kernels (select / expand / backup / finish_move / harvest): ONE 64-lane wavefront per
// board, 4 boards per 256-thread workgroup; board scalars are wave-uniform (SGPRs), the lanes
// are the <=131 edges of a tree node (three rounds).  Cross-lane traffic is ballots, mbcnt
// ranks and shuffle reductions.
// Rules kernels (actions() + state()), chosen by batch size in qzl::movegen_encode:
//   k_wave_rules                          < 8,192 boards: one launch, a wavefront per board
//   k_pool_paths_enc + k_pool_masks_enc   pooled pipeline, where every phase maps lanes to the
//                                         unit it has many of (qz_movegen_pool.h)
//   encoder groups                        ride in both: a tile of 16 boards -> one bit stream in
//                                         LDS -> 16-byte stores
// (The first wave-per-board kernel of round 1 lives in tests/hip/ as a test-only second
// implementation; it is not part of this library.)
// No MFMA anywhere: integer / indexing work.
//
// Reference semantics: see qz_rules.h (rules) and the per-kernel comments (mcts.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "qz_rules.h"
#include "qz_movegen_pool.h"
#include "qz_device.h"

using namespace qz;

namespace {

constexpr int WPB = 4;  // waves (= boards) per workgroup
constexpr int TPB = 64 * WPB;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t rfl64(uint64_t x) {
    return (uint64_t)rfl((uint32_t)x) | ((uint64_t)rfl((uint32_t)(x >> 32)) << 32);
}
__device__ __forceinline__ uint32_t rdl(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
__device__ __forceinline__ BB bb_rdl(BB a, int l) { return BB{rdl(a.w0, l), rdl(a.w1, l), rdl(a.w2, l)}; }
// number of set bits of a ballot below this lane
__device__ __forceinline__ int rank_below(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ Board load_board(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int b) {
    return unpack(rfl64(hb[b]), rfl64(vb[b]), rfl64(meta[b]));
}

// ============================================================================ rules kernels

// ---------------------------------------------------------------------------- pooled kernels
// Quoridor.actions() + state() as two launches over an HBM scratch area (qz_movegen_pool.h):
//   k_pool_paths_enc  path groups (lane = (board, player): board context + one shortest base
//                     path per player -> scratch; long dependent chains, no LDS: latency-bound)
//                     + ~35 % of the encoder groups
//   k_pool_masks_enc  two kinds of workgroups in ONE grid so that they overlap on the CUs:
//                     * mask groups: a tile of NB boards: slot tests -> pooled work list ->
//                       floods -> 140-bit masks.  Issue-bound.
//                     * encoder groups: a tile of NBE boards -> 2,106-bit bitmaps in LDS ->
//                       26x9x9 planes with 16-byte stores.  HBM-bound, independent of the masks.
template <int NBE>
struct EncShared {
    EncCtx ec[NBE];
    uint32_t bm[(NBE + 1) * POOL_BM_WORDS];      // word-aligned bitmaps, +1 all-zero pad board
    uint32_t st[POOL_STREAM_WORDS(NBE) + 1];      // the tile as one bit stream (2,106 bits per board)
    __attribute__((aligned(16))) float tbl[16][4];  // nibble -> four floats
};
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NBE>
__device__ __forceinline__ void encoder_group(EncShared<NBE>& sm, const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                              const uint64_t* __restrict__ meta, int n, const uint8_t* __restrict__ terminal,
                                              float* __restrict__ planes, int b0, int tid, bool nt = false) {
    const int nb = (n - b0) < NBE ? (n - b0) : NBE;
    if (tid < nb) {
        Board bd = unpack(hb[b0 + tid], vb[b0 + tid], meta[b0 + tid]);
        enc_ctx_build(sm.ec[tid], bd, terminal ? (terminal[b0 + tid] != 0) : false);
    }
    if (tid >= 64 && tid < 128) {  // a 256 B table: entry e lives in banks 4e..4e+3, so the b128 reads below never conflict
        const int e = (tid - 64) >> 2, j = tid & 3;
        sm.tbl[e][j] = (float)((e >> j) & 1);
    }
    __syncthreads();
    for (int w = tid; w < (nb + 1) * POOL_BM_WORDS; w += 256) {  // lane = one 32-bit word of a bitmap
        int bd = w / POOL_BM_WORDS, k = w - bd * POOL_BM_WORDS;
        sm.bm[w] = bd < nb ? pool_bitmap_word(sm.ec[bd], k) : 0u;
    }
    __syncthreads();
    const int nsw = POOL_STREAM_WORDS(nb);
    for (int w = tid; w < nsw; w += 256) sm.st[w] = pool_stream_word(sm.bm, w);
    __syncthreads();
    // lane = 16 bytes of output.  q advances by 256, so the nibble's shift is lane-invariant and
    // its word index advances by 32: per store one LDS word read, one bit-field extract, one
    // b128 table read.
    float4* out = reinterpret_cast<float4*>(planes + (size_t)b0 * QZ_PLANES_N);
    const int nf = nb * QZ_PLANES_N, nq = nf >> 2;
    const uint32_t sh = (uint32_t)(tid & 7) << 2;
    const float4* tbl = reinterpret_cast<const float4*>(sm.tbl);
    if (nt) {  // streaming stores: the planes leave the L2 while the kernel runs instead of in its end-of-kernel write-back
#pragma unroll 4
        for (int q = tid; q < nq; q += 256)
            __builtin_nontemporal_store(reinterpret_cast<const v4f*>(tbl)[(sm.st[q >> 3] >> sh) & 15u], reinterpret_cast<v4f*>(out) + q);
    } else {
#pragma unroll 4
        for (int q = tid; q < nq; q += 256) out[q] = tbl[(sm.st[q >> 3] >> sh) & 15u];
    }
    if ((nf & 3) && tid == 0) {  // odd number of boards in the last tile: 2 floats left
        const int bit = nf - 2;
        const uint32_t two = (sm.st[bit >> 5] >> (bit & 31)) & 3u;
        planes[(size_t)b0 * QZ_PLANES_N + nf - 2] = (float)(two & 1u);
        planes[(size_t)b0 * QZ_PLANES_N + nf - 1] = (float)(two >> 1);
    }
}
template <int NBE>
__global__ __launch_bounds__(256) void k_pool_paths_enc(const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                        const uint64_t* __restrict__ meta, int n, const uint8_t* __restrict__ terminal,
                                                        PoolHand* __restrict__ hands, int n_path_groups,
                                                        float* __restrict__ planes, int detour_mode) {
    __shared__ EncShared<NBE> sm;
    const int tid = (int)threadIdx.x;
    if ((int)blockIdx.x < n_path_groups) {
        // lane = (board, player): board context + one shortest base path per player -> the board's 184-byte hand-off record
        // (qz_movegen_pool.h: path as a tile sequence, need masks, jump plan, pawn moves).
        // Long dependent chains, no LDS: latency-bound, ~1 wave per SIMD chip-wide.
        const int task = (int)blockIdx.x * 256 + tid;
        const int b = task >> 1, p = (task & 1) + 1;
        if (b < n) {
            Board bd = unpack(hb[b], vb[b], meta[b]);
            bool term = terminal ? (terminal[b] != 0) : false;
            pool_k1_hand(bd, term, p, hands[b], detour_mode);
        }
        return;
    }
    encoder_group<NBE>(sm, hb, vb, meta, n, terminal, planes, ((int)blockIdx.x - n_path_groups) * NBE, tid);
}

// ---------------------------------------------------------------------------- small batches
// k_wave_rules: for batches too small to saturate the chip the critical path matters, not the
// instruction count: ONE launch, no hand-off through HBM.  Move-generation groups give every
// board a wavefront that runs the same phase functions as the pooled pipeline on LDS-resident
// records (lanes 0/1: base paths; lane = slot: cut tests; lane = work item: floods); encoder
// groups (encoder_group) run beside them in the same grid.
#ifdef QZ_RULES_STAMPS  // diagnostic build only (tests/hip/Makefile, benchmarks/rules_stamps.py): where a searching wavefront's time goes
__device__ unsigned int g_rules_stamps[4096][16];   // per board: s_memtime at the phase boundaries (low 32 bits) + counters
__device__ unsigned long long g_rules_enc[512][2];  // per encoder tile: s_memrealtime (100 MHz) at its first and after its last instruction
__device__ unsigned long long g_rules_rt[4096][2];  // per board: s_memrealtime at the wavefront's start and end
#define QZ_RS_MARK(k) { if (lane == 0 && bw < 4096) g_rules_stamps[bw][k] = (unsigned int)__builtin_amdgcn_s_memtime(); }
#define QZ_RS_SET(k, v) { if (lane == 0 && bw < 4096) g_rules_stamps[bw][k] = (unsigned int)(v); }
#else
#define QZ_RS_MARK(k)
#define QZ_RS_SET(k, v)
#endif
template <int G>
struct WaveBoardShared {
    PoolBoard ctx[G];
    PathTab tab[G][2];
    uint16_t items[G * 256];
    CoopSearch cs[G == 1 ? 2 : 1];  // G == 1: the two base-path searches on nine lanes each (qz_path_rows.h)
    BB conv[2][5];                  // their pn, ps, pe, pw, last as three-word sets
};
template <int NBE, int G>
union WaveRulesShared {
    WaveBoardShared<G> w[WPB];
    EncShared<NBE> enc;
};

// G boards per wavefront: the 2G base-path searches of a wave run side by side on 2G lanes (the
// search is one long dependent chain, so it costs a wave the same whether 2 or 8 lanes are
// live), and the work items of the G boards share the flood passes.
template <int NBE, int G, bool COOP>
__global__ __launch_bounds__(256) void k_wave_rules(const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                    const uint64_t* __restrict__ meta, int n, const uint8_t* __restrict__ terminal,
                                                    uint32_t* __restrict__ mask5, float* __restrict__ planes, int n_mg_groups,
                                                    int detour_mode, const int* __restrict__ n_dev) {
    __shared__ WaveRulesShared<NBE, G> sm;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n_dev) {  // the batch is the first *n_dev boards (the engine's miss list: its length is only known on the device)
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n = nd < n ? nd : n;
    }
    if ((int)blockIdx.x < n_mg_groups) {
        const int bw = ((int)blockIdx.x * WPB + wave) * G;  // first board of this wave
        if (bw >= n) return;  // whole wave leaves; only wave-level synchronisation below
        const int ng = (n - bw) < G ? (n - bw) : G;
#ifdef QZ_RULES_STAMPS
        if (lane == 0 && bw < 4096) g_rules_rt[bw][0] = __builtin_amdgcn_s_memrealtime();
        QZ_RS_SET(15, 0)
        QZ_RS_MARK(0)
#endif
        if (G == 1) {
            // wave-uniform short cut: a terminal board has no moves, a mover without walls only pawn
            // moves -- one lane, no records, no work list
            const Board bd = unpack(hb[bw], vb[bw], meta[bw]);
            const bool term = terminal ? (terminal[bw] != 0) : false;
            if (term || (bd.cur == 1 ? bd.w1 : bd.w2) <= 0) {
                if (lane == 0) {
                    const int loc = bd.cur == 1 ? bd.p1 : bd.p2, opp = bd.cur == 1 ? bd.p2 : bd.p1;
                    mask5[(size_t)bw * 5] = term ? 0u : pawn_actions_tab(bd.hb, bd.vb, loc, opp, bd.cur);
                }
                if (lane >= 1 && lane < 5) mask5[(size_t)bw * 5 + lane] = 0u;
#ifdef QZ_RULES_STAMPS
                QZ_RS_MARK(9)
                if (lane == 0 && bw < 4096) g_rules_rt[bw][1] = __builtin_amdgcn_s_memrealtime();
#endif
                return;
            }
        }
        WaveBoardShared<G>& ws = sm.w[wave];
        if (COOP && G == 1) {
            // base paths on nine lanes per player: lanes 0 / 1 build the board context and the two
            // graphs, lanes 0..8 / 9..17 search (flood + walk back, a row of the board per lane), all
            // lanes turn the rows into three-word sets, lanes 0 / 1 derive the need masks
            const Board bd = unpack(hb[bw], vb[bw], meta[bw]);  // not terminal, mover has walls (short cut above)
            QZ_RS_MARK(1)
            // board context on parallel lanes: the 2 x 12 corner values of the jump plans (lane = corner; one table
            // look-up each, all in flight together), the record's scalar fields on lane 0 in the meantime
            int c_which = 0;
            bool c_ok = false;
            uint32_t c_word = 0u;
            const int c_p = lane >= 12 ? 1 : 0, c_i = lane - 12 * c_p;
            if (lane < 24) {
                const int O = side_opp(bd, c_p + 1);
                const int grp = (int)((0x443322110000ull >> (4 * c_i)) & 15ull);  // jump_plan_corner(), its load first
                c_which = (int)((0x36E4E4u >> (2 * c_i)) & 3u);
                const int tile = O + (grp == 0 ? 0 : (grp == 1 ? -9 : (grp == 2 ? 9 : (grp == 3 ? -1 : 1))));
                c_ok = tile >= 0 && tile <= 80;
                c_word = corner_ref4(c_ok ? tile : 0);
            }
            if (lane == 0) {
                PoolBoard& out = ws.ctx[0];
                out.b = bd;
                out.flags = 1u;  // the mover has walls, the board is live (short cut above)
                for (int i = 0; i < 4; i++) out.blocked[i] = 0u;
                out.sh = static_ok_h(bd.hb, bd.vb);
                out.sv = static_ok_v(bd.hb, bd.vb);
                const int loc = bd.cur == 1 ? bd.p1 : bd.p2, opp = bd.cur == 1 ? bd.p2 : bd.p1;
                out.pawn = pawn_actions_tab(bd.hb, bd.vb, loc, opp, bd.cur);
            }
            if (lane < 24) {
                const int ref = c_ok ? (int)((c_word >> (8 * c_which)) & 0xFFu) : 64;
                JumpPlan& P = ws.ctx[0].plan[c_p];
                P.ref[c_i] = (int8_t)ref;
                P.val[c_i] = (int8_t)ref_value_fast(bd.hb, bd.vb, ref);
                if (c_i == 0) P.O = side_opp(bd, c_p + 1);
            }
            if (lane < 2) {
                CoopSearch& S = ws.cs[lane];
                S.hb = bd.hb;
                S.vb = bd.vb;
                S.plan = &ws.ctx[0].plan[lane];
                S.opp = side_opp(bd, lane + 1);
                S.start = side_start(bd, lane + 1);
                S.goal_row = lane == 0 ? 8 : 0;
            }
            wave_sync();
            QZ_RS_MARK(2)
            coop_find_path(ws.cs, 2);
            wave_sync();
            QZ_RS_MARK(3)
            const int len0 = ws.cs[0].len, len1 = ws.cs[1].len;  // wave-uniform
            for (int t = lane; t < 42; t += 64) {  // 2 searches x 5 sets x 3 words, then the four blocked sets
                if (t < 30) {
                    const int sidx = t / 15, rem = t - 15 * sidx, set = rem / 3, w = rem - 3 * set;
                    reinterpret_cast<uint32_t*>(&ws.conv[sidx][set])[w] = rows_word(ws.cs[sidx].sets[set], w);
                } else {
                    const int rem = t - 30, set = rem / 3, w = rem - 3 * set;
                    reinterpret_cast<uint32_t*>(&ws.ctx[0].base)[rem] = rows_word(ws.cs[0].sets[5 + set], w);
                }
            }
            const int n0 = (len0 > 0 ? len0 : 0) * 3, n1 = (len1 > 0 ? len1 : 0) * 3;
            for (int t = lane; t < n0 + n1; t += 64) {
                const int sidx = t < n0 ? 0 : 1, rem = t < n0 ? t : t - n0, kk = rem / 3, w = rem - 3 * kk;
                reinterpret_cast<uint32_t*>(&ws.tab[0][sidx].suffix[kk])[w] = rows_word(ws.cs[sidx].sfx[kk], w);
            }
            wave_sync();
            QZ_RS_MARK(4)
            QZ_RS_SET(10, len0)
            QZ_RS_SET(11, len1)
            if (lane < 2) {
                const CoopSearch& S = ws.cs[lane];
                K1Pre k1;
                k1.walls = true;
                k1.base = ws.ctx[0].base;  // (only the group detours read it)
                OrderedPath op;
                op.e.pn = ws.conv[lane][0];
                op.e.ps = ws.conv[lane][1];
                op.e.pe = ws.conv[lane][2];
                op.e.pw = ws.conv[lane][3];
                op.last = ws.conv[lane][4];
                op.e.found = S.found != 0;
                op.e.jump = S.jump != 0;
                op.len = S.len;
                CutMasks cm;
                cm.h = *reinterpret_cast<const uint64_t*>(S.cut_h);
                cm.v = *reinterpret_cast<const uint64_t*>(S.cut_v);
                pool_k1_post(bd, lane + 1, ws.ctx[0], k1, op, S.first_jump, S.far_jump, detour_mode & 0xFF, &cm);
            }
        } else if (lane < 2 * ng) {
            const int g = lane >> 1, b = bw + g;
            Board bd = unpack(hb[b], vb[b], meta[b]);
            const bool term = terminal ? (terminal[b] != 0) : false;
            pool_k1(bd, term, true, (lane & 1) + 1, ws.ctx[g], ws.tab[g][lane & 1], detour_mode & 0xFF);
        }
        wave_sync();
        QZ_RS_MARK(5)
        int total = 0;
        for (int g = 0; g < ng; g++) {  // wave-uniform
            const uint32_t m = pool_p2(ws.ctx[g], lane);
#pragma unroll
            for (int bit = 0; bit < 4; bit++) {
                const bool need = (m >> bit) & 1u;
                const uint64_t bal = __ballot(need);
                if (need) ws.items[total + rank_below(bal)] = (uint16_t)pool_item(g, lane, bit < 2, (bit & 1) + 1);
                total += __popcll(bal);
            }
        }
        wave_sync();
        QZ_RS_MARK(6)
        QZ_RS_SET(12, total)
        for (int base = 0; base < total; base += 64) {  // wave-uniform trip count
            const int j = base + lane;
            if (j < total) {
                const uint32_t item = ws.items[j];
                const int g = (int)(item >> 8), ix = (int)(item & 63u);
                const int side = (item & 0x80u) ? 1 : 0;
                const bool ok = (COOP && G == 1) ? pool_p3(ws.ctx[0], item, ws.cs[side].srcpos, ws.tab[0][side].suffix)
                                                 : pool_p3(ws.ctx[g], item, ws.tab[g][side]);
                if (!ok) atomicOr(&ws.ctx[g].blocked[((item & 0x40u) ? 0 : 2) + (ix >> 5)], 1u << (ix & 31));
            }
        }
        wave_sync();
        QZ_RS_MARK(7)
        if (lane < ng) {
            uint32_t m5[5];
            pool_p4(ws.ctx[lane], m5);
#pragma unroll
            for (int w = 0; w < 5; w++) mask5[(size_t)(bw + lane) * 5 + w] = m5[w];
        }
#ifdef QZ_RULES_STAMPS
        QZ_RS_MARK(8)
        QZ_RS_SET(15, 1)
        if (lane == 0 && bw < 4096) g_rules_rt[bw][1] = __builtin_amdgcn_s_memrealtime();
#endif
        return;
    }
#ifdef QZ_RULES_STAMPS
    const int et = (int)blockIdx.x - n_mg_groups;
    if (tid == 0 && et < 512) g_rules_enc[et][0] = __builtin_amdgcn_s_memrealtime();
#endif
    encoder_group<NBE>(sm.enc, hb, vb, meta, n, terminal, planes, ((int)blockIdx.x - n_mg_groups) * NBE, tid, (detour_mode & 0x100) != 0);
#ifdef QZ_RULES_STAMPS
    __syncthreads();
    if (tid == 0 && et < 512) g_rules_enc[et][1] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int NB>
struct MasksShared {
    PoolBoard ctx[NB];            // rebuilt from the hand-off records + the boards
    PoolHand hand[NB];            // the tile's hand-off records (path tile sequences: what the floods take suffix sets from)
    uint32_t srcpos[NB * 2][21];  // PathTab.srcpos of both players (84 B each), rebuilt from the sequences
    uint16_t items[NB * 256];     // every (slot, orientation, player) of every board at worst
    uint32_t n_items;
};
template <int NB, int NBE>
union MasksEncShared {
    MasksShared<NB> m;
    EncShared<NBE> enc;
};

// Second launch of the pooled pipeline: mask groups (issue-bound: slot tests, floods) and
// encoder groups (HBM-bound) are independent of each other, so they share one grid and
// overlap on the CUs.
// one mask group: the tile of NB boards from b0 (P2 slot tests -> pooled work list, P3 floods, P4 masks) from the hand-off
// records of launch 1 / the path groups
template <int NB>
__device__ __forceinline__ void mask_group(MasksShared<NB>& sm, const PoolHand* __restrict__ hands, int n, uint32_t* __restrict__ mask5,
                                           const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb, const uint64_t* __restrict__ meta,
                                           const int b0, const int tid) {
    const int lane = tid & 63;
    const int nb = (n - b0) < NB ? (n - b0) : NB;
    if (tid == 0) sm.n_items = 0u;
    {  // the tile's hand-off records into LDS (coalesced dword copy); srcpos tables cleared
        const uint32_t* src = reinterpret_cast<const uint32_t*>(hands + b0);
        uint32_t* dst = reinterpret_cast<uint32_t*>(sm.hand);
        const int nw = nb * (int)(sizeof(PoolHand) / 4);
        for (int i = tid; i < nw; i += 256) dst[i] = src[i];
        for (int i = tid; i < nb * 2 * 21; i += 256) (&sm.srcpos[0][0])[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    // lane = board: blocked sets, static slot tests from the 24-byte board; need masks, jump plans, pawn moves from the record
    if (tid < nb) pool_hand_rebuild_board(sm.ctx[tid], sm.hand[tid], unpack(hb[b0 + tid], vb[b0 + tid], meta[b0 + tid]));
    __syncthreads();
    // lane = (board, player): path edge sets, path tiles, jump positions, srcpos table from the tile sequence
    if (tid < 2 * nb) pool_hand_rebuild_path(sm.ctx[tid >> 1], (tid & 1) + 1, sm.hand[tid >> 1].seq[tid & 1], reinterpret_cast<uint8_t*>(sm.srcpos[tid]));
    __syncthreads();
    // P2: lane = (board, slot); the 64 lanes of a wave share a board
    for (int base = 0; base < nb * 64; base += 256) {
        int task = base + tid;
        int bd = task >> 6, ix = task & 63;
        uint32_t m = (task < nb * 64) ? pool_p2(sm.ctx[bd], ix) : 0u;
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            bool need = (m >> bit) & 1u;
            uint64_t bal = __ballot(need);
            if (bal != 0ull) {  // wave-uniform
                uint32_t pos = 0u;
                if (lane == 0) pos = atomicAdd(&sm.n_items, (uint32_t)__popcll(bal));
                pos = rfl(pos);
                if (need) sm.items[pos + (uint32_t)rank_below(bal)] = (uint16_t)pool_item(bd, ix, bit < 2, (bit & 1) + 1);
            }
        }
    }
    __syncthreads();
    // P3: lane = work item
    const uint32_t ni = sm.n_items;
    for (uint32_t it = (uint32_t)tid; it < ni; it += 256u) {
        uint32_t item = sm.items[it];
        int bd = (int)(item >> 8), ix = (int)(item & 63u);
        const int side = (item & 0x80u) ? 1 : 0;
        bool ok = pool_p3_seq(sm.ctx[bd], item, reinterpret_cast<const uint8_t*>(sm.srcpos[bd * 2 + side]), sm.hand[bd].seq[side]);
        if (!ok) atomicOr(&sm.ctx[bd].blocked[((item & 0x40u) ? 0 : 2) + (ix >> 5)], 1u << (ix & 31));
    }
    __syncthreads();
    // P4: legal sets -> 140-bit masks
    if (tid < nb) {
        uint32_t m5[5];
        pool_p4(sm.ctx[tid], m5);
#pragma unroll
        for (int w = 0; w < 5; w++) mask5[(size_t)(b0 + tid) * 5 + w] = m5[w];
    }
}
template <int NB, int NBE>
__global__ __launch_bounds__(256) void k_pool_masks_enc(const PoolHand* __restrict__ hands, int n,
                                                        uint32_t* __restrict__ mask5, int n_mask_groups, int enc_tile0,
                                                        const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                        const uint64_t* __restrict__ meta, const uint8_t* __restrict__ terminal,
                                                        float* __restrict__ planes) {
    __shared__ MasksEncShared<NB, NBE> smu;
    const int tid = (int)threadIdx.x;
    // grid order = dispatch order: [mask groups][encoder tiles] -- the mask groups are the launch's long pole (issue-bound, 29 KB of
    // LDS each).  Measured and rejected: encoder tiles in front of them (rounds 5: +2..17 us), a few persistent encoder workgroups
    // per CU (round 5: +3..11 us), and this launch running BESIDE the path groups' launch with its mask groups released by
    // per-path-group ready flags (round 6: +22..120 us, profiles/round6/rules/).
    const int bid = (int)blockIdx.x;
    if (bid >= n_mask_groups) {
        encoder_group<NBE>(smu.enc, hb, vb, meta, n, terminal, planes, (enc_tile0 + bid - n_mask_groups) * NBE, tid);
        return;
    }
    mask_group<NB>(smu.m, hands, n, mask5, hb, vb, meta, bid * NB, tid);
}

// Quoridor.step() + has_a_winner(): one thread per board, fully coalesced SoA traffic
__global__ __launch_bounds__(256) void k_step(uint64_t* hb, uint64_t* vb, uint64_t* meta, const uint8_t* action, int n,
                                              uint8_t* done, uint8_t* winner) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    Board b = unpack(hb[i], vb[i], meta[i]);
    int a = action[i];
    bool d = false;
    if (a < QZ_N_ACT) d = apply_action(b, a);
    hb[i] = b.hb;
    vb[i] = b.vb;
    meta[i] = pack_meta(b);
    if (done) done[i] = d ? 1 : 0;
    if (winner) winner[i] = (uint8_t)winner_of(b);
}

// ============================================================================ tree kernels

// wave-wide argmax with Python max() tie-breaking (first in order = smallest k)
__device__ __forceinline__ void wave_argmax(double& v, int& k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ov = __shfl_xor(v, off, 64);
        int ok = __shfl_xor(k, off, 64);
        if (ov > v || (ov == v && ok < k)) {
            v = ov;
            k = ok;
        }
    }
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ double wave_scan(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// ---------------------------------------------------------------------------- page pools
// Free pages live on a stack (free_list[0 .. top)).  A kernel either pops or pushes, never
// both: pops race only with pops, pushes only with pushes, so plain atomics on `top` suffice.
__device__ __forceinline__ uint32_t wave_pop(const uint32_t* free_list, int* top, int* low, int lane) {
    uint32_t page = QZ_NONE;
    if (lane == 0) {
        int old = atomicSub(top, 1);
        if (old <= 0) {
            atomicAdd(top, 1);  // empty: undo
        } else {
            atomicMin(low, old - 1);
            page = free_list[old - 1];
        }
    }
    return rfl(page);
}
// push the n pages of a table (n <= 256: four per lane at most) back; push-only kernels
__device__ __forceinline__ void wave_push(uint32_t* free_list, int* top, const uint32_t* ptab, uint32_t n, int lane) {
    if (n == 0u) return;
    int pos = 0;
    if (lane == 0) pos = atomicAdd(top, (int)n);
    pos = (int)rfl((uint32_t)pos);
    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) free_list[pos + (int)i] = ptab[i];
}
__device__ __forceinline__ void wave_free_tree_half(const EngineDev& E, int b, uint32_t half, int lane) {
    const size_t slot = tree_slot(E, b, half);
    const uint32_t n = rfl(E.tree_npages[slot]);
    wave_push(E.free_tree, E.pool_words + QZ_P_TREE_TOP, E.tree_ptab + slot * QZ_TREE_PT, n, lane);
    if (lane == 0) E.tree_npages[slot] = 0u;
}
__device__ __forceinline__ void wave_free_traj(const EngineDev& E, int b, int lane) {
    const uint32_t n = rfl(E.traj_npages[b]);
    wave_push(E.free_traj, E.pool_words + QZ_P_TRAJ_TOP, E.traj_ptab + (size_t)b * QZ_TRAJ_PT, n, lane);
    if (lane == 0) {
        E.traj_npages[b] = 0u;
        E.traj_cursor[b] = 0u;
    }
}

// Room for a block of k (1..131) consecutive edges at the end of tree T: logical offset, or
// QZ_NONE when the shared pool is empty / the page table or qz_config.edge_cap is exhausted.
// `neu` = the tree's allocation cursor, `np` = pages it maps.  A block never straddles a page;
// `mark_hole` zeroes cne in the skipped tail so that a linear scan (wave_reroot) sees no nodes there.
__device__ __forceinline__ uint32_t tree_alloc(const EngineDev& E, TreeView& T, uint32_t& neu, uint32_t& np, int k, int lane,
                                               bool mark_hole) {
    uint32_t off = neu;
    if ((off & (QZ_PAGE_EDGES - 1u)) + (uint32_t)k > QZ_PAGE_EDGES) {
        const uint32_t next = (off + QZ_PAGE_EDGES - 1u) & ~(QZ_PAGE_EDGES - 1u);
        if (mark_hole) {
            const uint32_t base = tree_phys(T, off);
            for (uint32_t i = (uint32_t)lane; i < next - off; i += 64u) T.pool[base + i].cne = 0;
        }
        off = next;
    }
    const uint32_t pg = off >> QZ_PAGE_SHIFT;
    if (pg >= (uint32_t)QZ_TREE_PT || off + (uint32_t)k > (uint32_t)E.edge_cap) return QZ_NONE;
    if (pg >= np) {
        const uint32_t page = wave_pop(E.free_tree, E.pool_words + QZ_P_TREE_TOP, E.pool_words + QZ_P_TREE_LOW, lane);
        if (page == QZ_NONE) return QZ_NONE;
        if (lane == 0) T.ptab[pg] = page;
        if (pg < 64u && lane == (int)pg) T.pt0 = page;
        if (pg >= 64u) __threadfence();  // (read back from memory by tree_phys: the store first)
        np = pg + 1u;
    }
    neu = off + (uint32_t)k;
    return off;
}

// MCTS._playout descent (mcts.py:107-113) + TreeNode.select/get_value (mcts.py:37-42, 64-70).
// The kernel lasts as long as the DEEPEST of the batch's descents, and a descent is one chain of
// dependent steps per level: edge block -> PUCT values -> argmax -> child block, ~0.4 us per level.
// Late-game lines are forced and a tree that is re-used ply after ply grows hundreds of levels
// deep, so:
//  * SPECULATIVE REPLAY.  Successive playouts of a tree share most of their path (a visit changes
//    Q + u of a well-visited node far too little to change its argmax).  Earlier descents are on
//    record, level after level, as (chosen edge, edge block) entries, and a recorded run of levels
//    is re-evaluated 64 at a time, lane = level: level i needs only its own edge block and
//    sqrt(N) of the edge chosen at level i-1 -- both in memory, so the 64 chains of loads run side
//    by side.  The longest prefix whose argmax comes out as recorded is exactly what the walk would
//    have done (same loads, same float64 expressions); the confirmed moves are applied to the
//    scratch board in one step (pawn deltas by ballot, walls one by one).  Nothing in a record is
//    taken on trust: an entry counts only if its block is the child block of the entry above.
//  * QZ_PATH_RECS RECORDS per board, each a whole root-to-leaf path.  A search alternates between
//    a deep line and its alternatives; one record would be overwritten by every shallow playout and
//    the next deep one would walk its hundreds of levels one by one.  Every edge remembers which
//    record went through it last (Edge::rid); where the path leaves the record it is following,
//    it goes on in the record of the edge it took, at the same level.  A descent that only extends
//    its record is appended to it; any other is copied over the least valuable record (oldest last
//    use, short before long: stamp + 4 x length; a pure value-by-length rule was measured worse).
//  * a round costs about two walked levels (1.75 vs 1.0 us, benchmarks/select_stamps.py), so it is
//    only tried where the record has eight more levels to offer, and after a round that ended early
//    the next eight levels are walked (in step with the record: the replay can resume at any level);
//  * records outlive the re-root: translate_records() renames them through the forwarding
//    addresses the copy leaves behind;
//  * every lane takes the float64 square root of ITS OWN edge's visit count while the division
//    is in flight: the winner's is the next level's sqrt(N_parent), off the critical path;
//  * nodes with <= 8 children (most of a long game: a mover without walls has 2-5 moves) pick
//    their maximum by a uniform scan over readlanes instead of six rounds of LDS-crossbar shuffles.
// np.sqrt of a visit count in float64 (mcts.py:69), correctly rounded: the library's sqrt(double) without the part that
// rescales subnormal arguments -- a count is 0 or >= 1 -- whose three constants the compiler kept in vector registers
// across the whole loop of k_advance.  v_rsq_f64 seeds two coupled Newton steps for sqrt(x) and 1 / (2 sqrt(x)); two
// residual corrections make the result exact to the last place (qz_selftest_sqrt checks every n < 2^20 against the host's).
__device__ __forceinline__ double sqrt_count(uint32_t n) {
    const double x = (double)n;
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return n == 0u ? 0.0 : g;
}
__device__ __forceinline__ double rdl_f64(double v, int l) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    return __longlong_as_double((long long)((uint64_t)rdl((uint32_t)u, l) | ((uint64_t)rdl((uint32_t)(u >> 32), l) << 32)));
}
// sum of v over the wavefront by DPP row shifts / row broadcasts (no LDS crossbar trip, no scalar popcount chain): the
// total arrives in lane 63 and is returned wave-uniform
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8: lane 15 of every row holds the row's sum
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// the moves of lanes 0 .. n-1 (lane j = j-th move from `bd`, movers alternate) applied at once; returns done.
// Pawn displacements and wall counts of the two movers are summed in ONE wave reduction: a lane contributes its biased
// displacement (delta + 18, 0..36) to the 12-bit field of its parity (even lanes = the current player) or 1 to its
// parity's 4-bit wall count (a player has at most 10 walls).
__device__ __forceinline__ bool apply_actions_wave(Board& bd, uint32_t act, int n, int lane) {
    const uint64_t in = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const bool mine = lane < n;
    // action_delta(a) + 18 for a = 0..11, six bits each: N S E W NN | SS EE WW NE NW | SE SW
    const uint32_t K0 = 27u | (9u << 6) | (19u << 12) | (17u << 18) | (36u << 24), K1 = 0u | (20u << 6) | (16u << 12) | (28u << 18) | (26u << 24),
                   K2 = 10u | (8u << 6);
    const uint32_t a = act < 12u ? act : 0u;
    const uint32_t tb = a < 5u ? K0 : (a < 10u ? K1 : K2), sh = 6u * (a < 5u ? a : (a < 10u ? a - 5u : a - 10u));
    const uint32_t dl = (tb >> sh) & 63u;
    const uint32_t odd = (uint32_t)lane & 1u;
    uint32_t contrib = 0u;
    if (mine) contrib = act < 12u ? (dl << (12u * odd)) : (1u << (24u + 4u * odd));
    const uint32_t tot = wave_sum_u32(contrib);
    uint64_t walls = __ballot(mine && act >= 12u);
    const int w_even = (int)((tot >> 24) & 15u), w_odd = (int)(tot >> 28);
    const int n_even = __popcll(in & 0x5555555555555555ull), n_odd = __popcll(in & 0xAAAAAAAAAAAAAAAAull);
    const int d_even = (int)(tot & 0xFFFu) - 18 * (n_even - w_even), d_odd = (int)((tot >> 12) & 0xFFFu) - 18 * (n_odd - w_odd);
    while (walls) {
        const int j = __ffsll((unsigned long long)walls) - 1;
        walls &= walls - 1ull;
        const int w = (int)rdl(act, j) - 12;
        if (w < 64) bd.hb |= 1ull << w;
        else bd.vb |= 1ull << (w - 64);
    }
    if (bd.cur == 1) {
        bd.p1 += d_even; bd.p2 += d_odd; bd.w1 -= w_even; bd.w2 -= w_odd;
    } else {
        bd.p2 += d_even; bd.p1 += d_odd; bd.w2 -= w_even; bd.w1 -= w_odd;
    }
    // only the last move can end the game (a finished position has no children): rotate n times, or n - 1
    const bool done = winner_of(bd) != 0;
    const int rot = done ? n - 1 : n;
    if (rot & 1) bd.cur = 3 - bd.cur;
    return done;
}
#ifdef QZ_SELECT_STAMPS  // diagnostic build only (tests/hip/Makefile, benchmarks/select_stamps.py): where a descent's time goes
__device__ unsigned int g_sel_stamps[64][4096][8];
#if QZ_SELECT_STAMPS >= 2  // light: counters + one stamp at each end (the per-phase stamps slow the walk)
#define QZ_SEL_MARK(acc)
#else
#define QZ_SEL_MARK(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += (unsigned int)(now_ - t_mark); t_mark = now_; }
#endif
#define QZ_SEL_COUNT(x) x
#else
#define QZ_SEL_MARK(acc)
#define QZ_SEL_COUNT(x)
#endif
// The per-board state the tree kernels work on, held in registers for a whole launch: a wavefront that runs many
// playouts of its board in one launch (k_advance) would otherwise pay a dependent memory round trip for every scalar it
// re-reads and every counter it bumps per playout (measured: 70 % of such a wavefront's cycles were s_waitcnt).
// regs_load at the start of a launch (and after a move: finish_move_board works on memory), regs_store at its end.
// LDS through address-space-3 pointers: through a generic pointer every access is a FLAT instruction, and the wait for a
// flat load is s_waitcnt vmcnt(0) lgkmcnt(0) -- which on gfx9-class hardware also drains every global STORE issued
// before it (measured in k_advance: the record commit, all LDS reads, cost 9,000 cycles per playout that way).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
struct BoardRegs {
    Board root;
    uint32_t rootN, root_ne, root_eoff, half;
    bool live;
    TreeView T;                       // page table of the current half (two registers per lane)
    uint32_t nn, neu, np;             // nodes, edge cursor, pages mapped
    uint32_t rlen;                    // lane r < QZ_PATH_RECS: descent record r's length.  (Its
                                      // last-use stamp lives in LDS, lc[LC_RSTAMP + r]: looked at once per descent, and a vector register the
                                      // 64-register build does not have -- it went to scratch, whose reload waits for every store in flight)
    uint32_t rec_last, rec_clock;
    // The per-board counters' deltas of this launch live in LDS (16 dwords per wavefront, bumped by fire-and-forget
    // ds_add / ds_max): as loop-carried scalars they cost k_advance two dozen SGPRs it does not have -- the compiler
    // parked them in VGPRs and those in scratch, whose reloads sat in the dependent chain of every playout.
    lds_u32* lc;
#ifdef QZ_ADV_STAMPS
    unsigned long long t_sel[4];  // cycles in: replay rounds, walked levels, record commit, rest of the descent
#endif
};
enum { LC_PLAYOUTS = 0, LC_TERMINAL, LC_OVERFLOW, LC_NONFINITE, LC_MAXDEPTH, LC_HITS, LC_EVALS, LC_SPARE, LC_LEVELS /*u64*/ = 8, LC_SCANNED /*u64*/ = 10,
       LC_EXPANDED /*u64*/ = 12, LC_RSTAMP = 16 /* .. 31: the descent records' last-use stamps */, LC_WORDS = 32 };
static_assert(QZ_PATH_RECS <= 16, "LC_RSTAMP holds sixteen stamps");
#ifndef QZ_LC_IN_VGPR
#define QZ_LC_IN_VGPR 1
#endif
// The counters' deltas of a launch.  Round 4: LDS words bumped by lane 0 (compare, exec save, move, ds_add, exec restore: six to
// eight instructions a time, seven times per playout of an issue-bound kernel).  Now: lane LC_LANE0 + i of the register that
// holds the record lengths in its lanes 0..15 IS counter i -- compare + select + add, no exec games, no LDS -- and the sixteen
// lanes go to the LDS words once, in regs_store (which adds them to the per-board counters in memory as before).  A launch's
// deltas fit 32 bits (<= 4,096 playouts of <= 2,048 levels): the high words of the 64-bit slots stay zero.
[[maybe_unused]] constexpr int LC_LANE0 = 32;
__device__ __forceinline__ void lc_add(BoardRegs& S, int i, uint32_t v, int lane) {
#if QZ_LC_IN_VGPR
    S.rlen += lane == LC_LANE0 + i ? v : 0u;
#else
    if (lane == 0) __hip_atomic_fetch_add(S.lc + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
}
__device__ __forceinline__ void lc_add64(BoardRegs& S, int i, unsigned long long v, int lane) {
#if QZ_LC_IN_VGPR
    S.rlen += lane == LC_LANE0 + i ? (uint32_t)v : 0u;
#else
    if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<lds_u64*>(S.lc + i), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
}
__device__ __forceinline__ void lc_max(BoardRegs& S, int i, uint32_t v, int lane) {
#if QZ_LC_IN_VGPR
    S.rlen = (lane == LC_LANE0 + i && v > S.rlen) ? v : S.rlen;
#else
    if (lane == 0) __hip_atomic_fetch_max(S.lc + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
}
// the first `cap` levels of the current descent once more in LDS (we: chosen edges, wb: blocks); nullptr: none
struct PathMirror {
    lds_u32* we;
    lds_u64* wb;
    uint32_t cap;    // 0: no mirror
    uint32_t valid;  // levels of the previous descent of THIS launch the mirror still holds (0: none): a replay of that descent's record reads them here
};
// the wave-uniform fields, forced into scalar registers (k_advance's loop: see there)
__device__ __forceinline__ void regs_uniform(BoardRegs& R) {
    R.rootN = rfl(R.rootN);
    R.root_ne = rfl(R.root_ne);
    R.root_eoff = rfl(R.root_eoff);
    R.half = rfl(R.half);
    R.nn = rfl(R.nn);
    R.neu = rfl(R.neu);
    R.np = rfl(R.np);
    R.rec_last = rfl(R.rec_last);
    R.rec_clock = rfl(R.rec_clock);
}
__device__ __forceinline__ BoardRegs regs_load(const EngineDev& E, const int b, const int lane, lds_u32* lc) {
    BoardRegs R;
    R.lc = (lds_u32*)(uintptr_t)rfl((uint32_t)(uintptr_t)lc);  // (one per wavefront: a scalar, not a vector register per lane)
#ifdef QZ_ADV_STAMPS
    R.t_sel[0] = R.t_sel[1] = R.t_sel[2] = R.t_sel[3] = 0ull;
#endif
    if (lane < LC_RSTAMP) lc[lane] = 0u;
    R.root = load_board(E.root_hb, E.root_vb, E.root_meta, b);
    R.rootN = rfl(E.root_N[b]);
    R.root_ne = rfl(E.root_ne[b]);
    R.root_eoff = rfl(E.root_eoff[b]);
    R.half = rfl(E.tree_half[b]);
    R.live = rfl(E.status[b]) == QZ_PLAYING;
    R.T = tree_view(E, b, R.half, lane);
    R.nn = rfl(E.n_nodes[b]);
    R.neu = rfl(E.n_edges[b]);
    R.np = rfl(E.tree_npages[tree_slot(E, b, R.half)]);
    R.rlen = lane < QZ_PATH_RECS ? E.rec_len[(size_t)b * QZ_PATH_RECS + lane] : 0u;
    if (lane < QZ_PATH_RECS) lc[LC_RSTAMP + lane] = E.rec_stamp[(size_t)b * QZ_PATH_RECS + lane];
    R.rec_last = rfl(E.rec_last[b]);
    R.rec_clock = rfl(E.rec_clock[b]);
    wave_sync();
    return R;
}
__device__ __forceinline__ void regs_store(const EngineDev& E, const int b, const int lane, const BoardRegs& R) {
    if (lane < QZ_PATH_RECS) {
        E.rec_len[(size_t)b * QZ_PATH_RECS + lane] = R.rlen;
        E.rec_stamp[(size_t)b * QZ_PATH_RECS + lane] = R.lc[LC_RSTAMP + lane];
    }
#if QZ_LC_IN_VGPR
    if (lane >= LC_LANE0 && lane < LC_LANE0 + LC_RSTAMP) R.lc[lane - LC_LANE0] = R.rlen;  // the counter lanes -> the LDS words lane 0 reads below
#endif
    wave_sync();
    if (lane == 0) {
        // the counters: all loads first, then all stores (one round trip, not one per counter)
        const uint32_t c0 = E.bc_playouts[b], c1 = E.bc_terminal[b], c2 = E.bc_overflow[b], c3 = E.bc_nonfinite[b], c7 = E.bc_maxdepth[b],
                       c8 = E.bc_memo_hits[b], c9 = E.bc_evals[b];
        const unsigned long long c4 = E.bc_levels[b], c5 = E.bc_scanned[b], c6 = E.bc_expanded[b];
        const lds_u32* lc = R.lc;
        const lds_u64* lc64 = reinterpret_cast<const lds_u64*>(R.lc);
        E.root_N[b] = R.rootN;
        E.root_ne[b] = R.root_ne;
        E.root_eoff[b] = R.root_eoff;
        E.n_nodes[b] = R.nn;
        E.n_edges[b] = R.neu;
        E.tree_npages[tree_slot(E, b, R.half)] = R.np;
        E.rec_last[b] = R.rec_last;
        E.rec_clock[b] = R.rec_clock;
        E.bc_maxdepth[b] = lc[LC_MAXDEPTH] > c7 ? lc[LC_MAXDEPTH] : c7;
        E.bc_playouts[b] = c0 + lc[LC_PLAYOUTS];
        E.bc_terminal[b] = c1 + lc[LC_TERMINAL];
        E.bc_overflow[b] = c2 + lc[LC_OVERFLOW];
        E.bc_nonfinite[b] = c3 + lc[LC_NONFINITE];
        E.bc_memo_hits[b] = c8 + lc[LC_HITS];
        E.bc_evals[b] = c9 + lc[LC_EVALS];
        E.bc_levels[b] = c4 + lc64[LC_LEVELS / 2];
        E.bc_scanned[b] = c5 + lc64[LC_SCANNED / 2];
        E.bc_expanded[b] = c6 + lc64[LC_EXPANDED / 2];
    }
}
// the descent of board b (MCTS._playout, mcts.py:107-113) on the register state R: no per-board scalar is read from or
// written to memory here; the leaf comes back in registers.  term: 0 live leaf; 1 terminal & winner == current_player;
// 2 terminal & winner != current_player; 3 board not playing (finished, waiting for harvest)
// at_leaf(board, finished): called once, as soon as the leaf is known -- BEFORE the descent goes on record -- so that the caller
// can put loads that only depend on the leaf (k_advance: the memo bucket) in flight under the record commit.
struct NoLeafHook {
    __device__ __forceinline__ void operator()(const Board&, bool) const {}
};
template <typename LeafHook = NoLeafHook>
__device__ __forceinline__ void select_core(EngineDev& E, BoardRegs& S, const int b, const int lane, const PathMirror PM, Board& leaf_out,
                                            uint32_t& pedge_out, uint32_t& plen_out, uint32_t& term_out, LeafHook at_leaf_hook = LeafHook()) {
#ifdef QZ_SELECT_STAMPS
    unsigned long long t_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_mark;
    unsigned int t_replay = 0, t_walk = 0, n_rounds = 0, n_narrow = 0, n_wide = 0;
#endif
    Board bd = S.root;
    uint32_t pedge = QZ_NONE;
    const uint32_t rootN = S.rootN;
    int ne = (int)S.root_ne;
    bool done = false, nonfinite = false;
    const bool live = S.live;
    uint32_t plen = 0u, scanned = 0u, replayed = 0u;
#ifdef QZ_ADV_STAMPS
    uint32_t sel_rounds = 0u, sel_failed = 0u;
    unsigned long long ts_ = __builtin_amdgcn_s_memtime();
#define QZ_TS(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.t_sel[k] += n_ - ts_; ts_ = n_; }
#else
#define QZ_TS(k)
#endif
    if (live && ne > 0) {
        constexpr uint32_t R = QZ_PATH_RECS, CAP = QZ_PATH_CAP;
        const TreeView T = S.T;
        Edge* const pool = T.pool;
        uint32_t base = tree_phys(T, S.root_eoff);
        double sq = sqrt_count(rootN);  // np.sqrt(self._parent._n_visits), float64
        uint32_t* const pe0 = E.path_edges + (size_t)b * (R + 1u) * CAP;
        unsigned long long* const pb0 = E.path_blocks + (size_t)b * (R + 1u) * CAP;
        uint32_t* const we = pe0 + (size_t)R * CAP;  // this descent (what the backup reads)
        unsigned long long* const wb = pb0 + (size_t)R * CAP;
        const bool use = (E.select_opts & 1) == 0;
#ifndef QZ_SCAN_AT_END
#define QZ_SCAN_AT_END 1
#endif
        // `scanned` (edge records looked at: a statistic, the roofline's algorithmic bytes) = the sum of the descent's nodes'
        // child counts.  Every level's block entry (base << 8 | count) is in the descent buffer when the descent ends, so with
        // the LDS mirror (k_advance) ONE pass lane = level adds them up at the leaf -- instead of four ballots + popcounts in
        // every replay round and an add per walked level, all on the scalar unit of an issue-bound kernel.
        const bool scan_at_end = QZ_SCAN_AT_END && PM.cap > 0u;
#ifndef QZ_APPLY_AT_LEAF
#define QZ_APPLY_AT_LEAF 1
#endif
        // The scratch board is only needed AT THE LEAF (the selection never looks at it), so with the LDS mirror (k_advance) the
        // descent does not replay its moves level by level (game.step(action), mcts.py:113): every block entry of the descent
        // buffer carries the level's move in its top byte, and the leaf's board is the root's + all of them in ONE wave reduction
        // (apply_actions_wave, lane = level) -- round 4 ran that reduction after every replay round and a scalar apply_action per
        // walked level, and kept the board's nine scalars alive across the whole descent of a kernel that has 78 of them.
        // Levels from QZ_PATH_CAP on have no entry: the board is brought up to date there and stepped level by level from then on.
        // INVARIANT the lazy board rests on: every entry [0, upto) it reads was WRITTEN BY THIS DESCENT (a replayed level is copied
        // from its record together with its move byte, a walked level is written with the move it chose) -- PM.valid is 0 at a
        // launch's start, so nothing of an earlier launch's buffer is ever trusted; translate_records keeps the move byte of the
        // entries it shifts, note_expansion's entry one past a record's end carries none and is never read as a level of the path.
        // Whoever lets a descent keep entries it did not write (a mirror carried across launches, say) must carry the move bytes too:
        // a stale byte gives a wrong leaf board, hence a wrong memo key, silently.  (QZ_APPLY_AT_LEAF=0 is the stepped A/B build.)
        bool lazy = QZ_APPLY_AT_LEAF && PM.cap > 0u;
        auto board_from_path = [&](const uint32_t upto) -> uint32_t {  // bd = the root's board + the moves of levels [0, upto); returns the levels' child counts summed
            wave_sync();
            uint32_t sc = 0u;
            for (uint32_t i0 = 0u; i0 < upto; i0 += 64u) {
                const uint32_t i = i0 + (uint32_t)lane;
                unsigned long long w = 0ull;
                if (i < upto) w = i < PM.cap ? PM.wb[i] : wb[i];
                const uint32_t left = upto - i0;
                done = apply_actions_wave(bd, (uint32_t)(w >> 56), (int)(left < 64u ? left : 64u), lane);
                sc += (uint32_t)(w & 0xFFull);
            }
            return wave_sum_u32(sc);
        };
        // lane r < R keeps record r's length and the time it was last useful
        if (!use && lane < (int)R) S.rlen = 0u;  // (lanes LC_LANE0.. of the same register are the launch's counters)
        uint32_t& rlen = S.rlen;      // (ONE copy: a second one lived in a register of its own across the whole descent)
        const uint32_t src = S.rec_last & (R - 1u);  // the record of the previous descent
        const uint32_t src_len = rdl(rlen, (int)src);
        // the record being followed: its levels below plen are this descent's (records are whole root-to-leaf paths and
        // the path to an edge is unique).  QZ_NONE: none -- then left_rec / left_at say which record was left last, where
        uint32_t cur = src_len > 0u ? src : QZ_NONE, cur_len = src_len;
        uint32_t left_rec = QZ_NONE, left_at = 0u;
        uint32_t used = 0u;                     // records that confirmed levels
        uint32_t walk_credit = 0u;
        bool at_leaf = false;
        while (!at_leaf) {
            // ---- replay of record cur from level plen, 64 levels per round
            bool left = false;
            bool sel_valid = false, sel_nan = false;  // a replay round's lane has already made this level's selection
            int sel_kk = 0;
            uint32_t sel_N = 0u, sel_rec = QZ_NONE;
            QZ_SEL_MARK(t_walk)
            QZ_TS(1)
#ifndef QZ_REPLAY_MIN
#define QZ_REPLAY_MIN 2   // recorded levels that must remain for a replay round to be tried (a round also selects the level after them)
#endif
#ifndef QZ_WALK_CREDIT
#define QZ_WALK_CREDIT 1  // levels walked after a round that confirmed fewer than QZ_REPLAY_MIN levels (4 / 2, 2 / 1, 1 / 1 measured 236.2 / 238.1 / 238.1 M playouts/s)
#endif
            while (cur != QZ_NONE && walk_credit == 0u && plen + (uint32_t)QZ_REPLAY_MIN <= cur_len && !left) {
                QZ_SEL_COUNT(n_rounds++;)
                const uint32_t* const pe = pe0 + (size_t)cur * CAP;
                const unsigned long long* const pb = pb0 + (size_t)cur * CAP;
                const uint32_t i = plen + (uint32_t)lane;
                // lane = level.  Levels below cur_len have a recorded choice to confirm; level cur_len -- one past the record's end --
                // may still have a BLOCK entry: the node the recorded descent ended on, written there when it was expanded
                // (note_expansion).  Such an "open" level has nothing to confirm, but its lane can do the selection.
                const bool rec = i < cur_len;
                bool ok = i <= cur_len && i < CAP;
                uint32_t lbase = 0u, chosen = QZ_NONE, prev = 0u;
                int lne = 0;
                // the record of the PREVIOUS descent of this launch is still in the LDS mirror (levels below PM.valid, not
                // yet overwritten above the current level): no memory round trip for its entries
#ifndef QZ_REPLAY_MIRROR
#define QZ_REPLAY_MIRROR 1
#endif
                const bool mir = QZ_REPLAY_MIRROR && cur == src && cur_len <= PM.valid;  // wave-uniform (PM.valid = 0 without a mirror)
                if (ok) {
                    unsigned long long w;
                    if (mir) {
                        w = i < PM.cap ? PM.wb[i] : 0ull;
                        if (rec) chosen = PM.we[i];
                        if (lane > 0) prev = PM.we[i - 1u];
                    } else {
                        w = pb[i];
                        if (rec) chosen = pe[i];
                        if (lane > 0) prev = pe[i - 1u];
                    }
                    lbase = (uint32_t)(w >> 8);
                    lne = (int)(w & 0xFFull);
                }
                // ONE round trip for the parent edge (its visit count and child block), the node's edge records and the RECORDED
                // edge's child block: the block address comes from the record, so the loads need not wait for the parent edge;
                // what the record says is only BELIEVED after the check below (a stale entry points at memory of the pool that
                // now means something else: the loads are harmless -- <= 8 records inside the pool -- and their values are thrown
                // away).  A lane without a believable entry reads record 0 of the pool and throws it away.
                // (the open level's entry may never have been written -- fresh memory, whatever it holds: its block must lie inside
                // the pool before anything is loaded from it; every other entry was written by a record commit)
                ok = ok && lne >= 1 && lne <= 8 && lbase <= (uint32_t)E.tree_pool_pages * QZ_PAGE_EDGES - 8u;
                if (!ok) {
                    lbase = 0u;
                    lne = 1;
                }
                // All of the round's loads are ISSUED TOGETHER, as raw dwords, before anything looks at them.  Of a child only
                // (Q, N, P) -- its first 16 bytes -- take part in the comparison; coff / act / cne are needed of ONE edge per
                // level, the recorded one (the level only counts if the argmax comes out as recorded), whose address the record
                // gives: 4 x 4 + 2 registers per lane instead of round 3's 8 x 6 (what set the kernel's register count).  The
                // children come four at a time: slot j is requested if ANY lane's node has more than j children (a wave-uniform
                // branch; a lane with fewer asks for its last child again: same cache line, and an equal value never replaces
                // the maximum); nodes with five to eight children take a second pass through the same registers.
                const uint32_t* const pp = reinterpret_cast<const uint32_t*>(&pool[lane > 0 ? prev : 0u]);
                const uint32_t* const cq = reinterpret_cast<const uint32_t*>(&pool[(ok && rec) ? chosen : 0u]);
                const uint32_t last_child = (uint32_t)lne - 1u;
                const bool any2 = __ballot(lne >= 2) != 0ull, any3 = __ballot(lne >= 3) != 0ull, any4 = __ballot(lne >= 4) != 0ull,
                           any5 = __ballot(lne >= 5) != 0ull;
                // (the parent edge first: what it says is needed first -- the link test, the square root -- while the children are
                // still on their way; all seven requests back to back, no branch between them: behind a branch the compiler's
                // wait for ONE of them becomes a wait for all, and a slot initialised with a copy of slot 0 waits for slot 0 --
                // measured: two dependent trips per round instead of one)
#ifndef QZ_EDGE_WIDE_LOADS
#define QZ_EDGE_WIDE_LOADS 0  // (1: the parent's three words as one 16-byte load, the backup's Q and N as 12-byte load / store: measured 2 % slower, 269.7 against 275.8 M playouts/s -- the compiler already pairs adjacent dwords and the wide forms cost four more scratch slots)
#endif
                // (every vector load instruction costs the CU's texture path one pass over the lanes' cache lines -- a gather: one
                // line per lane -- and at eight wavefronts per SIMD that path is what the kernel queues for: the parent's three
                // words come as ONE 16-byte load, the recorded edge's two as one 8-byte load)
#if QZ_EDGE_WIDE_LOADS
                const qz_u32x4_a8 pv = *reinterpret_cast<const qz_u32x4_a8*>(pp + 2);
                const qz_u32x2_a8 cv = *reinterpret_cast<const qz_u32x2_a8*>(cq + 4);
                const uint32_t pN = pv.x, pcoff = pv.z, pmisc = pv.w;
                const uint32_t ccoff = cv.x, cmisc = cv.y;
#else
                const uint32_t pN = pp[2], pcoff = pp[4], pmisc = pp[5];
                const uint32_t ccoff = cq[4], cmisc = cq[5];
#endif
                uint4 c0 = *reinterpret_cast<const uint4*>(&pool[lbase]);
                uint4 c1 = *reinterpret_cast<const uint4*>(&pool[lbase + (1u < last_child ? 1u : last_child)]);
                uint4 c2 = *reinterpret_cast<const uint4*>(&pool[lbase + (2u < last_child ? 2u : last_child)]);
                uint4 c3 = *reinterpret_cast<const uint4*>(&pool[lbase + (3u < last_child ? 3u : last_child)]);
#if QZ_EDGE_WIDE_LOADS
                // (the parent's P arrives with its load and is never looked at: left dead, its register is handed out as a temporary
                // while the load is still in flight, and the address arithmetic of the children waits for the parent -- vmcnt --
                // before it may write there: a second dependent trip.  It stays "in use" until every request is out.)
                asm volatile("" : : "v"(pv.y));
#endif
                // an entry counts only if its block IS the child block of the entry above (lane 0: the current node), with that
                // node's number of children (an entry past a record's end may be left over from another tree in the same pages)
                const uint32_t linked = tree_phys_lanes(T, pcoff);
                ok = ok && lbase == (lane > 0 ? linked : base) && lne == (lane > 0 ? (int)((pmisc >> 8) & 0xFFu) : ne);
                const double lsq = lane > 0 ? sqrt_count(pN) : sq;
                double lbest;
                uint32_t arg = 0u, lN;
#define QZ_PUCT_OF(c) (__hiloint2double((int)(c).y, (int)(c).x) + (double)(E.c_puct * __uint_as_float((c).w)) * lsq / (double)(1u + (c).z))
#define QZ_PUCT_NEXT(c, j)                                         \
    {                                                              \
        const double val_ = QZ_PUCT_OF(c);                         \
        if (val_ > lbest) { /* first maximum, like max() over the children dict */ \
            lbest = val_;                                          \
            arg = (j);                                             \
            lN = (c).z;                                            \
        }                                                          \
    }
                lbest = QZ_PUCT_OF(c0);
                lN = c0.z;
                if (any2) QZ_PUCT_NEXT(c1, 1u)
                if (any3) QZ_PUCT_NEXT(c2, 2u)
                if (any4) QZ_PUCT_NEXT(c3, 3u)
                if (any5) {  // (wave-uniform, rare in the late game: a mover without walls has two to five moves)
                    const bool any6 = __ballot(lne >= 6) != 0ull, any7 = __ballot(lne >= 7) != 0ull, any8 = __ballot(lne >= 8) != 0ull;
                    c0 = *reinterpret_cast<const uint4*>(&pool[lbase + (4u < last_child ? 4u : last_child)]);
                    if (any6) c1 = *reinterpret_cast<const uint4*>(&pool[lbase + (5u < last_child ? 5u : last_child)]);
                    if (any7) c2 = *reinterpret_cast<const uint4*>(&pool[lbase + (6u < last_child ? 6u : last_child)]);
                    if (any8) c3 = *reinterpret_cast<const uint4*>(&pool[lbase + (7u < last_child ? 7u : last_child)]);
                    // (a lane whose node has <= 4 children sees one of its first four again: equal, never greater)
                    QZ_PUCT_NEXT(c0, 4u)
                    if (any6) QZ_PUCT_NEXT(c1, 5u)
                    if (any7) QZ_PUCT_NEXT(c2, 6u)
                    if (any8) QZ_PUCT_NEXT(c3, 7u)
                }
#undef QZ_PUCT_NEXT
#undef QZ_PUCT_OF
                // ok: the lane's node IS the node of its level (given that the levels above came out as recorded), and arg is what
                // TreeNode.select picks there.  The level is CONFIRMED if that is the recorded edge.
                const bool match = ok && rec && (lbase + arg == chosen) && (lbest == lbest);
                const uint32_t lact = cmisc & 0xFFu, lcne = (cmisc >> 8) & 0xFFu;
                const uint64_t okm = __ballot(ok);
                const uint64_t bad = ~__ballot(match);
                const int nconf = bad ? (__ffsll((unsigned long long)bad) - 1) : 64;  // leading levels of this round that came out as recorded
                if (nconf > 0) {
                    // the confirmed levels into the descent buffer: its first PM.cap levels live in the LDS mirror only (a leaf
                    // that has to wait for the network writes them out: k_advance), deeper ones in memory
                    if (lane < nconf && i < CAP) {
                        if (i < PM.cap) {
                            if (!mir) {  // (replayed FROM the mirror: the entries are there)
                                PM.we[i] = chosen;
                                PM.wb[i] = ((unsigned long long)lact << 56) | ((unsigned long long)lbase << 8) | (unsigned long long)lne;  // (the record's entry -- this lane is ok -- + the move)
                            }
                        } else {
                            we[i] = chosen;
                            wb[i] = ((unsigned long long)lact << 56) | ((unsigned long long)lbase << 8) | (unsigned long long)lne;
                        }
                    }
                    used |= 1u << cur;
                    if (!lazy) done = apply_actions_wave(bd, lact, nconf, lane);
                    if (!scan_at_end) {   // edge records scanned by the confirmed levels (statistics): lne is 1..8, four ballots
                        const uint64_t inm = nconf >= 64 ? ~0ull : ((1ull << nconf) - 1ull);
                        const uint32_t l1 = (uint32_t)lne - 1u;  // 0..7
                        scanned += (uint32_t)nconf + (uint32_t)__popcll(__ballot((l1 & 1u) != 0u) & inm) + 2u * (uint32_t)__popcll(__ballot((l1 & 2u) != 0u) & inm) +
                                   4u * (uint32_t)__popcll(__ballot((l1 & 4u) != 0u) & inm);
                    }
                    plen += (uint32_t)nconf;
                    replayed += (uint32_t)nconf;
                    const int last = nconf - 1;
                    pedge = rdl(chosen, last);
                    const int cne = (int)rdl(lcne, last);
                    if (cne == 0) {  // the confirmed prefix ends on a leaf (or a finished game)
                        at_leaf = true;
                        break;
                    }
                    sq = sqrt_count(rdl(lN, last));
                    base = tree_phys(T, rdl(ccoff, last));
                    ne = cne;
                }
                // The first level that did NOT come out as recorded (or the open level past the record's end): if its lane's
                // node is right (ok), the lane has just done what the walk would do there -- the same loads, the same float64
                // expressions, first maximum wins -- so its pick IS the level's selection: the walk below only completes it
                // (the move, the record, the hint).  One level per round for free: round 3 walked it, a full dependent trip.
                if (nconf < 64 && ((okm >> nconf) & 1ull)) {
                    sel_valid = true;
                    sel_kk = (int)rdl(arg, nconf);
                    sel_N = rdl(lN, nconf);
                    sel_nan = rdl((uint32_t)!(lbest == lbest), nconf) != 0u;
                    sel_rec = rdl(chosen, nconf);  // the recorded edge of that level (QZ_NONE: the open level)
                    // (base / ne of that level: what the confirmed prefix left in base / ne -- the lane's block passed the same test)
                }
                if (nconf < 64) left = true;
                if (nconf < QZ_REPLAY_MIN) walk_credit = (uint32_t)QZ_WALK_CREDIT;
#ifdef QZ_ADV_STAMPS
                sel_rounds++;
                if (nconf < 8) sel_failed++;
#endif
            }
            QZ_SEL_MARK(t_replay)
            QZ_TS(0)
            if (at_leaf) break;
            if (E.max_depth > 0 && plen > (uint32_t)E.max_depth) break;
            if (walk_credit > 0u) walk_credit--;
            QZ_SEL_COUNT(if (ne <= 8) n_narrow++; else n_wide++;)
            // ---- one level of the walk (all lanes scan this node's children)
            uint32_t recorded = QZ_NONE;
            if (sel_valid) recorded = sel_rec;
            else if (cur != QZ_NONE && plen < cur_len) recorded = pe0[(size_t)cur * CAP + plen];  // uniform load, in flight during the scan
            if (!scan_at_end) scanned += (uint32_t)ne;
            int kk;
            uint32_t misc, w_coff;
            double w_sq;  // of the winning edge: act | cne << 8 | rid << 16, its child block, sqrt(its visit count) = the next level's sqrt(N_parent)
            if (sel_valid) {
#ifdef QZ_ADV_STAMPS
                S.t_sel[3] += 1ull;  // levels whose selection came from a replay round's lane
#endif
                // selected by the last replay round's lane (same arithmetic as below); what is left to fetch is the winner's
                // act | cne | rid and child block: 8 bytes of a record the round has just read
                kk = sel_kk;
                const uint32_t* const q = reinterpret_cast<const uint32_t*>(&pool[base + (uint32_t)kk]);
#if QZ_EDGE_WIDE_LOADS
                const qz_u32x2_a8 qv = *reinterpret_cast<const qz_u32x2_a8*>(q + 4);
                w_coff = rfl(qv.x);
                misc = rfl(qv.y);
#else
                w_coff = rfl(q[4]);
                misc = rfl(q[5]);
#endif
                w_sq = sqrt_count(sel_N);
                nonfinite = nonfinite || sel_nan;
            } else if (ne <= 8 && !(E.select_opts & 2)) {
                // a narrow node (most of a long game: a mover without walls has two to five moves): lane j < ne takes edge j -- both
                // halves of the record requested together, the square root of its own visit count beside the division (two
                // independent chains the scheduler interleaves) -- and the maximum is picked by a wave-uniform scan over
                // readlanes: first maximum wins like max() over the children dict (mcts.py:42)
                uint4 qa = make_uint4(0u, 0u, 0u, 0u);
                uint32_t qcoff = 0u, qmisc = 0u;
                if (lane < ne) {
                    const uint32_t* const q = reinterpret_cast<const uint32_t*>(&pool[base + (uint32_t)lane]);
                    qa = *reinterpret_cast<const uint4*>(q);
#if QZ_EDGE_WIDE_LOADS
                    const qz_u32x2_a8 qv = *reinterpret_cast<const qz_u32x2_a8*>(q + 4);
                    qcoff = qv.x;
                    qmisc = qv.y;
#else
                    qcoff = q[4];
                    qmisc = q[5];
#endif
                }
                const float cp = E.c_puct * __uint_as_float(qa.w);                       // c_puct * self._P in float32
                const double u = (double)cp * sq / (double)(1u + qa.z);                  // mcts.py:69
                const double val = __hiloint2double((int)qa.y, (int)qa.x) + u;           // mcts.py:70
                const double sqN = sqrt_count(qa.z);
                double bv = rdl_f64(val, 0);
                kk = 0;
                for (int j = 1; j < ne; j++) {
                    const double vj = rdl_f64(val, j);
                    if (vj > bv) {
                        bv = vj;
                        kk = j;
                    }
                }
                nonfinite = nonfinite || !(bv == bv);
                misc = rdl(qmisc, kk);
                w_coff = rdl(qcoff, kk);
                w_sq = rdl_f64(sqN, kk);
            } else {
                double best = -__builtin_inf();
                int bestk = 0x7fffffff;
                // everything the descent needs about the winning edge rides along with the
                // candidates, so the next level costs one dependent round trip, not three
                uint32_t mCOff = 0u, mMisc = 0u;
                double mSq = 0.0;
                for (int j = lane; j < ne; j += 64) {
                    // one 32-byte record per lane.  (The compiler fetches the second half -- coff, act | cne | rid -- inside the branch
                    // below, for the lanes whose candidate leads: a second trip to the cache.)
                    const uint4* q = reinterpret_cast<const uint4*>(&pool[base + (uint32_t)j]);
                    const uint4 qa = q[0], qc = q[1];
                    uint32_t N = qa.z;
                    float cp = E.c_puct * __uint_as_float(qa.w);        // c_puct * self._P in float32
                    double u = (double)cp * sq / (double)(1u + N);      // mcts.py:69
                    double val = __hiloint2double((int)qa.y, (int)qa.x) + u;  // mcts.py:70
                    double sqN = sqrt_count(N);                         // the next level's sqrt(N_parent) if this edge wins
                    // a lane's first candidate is always taken: with non-finite values (a diverged
                    // network) every comparison is false and Python's max() keeps the first child
                    if (val > best || bestk == 0x7fffffff) {
                        best = val;
                        bestk = j;
                        mSq = sqN;
                        mCOff = qc.x;
                        mMisc = qc.y;                                   // act | cne << 8 | rid << 16
                    }
                }
                wave_argmax(best, bestk);
                kk = (int)rfl((uint32_t)bestk);  // lane 0 always holds a valid pair (k = 0 is its own)
                nonfinite = nonfinite || !(best == best);
                const int wl = kk & 63;  // the winning edge is the winning lane's own best candidate
                misc = rdl(mMisc, wl);
                w_coff = rdl(mCOff, wl);
                w_sq = rdl_f64(mSq, wl);
            }
            const uint32_t e = base + (uint32_t)kk;
            const int a = (int)(misc & 0xFFu);
            if (lazy && plen >= CAP) {  // (rare: deeper than the descent buffer)
                board_from_path(CAP);
                lazy = false;
            }
            if (!lazy) done = apply_action(bd, a);  // game.step(action), mcts.py:113
            if (lane == 0 && plen < CAP) {
                const unsigned long long blk = ((unsigned long long)a << 56) | ((unsigned long long)base << 8) | (unsigned long long)(ne > 255 ? 255 : ne);
                if (plen < PM.cap) {  // (the mirror's levels reach memory only if the leaf has to wait for the network: k_advance)
                    PM.we[plen] = e;
                    PM.wb[plen] = blk;
                } else {
                    we[plen] = e;
                    wb[plen] = blk;
                }
            }
            if (use && rfl(recorded) != e && !(cur != QZ_NONE && plen >= cur_len)) {  // (beyond the end of the record it follows, a descent extends it)
                // the path leaves the record it was following (or follows none): go on in the record that took this
                // edge last, if that record still holds the edge at this level.  (Asking memory costs this level a second
                // dependent round trip; following the named record on trial instead -- the next replay round or walked level
                // shows whether it holds the path -- measured 2 % SLOWER, same-box A/B: the hint is wrong too often)
                if (cur != QZ_NONE) {
                    left_rec = cur;
                    left_at = plen;
                }
                // (A record NAMED by the edge and followed on trial -- the next replay round or walked level showing whether it
                // holds the path -- was measured 2 % slower in round 3; a generation number per record stamped into the edges, so
                // that the hint needs no question asked of memory, 3 % slower in round 4 -- 266.7 against 274.8 M playouts/s:
                // tails replaced in place keep their record's generation, and the launches' longest descents got longer.)
                const uint32_t rid = misc >> 16;
                cur = QZ_NONE;
                if (rid >= 1u && rid <= R && plen < CAP) {
                    const uint32_t rl = rdl(rlen, (int)(rid - 1u));
                    if (rl > plen && rfl(pe0[(size_t)(rid - 1u) * CAP + plen]) == e) {
                        cur = rid - 1u;
                        cur_len = rl;
                    }
                }
            }
            pedge = e;
            plen++;
            const int cne = (int)((misc >> 8) & 0xFFu);
            if (cne == 0) break;  // TreeNode.is_leaf(): never expanded (or terminal)
            if (E.max_depth > 0 && plen > (uint32_t)E.max_depth) break;  // the game is about to be dropped (select_core's caller): no point in going on
            if (plen > (uint32_t)QZ_TREE_PT * QZ_PAGE_EDGES) {  // deeper than a tree has edges: a cycle, i.e. corrupted storage.  Never hang the GPU
                if (lane == 0) atomicAdd(&E.counters[QZ_C_RUNAWAY], 1ull);
                break;
            }
            sq = w_sq;
            base = tree_phys(T, w_coff);
            ne = cne;
        }
        QZ_TS(1)
        if (lazy) {
            const uint32_t sc = board_from_path(plen);  // (plen <= CAP here)
            if (scan_at_end) scanned = sc;
        } else if (scan_at_end) {
            const uint32_t nrec = plen < CAP ? plen : CAP;
            uint32_t sc = 0u;
            for (uint32_t i = (uint32_t)lane; i < nrec; i += 64u) sc += (uint32_t)((i < PM.cap ? PM.wb[i] : wb[i]) & 0xFFull);
            scanned = wave_sum_u32(sc);
        }
        at_leaf_hook(bd, done);
        // (all wave-uniform by construction; said so explicitly, or the compiler carries them -- and the record bookkeeping
        // derived from them -- in vector registers it does not have: their scratch reloads each drain the store queue)
        cur = rfl(cur);
        cur_len = rfl(cur_len);
        left_rec = rfl(left_rec);
        left_at = rfl(left_at);
        used = rfl(used);
        plen = rfl(plen);
        // ---- put this descent on record
        if (use) {
            const uint32_t n = plen < CAP ? plen : CAP;
            const uint32_t clock = rfl(S.rec_clock) + 1u;
            uint32_t dest, from;  // levels [from, n) of this descent go into record dest, and their edges point at it
            if (cur != QZ_NONE) {
                dest = cur;  // the descent is record cur, or extends it
                from = n > cur_len ? cur_len : n;
            } else {
                // the descent left its last record at left_at and walked the rest.  In place if the new levels are at least as
                // many as the recorded ones they replace, else over the least valuable record (oldest last use, short before long)
                const uint32_t l_left = left_rec != QZ_NONE ? rdl(rlen, (int)left_rec) : 0u;
#ifndef QZ_INPLACE_SLACK
#define QZ_INPLACE_SLACK 4u  // (a new tail up to four levels shorter than the recorded one still replaces it: 0 / 1 / 4 / 16 / 64 / always measured 495 / 500 / 503 / 495 / 478 / 427 k plies/s)
#endif
                if (left_rec != QZ_NONE && n - left_at + QZ_INPLACE_SLACK >= l_left - left_at) {
                    dest = left_rec;
                    from = left_at;
                } else {
                    // lane r < R weighs record r (a long record is worth more than its age says: losing it costs a walk of its
                    // length); the minimum over the sixteen lanes by DPP row shifts, the FIRST record that has it wins (round 3: a
                    // scalar loop of sixteen readlane pairs, ~200 instructions of every descent that left its record)
#ifndef QZ_VICTIM_LEN_WEIGHT
#define QZ_VICTIM_LEN_WEIGHT 64u  // (0 / 1 / 4 / 16 / 64 / 256 / 4096 measured 490 / 491 / 504 / 516 / 517 / 516 / 515 k plies/s, and the launches' tails shorter)
#endif
                    static_assert(QZ_PATH_RECS <= 16, "the victim search reduces over one DPP row");
                    uint32_t v = 0xFFFFFFFFu;
                    if (lane < (int)R && (uint32_t)lane != left_rec && !((used >> lane) & 1u)) v = rlen == 0u ? 0u : S.lc[LC_RSTAMP + lane] + QZ_VICTIM_LEN_WEIGHT * rlen;
                    int m = (int)v;  // (unsigned minimum through a signed DPP chain: flip the sign bit)
                    m ^= (int)0x80000000;
                    {
                        int t;
                        t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x111, 0xf, 0xf, false); m = t < m ? t : m;
                        t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x112, 0xf, 0xf, false); m = t < m ? t : m;
                        t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x114, 0xf, 0xf, false); m = t < m ? t : m;
                        t = __builtin_amdgcn_update_dpp(0x7fffffff, m, 0x118, 0xf, 0xf, false); m = t < m ? t : m;
                    }
                    const uint32_t bestv = (uint32_t)__builtin_amdgcn_readlane(m, 15) ^ 0x80000000u;
                    const uint64_t who = __ballot(lane < (int)R && v == bestv);
                    dest = who ? (uint32_t)(__ffsll((unsigned long long)who) - 1) : 0u;
                    if (bestv == 0xFFFFFFFFu) dest = left_rec != QZ_NONE ? left_rec : 0u;  // every record was useful just now
                    from = 0u;
                }
            }
            dest = rfl(dest);
            from = rfl(from);
            const uint16_t stamp = (uint16_t)(dest + 1u);
            const uint32_t first = from > 0u ? from : (left_rec != QZ_NONE ? left_at : 0u);
            if (from < n) {
                wave_sync();  // lane 0 stored walked levels into the descent buffer, all lanes read it below
                uint32_t* const qe = pe0 + (size_t)dest * CAP;
                unsigned long long* const qb = pb0 + (size_t)dest * CAP;
                // (two loops with wave-uniform bounds, not one with a per-lane choice of the source: there the LDS reads had to
                // wait for every global load in flight -- the same destination registers -- among them the caller's memo probe)
                const uint32_t n_lds = n < PM.cap ? n : PM.cap;
                // The first 64 levels -- nearly always all there are: the tail below the point where the path left its record --
                // WITHOUT a loop: in front of a loop that stores, the compiler waits for every load in flight (s_waitcnt vmcnt(0)
                // in the preheader), i.e. for the caller's memo probe, issued a moment ago precisely to be in flight under these
                // stores; and the probe's own wait then counted the stores' acknowledgements as well: two dependent trips
                // where one was meant (stamps: commit 2.2 k + probe 1.6 k cycles per playout).
                {
                    const uint32_t i = from + (uint32_t)lane;
                    if (i < n_lds) {
                        const uint32_t ed = PM.we[i];
                        qe[i] = ed;
                        qb[i] = PM.wb[i];
                        if (i >= first) pool[ed].rid = stamp;
                    }
                }
                if (n > from + 64u || n > PM.cap) {  // wave-uniform, rare: a longer tail, or levels beyond the mirror
                    for (uint32_t i = from + 64u + (uint32_t)lane; i < n_lds; i += 64u) {
                        const uint32_t ed = PM.we[i];
                        qe[i] = ed;
                        qb[i] = PM.wb[i];
                        if (i >= first) pool[ed].rid = stamp;
                    }
                    for (uint32_t i = (from > PM.cap ? from : PM.cap) + (uint32_t)lane; i < n; i += 64u) {
                        const uint32_t ed = we[i];
                        qe[i] = ed;
                        qb[i] = wb[i];
                        if (i >= first) pool[ed].rid = stamp;
                    }
                }
            }
            if (lane < (int)R) {
                if ((uint32_t)lane == dest) {
                    if (from < n) rlen = n;
                    S.lc[LC_RSTAMP + lane] = clock;
                } else if ((used >> lane) & 1u) S.lc[LC_RSTAMP + lane] = clock;
            }
            S.rec_last = dest;
            S.rec_clock = clock;
        }
        QZ_TS(2)
    } else {
        // a root that is not expanded yet (the first playout of a new game or of a fresh-root restart) IS the leaf: the hook
        // must see it too -- k_advance's memo probe is issued from the hook and finished by the caller whatever the path here
        // (round 4 left the probe of such a root unissued: an indeterminate key compare)
        at_leaf_hook(bd, done);
    }
#ifdef QZ_SELECT_STAMPS
    QZ_SEL_MARK(t_walk)
    if (lane == 0 && b < 4096) {
        unsigned int* o = g_sel_stamps[E.bc_playouts[b] & 63u][b];
        o[0] = (unsigned int)(__builtin_amdgcn_s_memtime() - t_begin); o[1] = t_replay; o[2] = t_walk; o[3] = n_rounds;
        o[4] = n_narrow; o[5] = n_wide; o[6] = plen; o[7] = replayed;
    }
#endif
    uint32_t t = 0u;
    if (!live) t = 3u;
    else if (done) t = (winner_of(bd) == bd.cur) ? 1u : 2u;
    leaf_out = bd;
    pedge_out = rfl(pedge);
    plen_out = rfl(plen);
    term_out = rfl(t);
#ifdef QZ_ADV_STAMPS
    lc_add(S, LC_SPARE, replayed, lane);      // levels confirmed by replay rounds
    lc_add(S, 14, sel_rounds, lane);          // replay rounds
    lc_add(S, 15, sel_failed, lane);          // ... that confirmed fewer than 8 levels
#endif
    if (nonfinite) lc_add(S, LC_NONFINITE, 1u, lane);
    lc_add64(S, LC_SCANNED, (unsigned long long)scanned, lane);
    lc_max(S, LC_MAXDEPTH, plen, lane);
    if (plen >= 256u && lane == 0) {  // telemetry of the descents that set the kernel's duration (a handful of boards)
        atomicAdd(&E.counters[QZ_C_DEEP_DESCENTS], 1ull);
        if (2u * replayed < plen) atomicAdd(&E.counters[QZ_C_DEEP_COLD], 1ull);
        atomicAdd(&E.counters[QZ_C_DEEP_LEVELS], (unsigned long long)plen);
        atomicAdd(&E.counters[QZ_C_DEEP_REPLAYED], (unsigned long long)replayed);
    }
}
// qz_config.max_depth: the reference backs a playout up by RECURSION (TreeNode.update_recursive, mcts.py:55-62: one Python
// frame per node of the path) and never raises the interpreter's recursion limit of 1,000, so a playout whose path is
// longer than ~992 levels ends the reference's whole self-play run with a RecursionError.  A board that gets there is
// dropped (status ABORTED, counted in aborted_depth; restarted by k_release / k_round_tail) -- the one thing a batched
// engine can do that mirrors "the reference cannot play this game on".  Returns true if the board was dropped.
// a dropped game's root position, cause (the QZ_C_ABORT_* counter it is counted under), ply and board slot go into the
// engine's drop log (qz_engine_dropped_games): what the reference could not play on from.  Lane 0 only.
__device__ __forceinline__ void log_dropped_game(const EngineDev& E, const int b, const int cause) {
    const unsigned long long k = atomicAdd(&E.counters[QZ_C_DROPS_LOGGED], 1ull) % QZ_DROP_LOG;
    unsigned long long* o = E.drop_log + k * 4ull;
    o[0] = E.root_hb[b];
    o[1] = E.root_vb[b];
    o[2] = E.root_meta[b];
    o[3] = (unsigned long long)(uint32_t)cause | ((unsigned long long)E.ply[b] << 8) | ((unsigned long long)(uint32_t)b << 40);
}
__device__ __forceinline__ bool drop_if_too_deep(EngineDev& E, const int b, const int lane, const uint32_t plen) {
    if (E.max_depth <= 0 || plen <= (uint32_t)E.max_depth) return false;
    if (lane == 0) {
        E.status[b] = QZ_ABORTED;
        atomicAdd(&E.counters[QZ_C_ABORT_DEPTH], 1ull);
        log_dropped_game(E, b, QZ_C_ABORT_DEPTH);
    }
    return true;
}
// the lock-step kernels' descent: state from memory, leaf to memory (E.leaf_*: what the rules op, the evaluator and
// k_expand_backup read)
__device__ __forceinline__ void select_board(EngineDev& E, BoardRegs& S, const int b, const int lane) {
    Board bd;
    uint32_t pedge, plen, t;
    select_core(E, S, b, lane, PathMirror{(lds_u32*)nullptr, (lds_u64*)nullptr, 0u, 0u}, bd, pedge, plen, t);
    if (drop_if_too_deep(E, b, lane, plen)) t = 3u;  // (the board no longer plays: ignored by the rules op, the evaluator's output and expand_backup)
    if (lane == 0) {
        E.leaf_hb[b] = bd.hb;
        E.leaf_vb[b] = bd.vb;
        E.leaf_meta[b] = pack_meta(bd);
        E.leaf_pedge[b] = pedge;
        E.path_len[b] = plen;
        E.leaf_term[b] = (uint8_t)t;
    }
}

__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_select(EngineDev E) {
    __shared__ uint32_t s_lc[WPB][LC_WORDS];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    BoardRegs S = regs_load(E, b, lane, (lds_u32*)s_lc[wave]);
    select_board(E, S, b, lane);
    regs_store(E, b, lane, S);
}

// TreeNode.expand (mcts.py:27-35): a block of k edges in actions() order under physical edge `pedge` (QZ_NONE: the
// root), priors from prior(a) -- called by the lane that owns action a (a = lane, lane + 64, lane + 128).
// -> (first physical edge << 8 | edge count) of the new node's block, 0 if nothing was built (no legal move, arena full)
template <typename PriorFn>
__device__ __forceinline__ unsigned long long expand_node(EngineDev& E, BoardRegs& S, const int lane, const uint32_t pedge, const uint32_t m0,
                                                          const uint32_t m1, const uint32_t m2, const uint32_t m3, const uint32_t m4, PriorFn prior) {
    unsigned long long built = 0ull;
    uint32_t pawn = m0 & 0xFFFu;
    uint64_t lh = ((uint64_t)m0 >> 12) | ((uint64_t)m1 << 20) | ((uint64_t)(m2 & 0xFFFu) << 52);
    uint64_t lv = ((uint64_t)m2 >> 12) | ((uint64_t)m3 << 20) | ((uint64_t)(m4 & 0xFFFu) << 52);
    int k = __popc(pawn) + __popcll(lh) + __popcll(lv);
    if (k > 0) {
        uint32_t neu = S.neu, np = S.np;
        uint32_t off = QZ_NONE;
        if (E.node_cap <= 0 || S.nn < (uint32_t)E.node_cap) off = tree_alloc(E, S.T, neu, np, k, lane, false);
        if (off != QZ_NONE) {
            const uint32_t base = tree_phys(S.T, off);
            // (the record's constant fields from a zero the optimiser cannot see through: as literals they were hoisted out of
            // k_advance's loop into vector registers of their own -- 0.0, QZ_NONE -- and those registers spilled)
            uint32_t zero = 0u;
            asm volatile("" : "+v"(zero));
            // (a mover without walls -- most leaves of a long game -- has pawn moves only: one pass over action ids 0..11)
            const int a_end = (lh | lv) == 0ull ? 12 : QZ_N_ACT;
            for (int a = lane; a < a_end; a += 64) {
                uint32_t w = a < 32 ? m0 : (a < 64 ? m1 : (a < 96 ? m2 : (a < 128 ? m3 : m4)));
                if ((w >> (a & 31)) & 1u) {
                    uint32_t e = base + (uint32_t)order_index(pawn, lh, lv, a);
                    uint4* const q = reinterpret_cast<uint4*>(&S.T.pool[e]);
                    q[0] = make_uint4(zero, zero, zero, __float_as_uint(prior(a)));   // Q = 0.0 | N = 0 | P
                    q[1] = make_uint4(zero, (uint32_t)a | zero, pedge, ~zero);       // coff = 0 | act, cne = 0, rid = 0 | pedge | spare = QZ_NONE
                }
            }
            if (pedge != QZ_NONE) {
                if (lane == 0) {
                    S.T.pool[pedge].coff = off;
                    S.T.pool[pedge].cne = (uint8_t)k;
                }
            } else {
                S.root_eoff = off;
                S.root_ne = (uint32_t)k;
            }
            S.nn += 1u;
            S.neu = neu;
            S.np = np;
            lc_add64(S, LC_EXPANDED, (unsigned long long)k, lane);
            built = ((unsigned long long)base << 8) | (unsigned long long)(k > 255 ? 255 : k);
        } else {
            lc_add(S, LC_OVERFLOW, 1u, lane);
            S.np = np;
        }
    }
    return built;
}
// The node a descent ended on has just been expanded: its block goes into the descent's record ONE PAST the record's end
// (level plen: the record holds levels 0 .. plen-1), where the next descent's replay round finds it -- a level with nothing to
// confirm but everything its lane needs to make the selection there (select_core).  blk = expand_node's result.
__device__ __forceinline__ void note_expansion(EngineDev& E, const BoardRegs& S, const int b, const int lane, const PathMirror PM, const uint32_t plen,
                                               const unsigned long long blk) {
    if (blk == 0ull || plen >= (uint32_t)QZ_PATH_CAP || (E.select_opts & 1)) return;
    if (lane == 0) {
        E.path_blocks[((size_t)b * (QZ_PATH_RECS + 1u) + (S.rec_last & (QZ_PATH_RECS - 1u))) * QZ_PATH_CAP + plen] = blk;
        if (plen < PM.cap) PM.wb[plen] = blk;
    }
}
// node.update_recursive(-leaf_value) (mcts.py:44-62, 127): the leaf edge gets -leaf_value, its
// parent +leaf_value, ... up to the root.  The descent recorded its edges (the descent buffer `path`, its first PM.cap
// levels once more in LDS), so all levels are updated in parallel (lane = level); a path longer than the record falls
// back to walking the parent links.  term: the leaf's code (statistics only).
__device__ __forceinline__ void backup_leaf(EngineDev& E, BoardRegs& S, const int b, const int lane, const PathMirror PM, const double leaf_value,
                                            const uint32_t pedge, const uint32_t plen, const uint32_t term) {
    Edge* pool = E.edge_pool;
    if (plen <= (uint32_t)QZ_PATH_CAP) {
        const uint32_t* path = E.path_edges + ((size_t)b * (QZ_PATH_RECS + 1) + QZ_PATH_RECS) * QZ_PATH_CAP;  // the descent buffer
        for (uint32_t i = (uint32_t)lane; i < plen; i += 64u) {
            uint32_t pe;
            if (i < PM.cap) pe = PM.we[i];
            else pe = path[i];
            double val = ((plen - 1u - i) & 1u) ? leaf_value : -leaf_value;
#if QZ_EDGE_WIDE_LOADS
            // (Q and N as one 12-byte load and one 12-byte store: two passes of the texture path per level, not four)
            uint32_t* const rec = reinterpret_cast<uint32_t*>(&pool[pe]);
            const qz_u32x3_a16 old = *reinterpret_cast<const qz_u32x3_a16*>(rec);
            const uint32_t N = old.z + 1u;  // mcts.py:51
            double Q = __hiloint2double((int)old.y, (int)old.x);
            Q += 1.0 * (val - Q) / (double)N;  // mcts.py:53
            qz_u32x3_a16 neu;
            neu.x = (uint32_t)__double2loint(Q);
            neu.y = (uint32_t)__double2hiint(Q);
            neu.z = N;
            *reinterpret_cast<qz_u32x3_a16*>(rec) = neu;
#else
            uint32_t N = pool[pe].N + 1u;  // mcts.py:51
            double Q = pool[pe].Q;
            Q += 1.0 * (val - Q) / (double)N;  // mcts.py:53
            pool[pe].N = N;
            pool[pe].Q = Q;
#endif
        }
    } else if (lane == 0) {
        double val = -leaf_value;
        uint32_t pe = pedge;
        while (pe != QZ_NONE) {
            uint32_t N = pool[pe].N + 1u;
            double Q = pool[pe].Q;
            Q += 1.0 * (val - Q) / (double)N;
            pool[pe].N = N;
            pool[pe].Q = Q;
            val = -val;  // mcts.py:61
            pe = pool[pe].pedge;
        }
    }
    S.rootN += 1u;  // the root is updated too
    lc_add(S, LC_PLAYOUTS, 1u, lane);
    lc_add64(S, LC_LEVELS, (unsigned long long)plen, lane);
    if (term != 0u) lc_add(S, LC_TERMINAL, 1u, lane);
}
// mcts.py:125: +1 if winner == current_player else -1 (always +1 in practice: the reference does not rotate players
// on a terminal move); term = 1 | 2
__device__ __forceinline__ double terminal_value(const EngineDev& E, const uint32_t term) {
    double leaf_value = (term == 1u) ? 1.0 : -1.0;
    if (E.fix_terminal_sign) leaf_value = -leaf_value;
    return leaf_value;
}
// the lock-step kernels' TreeNode.expand + update_recursive of board b's current leaf (E.leaf_*: written by the last
// descent) with the caller's p [B][140] / v [B] and the legal sets of the rules op (E.leaf_mask)
__device__ __forceinline__ void expand_backup_board(EngineDev& E, BoardRegs& S, const float* __restrict__ p, const float* __restrict__ v, const int b, const int lane) {
    const uint32_t term = rfl(E.leaf_term[b]);
    if (term == 3u) return;
    const uint32_t pedge = rfl(E.leaf_pedge[b]), plen = rfl(E.path_len[b]);
    double leaf_value;
    if (term == 0u) {
        leaf_value = (double)v[b];
        const uint32_t* mask = E.leaf_mask + (size_t)b * 5;
        const float* prow = p + (size_t)b * QZ_N_ACT;
        const unsigned long long blk = expand_node(E, S, lane, pedge, rfl(mask[0]), rfl(mask[1]), rfl(mask[2]), rfl(mask[3]), rfl(mask[4]), [&](int a) { return prow[a]; });
        note_expansion(E, S, b, lane, PathMirror{(lds_u32*)nullptr, (lds_u64*)nullptr, 0u, 0u}, plen, blk);
    } else {
        leaf_value = terminal_value(E, term);
    }
    backup_leaf(E, S, b, lane, PathMirror{(lds_u32*)nullptr, (lds_u64*)nullptr, 0u, 0u}, leaf_value, pedge, plen, term);
}

__global__ __launch_bounds__(TPB) void k_expand_backup(EngineDev E, const float* __restrict__ p, const float* __restrict__ v) {
    __shared__ uint32_t s_lc[WPB][LC_WORDS];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    BoardRegs S = regs_load(E, b, lane, (lds_u32*)s_lc[wave]);
    expand_backup_board(E, S, p, v, b, lane);
    regs_store(E, b, lane, S);
}
// Playout i's expansion + backup and playout i+1's descent of the same board in ONE launch, by the same wavefront: a
// kernel boundary flushes the eight XCDs' L2s, so a descent launched on its own fetches every edge record of its chain
// from the Infinity Cache / HBM (~1 us per level); here the records the backup just touched are still in this XCD's L2.
// Same operations in the same order per board as k_expand_backup followed by k_select.
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_expand_backup_select(EngineDev E, const float* __restrict__ p, const float* __restrict__ v) {
    __shared__ uint32_t s_lc[WPB][LC_WORDS];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    BoardRegs S = regs_load(E, b, lane, (lds_u32*)s_lc[wave]);
    expand_backup_board(E, S, p, v, b, lane);
    wave_sync();  // the backup's stores (other lanes) before the descent's loads
    select_board(E, S, b, lane);
    regs_store(E, b, lane, S);
}

// softmax(1/temp * log(visits + 1e-10)) over the root's children (mcts.py:6-9, 141-144).
// Returns this lane's probabilities for edges lane, lane+64, lane+128 in pr[3].
__device__ __forceinline__ void root_pi(const Edge* __restrict__ re, int ne, double inv_temp, int lane, double pr[3]) {
    double x[3];
    double mx = -__builtin_inf();
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        x[r] = -__builtin_inf();
        if (k < ne) {
            x[r] = inv_temp * log((double)re[k].N + 1e-10);
            mx = fmax(mx, x[r]);
        }
    }
    mx = wave_max(mx);
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        pr[r] = (k < ne) ? exp(x[r] - mx) : 0.0;
        s += pr[r];
    }
    s = wave_sum(s);
#pragma unroll
    for (int r = 0; r < 3; r++) pr[r] = pr[r] / s;
}
// the root's edge block of board b (NULL if the root was never expanded)
__device__ __forceinline__ Edge* root_block(const EngineDev& E, int b, int lane, int& ne) {
    ne = (int)rfl(E.root_ne[b]);
    if (ne == 0) return nullptr;
    const TreeView T = tree_view(E, b, rfl(E.tree_half[b]), lane);
    return T.pool + tree_phys(T, rfl(E.root_eoff[b]));
}

__global__ __launch_bounds__(TPB) void k_root_pi(EngineDev E, double* __restrict__ pi, int32_t* __restrict__ visits) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    for (int a = lane; a < QZ_N_ACT; a += 64) {
        if (pi) pi[(size_t)b * QZ_N_ACT + a] = 0.0;
        if (visits) visits[(size_t)b * QZ_N_ACT + a] = -1;
    }
    int ne;
    const Edge* re = root_block(E, b, lane, ne);
    if (!re) return;
    double pr[3];
    root_pi(re, ne, 1.0 / (double)E.temp, lane, pr);
    wave_sync();
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        if (k < ne) {
            int a = re[k].act;
            if (pi) pi[(size_t)b * QZ_N_ACT + a] = pr[r];
            if (visits) visits[(size_t)b * QZ_N_ACT + a] = (int32_t)re[k].N;
        }
    }
}

__global__ __launch_bounds__(TPB) void k_root_children(EngineDev E, int32_t* visits, double* q, float* prior, int32_t* root_visits) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    for (int a = lane; a < QZ_N_ACT; a += 64) {
        if (visits) visits[(size_t)b * QZ_N_ACT + a] = -1;
        if (q) q[(size_t)b * QZ_N_ACT + a] = 0.0;
        if (prior) prior[(size_t)b * QZ_N_ACT + a] = 0.f;
    }
    if (root_visits && lane == 0) root_visits[b] = (int32_t)E.root_N[b];
    int ne;
    const Edge* re = root_block(E, b, lane, ne);
    if (!re) return;
    wave_sync();
    for (int k = lane; k < ne; k += 64) {
        int a = re[k].act;
        if (visits) visits[(size_t)b * QZ_N_ACT + a] = (int32_t)re[k].N;
        if (q) q[(size_t)b * QZ_N_ACT + a] = re[k].Q;
        if (prior) prior[(size_t)b * QZ_N_ACT + a] = re[k].P;
    }
}

// copy one node's edge block (logical s_off in S) to logical d_off in D; the copies hang under
// physical edge `pedge`.  coff / cne still name the SOURCE blocks of the children: they are
// fixed up when wave_reroot's scan reaches them.
__device__ __forceinline__ void copy_block(const TreeView& S, uint32_t s_off, const TreeView& D, uint32_t d_off, int ne,
                                           uint32_t pedge, int lane) {
    const uint32_t sb = tree_phys(S, s_off), db = tree_phys(D, d_off);
    for (int k = lane; k < ne; k += 64) {
        Edge ed = S.pool[sb + (uint32_t)k];
        ed.pedge = pedge;
        ed.spare = QZ_NONE;
        D.pool[db + (uint32_t)k] = ed;
        S.pool[sb + (uint32_t)k].spare = db + (uint32_t)k;  // forwarding address: where this edge lives now (translate_records)
    }
}

// k_select's descent records across a re-root.  The subtree below root edge `edge` has just been copied and every copied
// edge left a forwarding address behind (Edge::spare of the OLD record).  A record that went through `edge` stays valid
// one level shorter: its entries are shifted up by one and renamed through the forwarding addresses; it is cut where
// the copy was (pool exhausted).  Every other record described a subtree that is gone.  edge == QZ_NONE: all gone.
__device__ __forceinline__ void translate_records(EngineDev& E, int b, int lane, uint32_t edge, const bool in_place = false) {
    constexpr uint32_t R = QZ_PATH_RECS, CAP = QZ_PATH_CAP;
    const Edge* pool = E.edge_pool;
    wave_sync();  // the forwarding addresses were stored by other lanes
    // lane r < R looks at record r: its length and its first edge in ONE trip for all sixteen (round 3 asked record after
    // record: two dependent trips each, a third of the latency of a move); only the records that went through the move are
    // then shifted, the others are simply emptied
    const uint32_t len_l = lane < (int)R ? E.rec_len[(size_t)b * R + lane] : 0u;
    const uint32_t first_l = lane < (int)R ? E.path_edges[((size_t)b * (R + 1u) + (uint32_t)lane) * CAP] : QZ_NONE;
    uint64_t keep = __ballot(lane < (int)R && edge != QZ_NONE && len_l > 1u && first_l == edge);
    if (lane < (int)R && !((keep >> lane) & 1ull)) E.rec_len[(size_t)b * R + lane] = 0u;
    while (keep) {
        const uint32_t r = (uint32_t)(__ffsll((unsigned long long)keep) - 1);
        keep &= keep - 1ull;
        const uint32_t len = rdl(len_l, (int)r);
        uint32_t* const re = E.path_edges + ((size_t)b * (R + 1u) + r) * CAP;
        unsigned long long* const rb = E.path_blocks + ((size_t)b * (R + 1u) + r) * CAP;
        uint32_t newlen = len - 1u;
        {
            // (level len, one past the end, may hold the block of the node the recorded descent ended on -- note_expansion --:
            // it moves up with the rest.  Nothing says it is there: whatever arrives at the new index is only believed after
            // the replay round's link test, like every entry.)
            for (uint32_t c = 1u; c <= len; c += 64u) {
                const uint32_t i = c + (uint32_t)lane;
                const bool open = i == len && i < CAP;
                const bool in = i < len || open;
                uint32_t fe = QZ_NONE, fb = QZ_NONE;
                unsigned long long bo = 0ull;
                if (in) {
                    bo = rb[i];
                    const uint32_t be = (uint32_t)(bo >> 8) < (uint32_t)E.tree_pool_pages * QZ_PAGE_EDGES ? (uint32_t)(bo >> 8) : 0u;
                    if (in_place) {  // the subtree stays where it is: the entries only move up one level
                        fe = open ? 0u : re[i];
                        fb = be;
                    } else {
                        fe = open ? 0u : pool[re[i]].spare;
                        fb = pool[be].spare;
                    }
                }
                const uint64_t bad = __ballot(in && (fe == QZ_NONE || fb == QZ_NONE));
                const uint32_t nvalid = bad ? (uint32_t)(__ffsll((unsigned long long)bad) - 1) : 64u;
                if (in && (uint32_t)lane < nvalid) {
                    re[i - 1u] = fe;
                    rb[i - 1u] = ((unsigned long long)fb << 8) | (bo & 0xFF000000000000FFull);  // (the level's move -- top byte -- stays with its entry: board_from_path)
                }
                if (bad) {
                    newlen = c - 1u + nvalid;
                    break;
                }
            }
        }
        if (lane == 0) E.rec_len[(size_t)b * R + r] = newlen;
    }
}

// MCTS.update_with_move (mcts.py:146-151): keep the chosen child's subtree by copying it,
// breadth first, into fresh pages mapped by the board's other page table; `edge` is the
// physical root edge of the move or QZ_NONE for a fresh root.  The breadth-first queue is the
// copy itself: blocks are appended in discovery order, so scanning the new tree's edges in
// logical order and expanding every edge with cne > 0 visits the nodes breadth first.  The old
// pages are handed back by k_release (launched right after every kernel that re-roots).
// in_place (the asynchronous loop, while the tree's allocation cursor is below qz_config.compact_edges): the kept subtree
// is NOT copied -- the chosen child's block simply becomes the root's block where it lies (its edges lose their parent
// link), the rest of the old tree stays behind as garbage until a later move compacts (the copy below), and a child
// without a subtree makes the whole tree garbage: the cursor goes back to 0 in the pages already mapped.  Same tree as
// far as any descent, expansion or backup can tell; a move costs a few stores instead of a breadth-first copy whose
// duration (up to tens of ms for the largest trees) every other board of the launch had to wait for.
// Sliced (k_advance only: `deadline` in s_memrealtime ticks, `st` = the board's 8 words of E.compact_state): the queue of the
// breadth-first copy is the new tree itself and the old tree is not modified, so the copy can stop between two windows and go on
// in a later launch from (scan position, allocation cursor, pages mapped, node count).  A 1,000-level line copies at one level
// per memory round trip -- ~2 M cycles, measured -- and a launch ends with its LAST wave: rather than hold up 4,095 boards, the
// wave saves those words when the launch's budget is spent and returns false; the board sits out until its copy is done.
__device__ __forceinline__ bool wave_reroot(EngineDev& E, int b, int lane, uint32_t edge, const bool in_place = false,
                                            const unsigned long long deadline = 0ull, uint32_t* const st = nullptr) {
    const uint32_t half = rfl(E.tree_half[b]);
    const TreeView S = tree_view(E, b, half, lane);
    uint32_t s_off = 0u, childN = 0u;
    int s_ne = 0;
    if (edge != QZ_NONE) {
        s_ne = (int)rfl((uint32_t)S.pool[edge].cne);
        s_off = rfl(S.pool[edge].coff);
        childN = rfl(S.pool[edge].N);
    }
    if (in_place) {
        if (s_ne > 0) {
            const uint32_t base = tree_phys(S, s_off);
            for (int k = lane; k < s_ne; k += 64) S.pool[base + (uint32_t)k].pedge = QZ_NONE;
        }
        translate_records(E, b, lane, s_ne > 0 ? edge : QZ_NONE, true);
        if (lane == 0) {
            E.path_len[b] = 0u;
            E.pl_done[b] = 0u;
            E.root_N[b] = childN;
            E.root_eoff[b] = s_ne > 0 ? s_off : 0u;
            E.root_ne[b] = (uint32_t)s_ne;
            if (s_ne == 0) {
                E.n_nodes[b] = 0u;
                E.n_edges[b] = 0u;
                if (E.compact_edges > 0) E.compact_at[b] = (uint32_t)E.compact_edges;
            }
        }
        return true;
    }
    uint32_t new_nodes = 0u, new_edges = 0u, dnp = 0u, root_off = 0u, truncated = 0u;
    int root_ne = 0;
    bool flipped = false;
    if (s_ne > 0) {
        TreeView D = tree_view(E, b, half ^ 1u, lane);
        const bool resumed = st != nullptr && rfl(st[0]) != 0u;
        uint32_t off = 0u, q = 0u;
        bool exhausted = false;
        if (resumed) {
            q = rfl(st[1]);
            new_edges = rfl(st[2]);
            dnp = rfl(st[3]);
            new_nodes = rfl(st[4]);
            truncated = rfl(st[5]);
            exhausted = rfl(st[6]) != 0u;
            off = rfl(st[7]);
        } else {
            off = tree_alloc(E, D, new_edges, dnp, s_ne, lane, true);
            if (off != QZ_NONE) {
                copy_block(S, s_off, D, off, s_ne, QZ_NONE, lane);
                new_nodes = 1u;
            }
        }
        if (off != QZ_NONE) {
            root_off = off;
            root_ne = s_ne;
            flipped = true;
            const uint32_t q_first = q;
            while (q < new_edges) {  // wave-uniform
                wave_sync();
                if (st != nullptr && q > q_first && __builtin_amdgcn_s_memrealtime() > deadline) {  // (every call gets somewhere)
                    if (lane == 0) {
                        st[0] = 1u;
                        st[1] = q;
                        st[2] = new_edges;
                        st[3] = dnp;
                        st[4] = new_nodes;
                        st[5] = truncated;
                        st[6] = exhausted ? 1u : 0u;
                        st[7] = off;
                        E.tree_npages[tree_slot(E, b, half ^ 1u)] = dnp;  // (what a reset of the engine hands back)
                    }
                    return false;
                }
                const uint32_t wbase = q & ~63u;
                const uint32_t lim = new_edges < wbase + 64u ? new_edges : wbase + 64u;
                const uint32_t idx = wbase + (uint32_t)lane;
                const uint32_t pb = tree_phys(D, wbase);
                // lane = one edge of the window of the NEW tree; the expanded ones are nodes whose children's blocks are
                // still in the old tree.  Blocks are allocated one after the other (discovery order: the layout does not
                // depend on how the copies are scheduled), then copied side by side: a node with <= 8 children by ITS lane
                // (64 nodes' loads in flight together; the one-node-at-a-time form spent two memory round trips per node
                // and made a move of a 10,000-node tree last milliseconds), wider nodes by the whole wavefront.
                uint32_t cne = 0u, c_src = 0u;
                if (idx >= q && idx < lim) {
                    cne = D.pool[pb + (uint32_t)lane].cne;
                    c_src = D.pool[pb + (uint32_t)lane].coff;
                }
                uint64_t m = __ballot(cne > 0u);
                uint32_t mydoff = QZ_NONE;
                while (m) {  // the nodes found in this window, in order
                    const int l = __ffsll((unsigned long long)m) - 1;
                    m &= m - 1ull;
                    const int c_ne = (int)rdl(cne, l);
                    uint32_t doff = QZ_NONE;
                    if (!exhausted) doff = tree_alloc(E, D, new_edges, dnp, c_ne, lane, true);
                    if (doff == QZ_NONE) {  // pool empty: the rest of the subtree is cut off (counted)
                        exhausted = true;
                        truncated++;
                        if (lane == l) {
                            D.pool[pb + (uint32_t)l].cne = 0;
                            D.pool[pb + (uint32_t)l].coff = 0u;
                        }
                    } else {
                        if (lane == l) mydoff = doff;
                        new_nodes++;
                    }
                }
                const bool mine = cne > 0u && mydoff != QZ_NONE;
                uint64_t wide = __ballot(mine && cne > 8u);
                while (wide) {
                    const int l = __ffsll((unsigned long long)wide) - 1;
                    wide &= wide - 1ull;
                    copy_block(S, rdl(c_src, l), D, rdl(mydoff, l), (int)rdl(cne, l), pb + (uint32_t)l, lane);
                }
                {
                    const bool narrow = mine && cne <= 8u;
                    const uint32_t sp = tree_phys_lanes(S, c_src), dp = tree_phys_lanes(D, mydoff == QZ_NONE ? 0u : mydoff);  // (all lanes: shuffles)
                    Edge ed[8];  // all loads first (the stores below may alias them as far as the compiler knows)
#pragma unroll
                    for (uint32_t k = 0u; k < 8u; k++)
                        if (narrow && k < cne) ed[k] = S.pool[sp + k];
#pragma unroll
                    for (uint32_t k = 0u; k < 8u; k++) {
                        if (narrow && k < cne) {
                            ed[k].pedge = pb + (uint32_t)lane;
                            ed[k].spare = QZ_NONE;
                            D.pool[dp + k] = ed[k];
                            S.pool[sp + k].spare = dp + k;  // forwarding address (translate_records)
                        }
                    }
                }
                if (mine) D.pool[pb + (uint32_t)lane].coff = mydoff;
                q = lim;
            }
        } else {
            truncated = 1u;
        }
    }
    translate_records(E, b, lane, flipped ? edge : QZ_NONE);
    if (lane == 0) {
        if (st != nullptr) st[0] = 0u;
        // successful copy: the old tree (now the other half) goes back to the pool; otherwise the
        // board restarts from a fresh root and its current half is returned
        E.release[b] = flipped ? 1 : 2;
        if (flipped) {
            E.tree_half[b] = (uint8_t)(half ^ 1u);
            E.tree_npages[tree_slot(E, b, half ^ 1u)] = dnp;
        }
        E.path_len[b] = 0u;
        E.pl_done[b] = 0u;  // (asynchronous self-play: playouts on the new root)
        // the next compaction of this board: when its cursor has doubled (a tree whose LIVE part is larger than the
        // engine-wide threshold would otherwise be copied again at every move; measured: a handful of such boards made
        // every launch of a 1,024-board engine last 10-20 ms)
        if (E.compact_edges > 0) {
            uint32_t next = 2u * new_edges;
            if (next < (uint32_t)E.compact_edges) next = (uint32_t)E.compact_edges;
            const uint32_t cap = (uint32_t)QZ_TREE_PT * QZ_PAGE_EDGES * 3u / 4u;
            E.compact_at[b] = next < cap ? next : cap;
        }
        E.n_nodes[b] = new_nodes;
        E.n_edges[b] = new_edges;
        E.root_N[b] = childN;
        E.root_eoff[b] = root_off;
        E.root_ne[b] = (uint32_t)root_ne;
        if (truncated) E.bc_overflow[b] += truncated;
    }
    return true;
}

// lane 0 only; pages are given back by the caller (push-only kernels)
__device__ __forceinline__ void reset_board_state(EngineDev& E, int b) {
    Board o = opening();
    E.root_hb[b] = o.hb;
    E.root_vb[b] = o.vb;
    E.root_meta[b] = pack_meta(o);
    E.n_nodes[b] = 0u;
    E.n_edges[b] = 0u;
    E.root_N[b] = 0u;
    E.root_ne[b] = 0u;
    E.root_eoff[b] = 0u;
    E.path_len[b] = 0u;
    for (int r = 0; r < QZ_PATH_RECS; r++) E.rec_len[(size_t)b * QZ_PATH_RECS + r] = 0u;
    E.ply[b] = 0u;
    E.status[b] = QZ_PLAYING;
    E.winner[b] = 0;
    E.release[b] = 0;
    E.pl_done[b] = 0u;
    E.pend_slot[b] = QZ_NONE;
    E.reroot_pend[b] = 0u;
    E.compact_state[(size_t)b * 8] = 0u;
    E.compact_at[b] = E.compact_edges > 0 ? (uint32_t)E.compact_edges : 0u;
    E.game_serial[b] = E.game_serial[b] + 1u;
}

// PUSH-ONLY.  reset_boards: Quoridor.reset() + fresh trees + empty trajectories; otherwise
// only the trees are dropped (MCTSPlayer.reset_player, mcts.py:168-169).
__global__ __launch_bounds__(TPB) void k_reset(EngineDev E, int reset_boards) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    wave_free_tree_half(E, b, 0u, lane);
    wave_free_tree_half(E, b, 1u, lane);
    if (reset_boards) {
        wave_free_traj(E, b, lane);
        if (lane == 0) reset_board_state(E, b);
    } else if (lane == 0) {
        E.n_nodes[b] = 0u;
        E.n_edges[b] = 0u;
        E.root_N[b] = 0u;
        E.root_ne[b] = 0u;
        E.root_eoff[b] = 0u;
        E.path_len[b] = 0u;
        for (int r = 0; r < QZ_PATH_RECS; r++) E.rec_len[(size_t)b * QZ_PATH_RECS + r] = 0u;
        E.release[b] = 0;
        E.pl_done[b] = 0u;
        E.pend_slot[b] = QZ_NONE;
        E.reroot_pend[b] = 0u;
        E.compact_state[(size_t)b * 8] = 0u;
        E.compact_at[b] = E.compact_edges > 0 ? (uint32_t)E.compact_edges : 0u;
    }
}

// PUSH-ONLY, launched after every kernel that re-roots (k_finish_move, k_update_with_move):
// returns the pages of the trees that were replaced and restarts dropped games.
__global__ __launch_bounds__(TPB) void k_release(EngineDev E) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    const uint32_t rel = rfl((uint32_t)E.release[b]);
    const uint32_t half = rfl(E.tree_half[b]);
    if (rfl((uint32_t)E.status[b]) == QZ_ABORTED) {
        wave_free_tree_half(E, b, 0u, lane);
        wave_free_tree_half(E, b, 1u, lane);
        wave_free_traj(E, b, lane);
        if (lane == 0) reset_board_state(E, b);
        return;
    }
    if (rel & 1u) wave_free_tree_half(E, b, half ^ 1u, lane);
    if (rel & 2u) wave_free_tree_half(E, b, half, lane);
    if (rel && lane == 0) E.release[b] = 0;
}

__global__ __launch_bounds__(TPB) void k_update_with_move(EngineDev E, const uint8_t* __restrict__ moves) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    int mv = (int)rfl(moves[b]);
    uint32_t edge = QZ_NONE;
    int ne;
    const Edge* re = root_block(E, b, lane, ne);
    if (mv < QZ_N_ACT && re) {
        for (int base = 0; base < ne; base += 64) {
            int k = base + lane;
            bool hit = k < ne && re[k].act == mv;
            uint64_t m = __ballot(hit);
            if (m) edge = (uint32_t)(re - E.edge_pool) + (uint32_t)base + (uint32_t)(__ffsll((unsigned long long)m) - 1);
        }
    }
    wave_reroot(E, b, lane, edge);
}

// ---------------------------------------------------------------------------- sampling
// Philox4x32-10, keyed by the engine seed; the counter names (board, game, ply, stream, draw)
struct Philox {
    uint32_t k0, k1;
    __device__ uint4 operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b,
                     n3 = (uint32_t)p0;
            c0 = n0;
            c1 = n1;
            c2 = n2;
            c3 = n3;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {  // (0,1)
    uint64_t x = ((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11);
    return ((double)(x & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
}
// Gamma(alpha, 1) for alpha < 1: Marsaglia-Tsang on alpha+1, then the U^(1/alpha) boost
__device__ double gamma_small(const Philox& ph, uint32_t c0, uint32_t c1, uint32_t c2, double alpha) {
    double d = alpha + 1.0 - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0; t < 64; t++) {
        uint4 r = ph(c0, c1, c2, 2u * t);
        uint4 s = ph(c0, c1, c2, 2u * t + 1u);
        double u1 = u01(r.x, r.y), u2 = u01(r.z, r.w);
        double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double vv = 1.0 + c * x;
        if (vv <= 0.0) continue;
        vv = vv * vv * vv;
        double u = u01(s.x, s.y);
        if (u < 1.0 - 0.0331 * x * x * x * x || log(u) < 0.5 * x * x + d * (1.0 - vv + log(vv))) {
            double ub = u01(s.z, s.w);
            return d * vv * pow(ub, 1.0 / alpha);
        }
    }
    return alpha;
}

// MCTSPlayer.choose_action tail + one iteration of start_self_play (mcts.py:174-187,
// quoridor.py:585-602).  POP-ONLY (trajectory page, pages of the re-rooted tree).
__device__ __forceinline__ void finish_move_board(EngineDev& E, const int b, const int lane, const uint8_t* __restrict__ forced,
                                                  float* __restrict__ pi_out, uint8_t* __restrict__ move_out, const int reroot_mode = 0) {
    // reroot_mode: 0 = copy the kept subtree now (the lock-step engine), 1 = keep it in place, 2 = copy it LATER: the
    // edge is left in E.reroot_pend for the board's next k_advance launch, where the copy runs beside the other boards'
    // playouts instead of holding up a launch of its own (everything else of the move happens here)
    if (move_out && lane == 0) move_out[b] = QZ_NO_MOVE_U8;
    if (pi_out)
        for (int a = lane; a < QZ_N_ACT; a += 64) pi_out[(size_t)b * QZ_N_ACT + a] = 0.f;
    if (rfl(E.status[b]) != QZ_PLAYING) return;
    Board bd = load_board(E.root_hb, E.root_vb, E.root_meta, b);
    const uint32_t ply = rfl(E.ply[b]);
    int ne;
    Edge* re = root_block(E, b, lane, ne);
    // a game that cannot go on is dropped, counted by cause, and restarted by k_release:
    //   no legal move at the root (the reference prints "board is full" and crashes in
    //   start_self_play's unpack, mcts.py:195-196); qz_config.max_plies reached; no room left
    //   for its trajectory (page table full / pool empty)
    int abort_cause = -1;
    if (!re) abort_cause = QZ_C_ABORT_NO_MOVE;
    else if (E.max_plies > 0 && ply >= (uint32_t)E.max_plies) abort_cause = QZ_C_ABORT_MAX_PLIES;
    // the forced move must be a child of the root: otherwise the board stays where it is and the
    // sticky error counter goes up (qz_stats.bad_forced_moves)
    const int fm = forced ? (int)rfl(forced[b]) : QZ_NO_MOVE_U8;
    int act[3];
    int chosen_k = -1;
    if (abort_cause < 0) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            int k = lane + 64 * r;
            act[r] = k < ne ? (int)re[k].act : -1;
        }
        if (fm < QZ_N_ACT) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                uint64_t m = __ballot(act[r] == fm);
                if (m) chosen_k = 64 * r + (__ffsll((unsigned long long)m) - 1);
            }
            if (chosen_k < 0) {
                if (lane == 0) atomicAdd((unsigned long long*)&E.counters[QZ_C_BAD_FORCED], 1ull);
                return;
            }
        }
    }
    // room for this ply's record: [ne][0][board x6][pi f32 x ne][act u8 x ne, padded to 4]
    uint32_t* rec = nullptr;
    if (abort_cause < 0) {
        const uint32_t need = QZ_TRAJ_HDR + (uint32_t)ne + (((uint32_t)ne + 3u) >> 2);
        uint32_t* ptab = E.traj_ptab + (size_t)b * QZ_TRAJ_PT;
        uint32_t npg = rfl(E.traj_npages[b]), cur = rfl(E.traj_cursor[b]);
        if (npg == 0u || cur + need > E.traj_page_dwords) {
            uint32_t page = QZ_NONE;
            if (npg >= (uint32_t)QZ_TRAJ_PT) abort_cause = QZ_C_ABORT_MAX_PLIES;
            else {
                page = wave_pop(E.free_traj, E.pool_words + QZ_P_TRAJ_TOP, E.pool_words + QZ_P_TRAJ_LOW, lane);
                if (page == QZ_NONE) abort_cause = QZ_C_ABORT_POOL;
            }
            if (abort_cause < 0) {
                if (lane == 0) {
                    if (npg > 0u && cur < E.traj_page_dwords) E.traj_pool[(size_t)ptab[npg - 1u] * E.traj_page_dwords + cur] = QZ_TRAJ_SKIP;
                    ptab[npg] = page;
                    E.traj_npages[b] = npg + 1u;
                }
                cur = 0u;
                rec = E.traj_pool + (size_t)page * E.traj_page_dwords;
            }
        } else {
            rec = E.traj_pool + (size_t)rfl(ptab[npg - 1u]) * E.traj_page_dwords + cur;
        }
        if (abort_cause < 0 && lane == 0) E.traj_cursor[b] = cur + need;
    }
    if (abort_cause >= 0) {
        if (lane == 0) {
            atomicAdd((unsigned long long*)&E.counters[abort_cause], 1ull);
            E.status[b] = QZ_ABORTED;
            log_dropped_game(E, b, abort_cause);
        }
        return;
    }
    double pr[3];
    root_pi(re, ne, 1.0 / (double)E.temp, lane, pr);

    // record (board, pi) BEFORE the move (quoridor.py:589-591)
    wave_sync();  // pi_out was zero-filled by other lanes above
    uint8_t* rec_act = reinterpret_cast<uint8_t*>(rec + QZ_TRAJ_HDR + ne);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        if (k < ne) {
            reinterpret_cast<float*>(rec)[QZ_TRAJ_HDR + k] = (float)pr[r];
            rec_act[k] = (uint8_t)act[r];
            if (pi_out) pi_out[(size_t)b * QZ_N_ACT + act[r]] = (float)pr[r];
        }
    }
    if (lane == 0) {
        rec[0] = (uint32_t)ne;
        rec[1] = 0u;
        const uint64_t m = pack_meta(bd);
        rec[2] = (uint32_t)bd.hb;
        rec[3] = (uint32_t)(bd.hb >> 32);
        rec[4] = (uint32_t)bd.vb;
        rec[5] = (uint32_t)(bd.vb >> 32);
        rec[6] = (uint32_t)m;
        rec[7] = (uint32_t)(m >> 32);
    }

    // the move
    if (fm >= QZ_N_ACT) {
        Philox ph{(uint32_t)E.seed, (uint32_t)(E.seed >> 32)};
        uint32_t serial = rfl(E.game_serial[b]);
        double w[3];
        if (E.is_selfplay) {
            // 0.75*probs + 0.25*Dirichlet(alpha * ones(k)) (mcts.py:181)
            double g[3], gs = 0.0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                int k = lane + 64 * r;
                g[r] = (k < ne) ? gamma_small(ph, (uint32_t)b, serial, (ply << 8) | (uint32_t)k, (double)E.dirichlet_alpha) : 0.0;
                gs += g[r];
            }
            gs = wave_sum(gs);
#pragma unroll
            for (int r = 0; r < 3; r++) w[r] = (1.0 - (double)E.noise_frac) * pr[r] + (double)E.noise_frac * (g[r] / gs);
        } else {
#pragma unroll
            for (int r = 0; r < 3; r++) w[r] = pr[r];
        }
        // np.random.choice(acts, p=w): cdf = cumsum(w); idx = searchsorted(cdf/cdf[-1], u, 'right')
        double carry = 0.0, cdf[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            cdf[r] = carry + wave_scan(w[r], lane);
            carry = __shfl(cdf[r], 63, 64);
        }
        uint4 ur = ph((uint32_t)b, serial, (ply << 8) | 0xFFu, 0xC401CEu);
        double target = u01(ur.x, ur.y) * carry;
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            int k = lane + 64 * r;
            cnt += __popcll(__ballot(k < ne && cdf[r] <= target));
        }
        chosen_k = cnt < ne ? cnt : ne - 1;
    }
    const int mv = (int)rfl((uint32_t)re[chosen_k].act);
    if (move_out && lane == 0) move_out[b] = (uint8_t)mv;

    // update_with_move(move) in self-play, update_with_move(-1) otherwise (mcts.py:182,187)
    {
        const uint32_t keep = E.is_selfplay ? (uint32_t)(re - E.edge_pool) + (uint32_t)chosen_k : QZ_NONE;
        // (a child without a subtree makes the whole tree garbage: the cursor goes back to 0 in place, nothing to copy)
        const bool has_subtree = keep != QZ_NONE && rfl((uint32_t)E.edge_pool[keep].cne) != 0u;
        if (reroot_mode == 2 && has_subtree) {
            if (lane == 0) {
                E.reroot_pend[b] = keep == QZ_NONE ? 1u : keep + 2u;  // 0 = nothing pending, 1 = fresh root, e + 2 = keep edge e
                E.pl_done[b] = 0u;
            }
        } else {
            wave_reroot(E, b, lane, keep, reroot_mode != 0);
        }
    }

    // self.step(move); has_a_winner() (quoridor.py:593-596)
    bool done = apply_action(bd, mv);
    if (lane == 0) {
        E.root_hb[b] = bd.hb;
        E.root_vb[b] = bd.vb;
        E.root_meta[b] = pack_meta(bd);
        E.ply[b] = ply + 1u;
        atomicAdd((unsigned long long*)&E.counters[QZ_C_PLIES], 1ull);
        if (done) {
            E.status[b] = QZ_FINISHED;
            E.winner[b] = (uint8_t)winner_of(bd);
            atomicAdd((unsigned long long*)&E.counters[QZ_C_PENDING_GAMES], 1ull);
            atomicAdd((unsigned long long*)&E.counters[QZ_C_PENDING_PLIES], (unsigned long long)(ply + 1u));
        }
    }
}
__global__ __launch_bounds__(TPB) void k_finish_move(EngineDev E, const uint8_t* __restrict__ forced, float* __restrict__ pi_out,
                                                     uint8_t* __restrict__ move_out) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    finish_move_board(E, b, lane, forced, pi_out, move_out);
}

// ---------------------------------------------------------------------------- leaf-evaluation memo
// (layout: qz_device.h, MemoDev)
#ifndef QZ_MEMO_HASH32
#define QZ_MEMO_HASH32 1
#endif
// (which bucket a board's evaluation lives in: any function of the key will do -- a hit is decided by the compare of all 24 bytes)
__device__ __forceinline__ uint64_t memo_hash(uint64_t hb, uint64_t vb, uint64_t meta) {
#if QZ_MEMO_HASH32
    // Every 32-bit word of the key times an odd constant, the product's two halves folded together (s_mul_i32 + s_mul_hi_u32: a
    // product's low half only depends on the operand's bits below it, the high half brings the upper bits down), xor of the six,
    // one xor-shift / multiply / xor-shift: 27 scalar instructions.  Round 4's form -- five 64 x 64-bit multiplies, six scalar
    // instructions each -- was ~50 of the ~1,100 instructions of every playout of k_advance (the board is wave-uniform: the hash
    // runs on the scalar unit), and the kernel is issue-bound.  Checked offline on 780,000 leaf boards two plies around 1,024
    // synthetic positions: buckets of 2^17 and 2^20 fill like a Poisson process (overfull buckets 122,610 / 41,630 against 122,657 /
    // 41,620 expected; round 4's hash: the same).  A first version WITHOUT the high halves left 20 % more overfull buckets at 2^20.
    const uint32_t w[6] = {(uint32_t)hb, (uint32_t)(hb >> 32), (uint32_t)vb, (uint32_t)(vb >> 32), (uint32_t)meta, (uint32_t)(meta >> 32)};
    const uint32_t c[6] = {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD6E8FEB9u};
    uint32_t x = 0u;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint64_t p = (uint64_t)w[i] * (uint64_t)c[i];
        x ^= (uint32_t)p ^ (uint32_t)(p >> 32);
    }
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 13;
    return (uint64_t)x;
#else
    uint64_t x = (hb * 0x9E3779B97F4A7C15ull) ^ ((vb + 0xD1B54A32D192ED03ull) * 0xC2B2AE3D27D4EB4Full) ^ (meta * 0x165667B19E3779F9ull);
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return x;
#endif
}
// the key dword a lane compares its loaded dword with: entry dwords 0..5 = hb, vb, meta | epoch << 48
__device__ __forceinline__ uint32_t memo_key_dword(int pos, uint64_t hb, uint64_t vb, uint64_t mk) {
    return pos == 0 ? (uint32_t)hb : (pos == 1 ? (uint32_t)(hb >> 32) : (pos == 2 ? (uint32_t)vb : (pos == 3 ? (uint32_t)(vb >> 32) : (pos == 4 ? (uint32_t)mk : (uint32_t)(mk >> 32)))));
}
__device__ __forceinline__ bool memo_is_small(const Board& bd) { return (bd.cur == 1 ? bd.w1 : bd.w2) <= 0; }
struct MemoHit {
    uint32_t m0, m1, m2, m3, m4;  // the legal set
    float v;
    float p_lane;        // small table: the prior of pawn code `lane` (lane < 12)
    const float* p_row;  // big table: p[140]; nullptr for a small-table hit
};
// The probe in two halves, so that its one trip to memory (an 8-GB table: HBM, often a TLB miss) runs under other work:
// memo_probe_issue requests the bucket (small table: 512 B = 2 dwords per lane; big table: the two entries' first 16 dwords,
// one per lane) and memo_probe_finish compares the 24-byte keys by ballot.  All lanes must call both.
struct MemoProbe {
    uint32_t d0, d1;            // the lane's dwords of the bucket
    const uint32_t* bucket;     // big table: the bucket (p rows behind the headers); small table: unused
    bool small;
};
__device__ __forceinline__ MemoProbe memo_probe_issue(const EngineDev& E, const Board& bd, const int lane) {
    // (no branch around the loads: a conditionally loaded value meets its default in a phi, and the copy the phi needs waits
    // for the load on the spot -- the requests would not be in flight under anything.  Without a memo the lanes read the
    // descent records instead, an array that always exists, and memo_probe_finish ignores what comes back.)
    MemoProbe P;
    P.small = memo_is_small(bd);
    const uint64_t h = memo_hash(bd.hb, bd.vb, pack_meta(bd));
    const uint32_t* const Bs = E.memo.small + (size_t)((uint32_t)h & E.memo.small_mask) * (QZ_MEMO_S_WAYS * QZ_MEMO_S_DW);
    const uint32_t* const Bb = E.memo.big + (size_t)((uint32_t)h & E.memo.big_mask) * (QZ_MEMO_B_WAYS * QZ_MEMO_B_DW);
    const uint32_t* const B = E.memo.small ? (P.small ? Bs : Bb) : E.path_edges;
    const uint32_t pos = (uint32_t)lane & 31u;
    const uint32_t ib = ((uint32_t)lane >> 5) * QZ_MEMO_B_DW + (pos < 16u ? pos : 15u);
    const uint32_t i0 = P.small ? (uint32_t)lane : ib, i1 = P.small ? (uint32_t)lane + 64u : ib;
    P.bucket = B;
    P.d0 = B[E.memo.small ? i0 : 0u];
    P.d1 = B[E.memo.small ? i1 : 0u];
    return P;
}
__device__ __forceinline__ bool memo_probe_finish(const EngineDev& E, const uint32_t epoch, const Board& bd, const int lane, const MemoProbe& P, MemoHit& H) {
    if (!E.memo.small) return false;
    const uint64_t mk = pack_meta(bd) | ((uint64_t)epoch << 48);
    const int pos = lane & 31;
    const uint32_t kd = memo_key_dword(pos, bd.hb, bd.vb, mk);
    if (P.small) {
        const uint32_t d0 = P.d0, d1 = P.d1;
        const uint64_t e0 = __ballot(pos >= 6 || d0 == kd), e1 = __ballot(pos >= 6 || d1 == kd);
        // (the four halves as scalar words, compared with -1 by the scalar unit: written on the 64-bit ballots the compiler
        // made 64-bit VECTOR compares against constants it kept -- and spilled -- in vector register pairs)
        const uint32_t a0 = rfl((uint32_t)e0), a1 = rfl((uint32_t)(e0 >> 32)), a2 = rfl((uint32_t)e1), a3 = rfl((uint32_t)(e1 >> 32));
        const int e = a0 == 0xFFFFFFFFu ? 0 : (a1 == 0xFFFFFFFFu ? 1 : (a2 == 0xFFFFFFFFu ? 2 : (a3 == 0xFFFFFFFFu ? 3 : -1)));
        if (e < 0) return false;
        const uint32_t d = (e & 2) ? d1 : d0;
        const int base = (e & 1) * 32;
        H.v = __uint_as_float(rdl(d, base + 6));
        H.m0 = rdl(d, base + 7);
        H.m1 = H.m2 = H.m3 = H.m4 = 0u;
        H.p_lane = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((base + 8 + (lane < 12 ? lane : 0)) << 2, (int)d));
        H.p_row = nullptr;
        return true;
    }
    const uint32_t d = P.d0;  // (lanes 16..31 of a half hold dword 15 again: not part of the key)
    const uint64_t e0 = __ballot(pos >= 6 || d == kd);
    const uint32_t a0 = rfl((uint32_t)e0), a1 = rfl((uint32_t)(e0 >> 32));
    const int e = a0 == 0xFFFFFFFFu ? 0 : (a1 == 0xFFFFFFFFu ? 1 : -1);
    if (e < 0) return false;
    const int base = e * 32;
    H.v = __uint_as_float(rdl(d, base + 6));
    H.m0 = rdl(d, base + 8);
    H.m1 = rdl(d, base + 9);
    H.m2 = rdl(d, base + 10);
    H.m3 = rdl(d, base + 11);
    H.m4 = rdl(d, base + 12);
    H.p_lane = 0.f;
    H.p_row = reinterpret_cast<const float*>(P.bucket + e * QZ_MEMO_B_DW + 16);
    return true;
}
// store one evaluation (wave-cooperative; k_round_tail only: no probe runs at the same time)
// -> 0: nothing stored (no memo, or the evaluation is already there), 1: stored, 2: another wave has the bucket this round (skipped)
// A bucket takes ONE insert per round: its lock word holds the number of the last round in which a wavefront wrote it (round_id =
// the engine's count of k_advance launches, never 0), taken by compare-and-swap from the value the bucket was read with.  Nobody
// unlocks, so there is no fence and no second atomic per insert (round 4: lock 0 / 1, __threadfence -- a write-back of the XCD's
// whole L2 -- and an exchange to unlock, for every one of a round's thousands of inserts), and two writers can never tear an
// entry: the next writer of the bucket is a wavefront of a LATER launch.  Probes run in k_advance only: the kernel boundary
// publishes the entry.
__device__ __forceinline__ int memo_insert(const EngineDev& E, const uint64_t hb, const uint64_t vb, const uint64_t meta, const uint32_t* __restrict__ mask,
                                           const float* __restrict__ prow, const float v, const int lane, const uint32_t round_id) {
    if (!E.memo.small) return 0;
    const uint32_t epoch = rfl(*E.memo.epoch);
    const uint64_t mk = meta | ((uint64_t)epoch << 48);
    const uint64_t h = memo_hash(hb, vb, meta);
    const int pos = lane & 31;
    const uint32_t kd = memo_key_dword(pos, hb, vb, mk);
    const Board bd = unpack(hb, vb, meta);
    uint32_t* B;
    int way;
    uint32_t* lock;
    uint32_t held;  // the lock word as the bucket was read
    if (memo_is_small(bd)) {
        B = E.memo.small + (size_t)((uint32_t)h & E.memo.small_mask) * (QZ_MEMO_S_WAYS * QZ_MEMO_S_DW);
        const uint32_t d0 = B[lane], d1 = B[lane + 64];
        const uint64_t e0 = __ballot(pos >= 6 || d0 == kd), e1 = __ballot(pos >= 6 || d1 == kd);
        if ((uint32_t)e0 == 0xFFFFFFFFu || (uint32_t)(e0 >> 32) == 0xFFFFFFFFu || (uint32_t)e1 == 0xFFFFFFFFu || (uint32_t)(e1 >> 32) == 0xFFFFFFFFu) return 0;
        // first way of another epoch (empty / flushed), else a way picked by the hash
        const uint64_t l0 = __ballot(pos == 5 && (d0 >> 16) == epoch), l1 = __ballot(pos == 5 && (d1 >> 16) == epoch);
        const uint32_t livem = (uint32_t)((l0 >> 5) & 1ull) | ((uint32_t)((l0 >> 37) & 1ull) << 1) | ((uint32_t)((l1 >> 5) & 1ull) << 2) | ((uint32_t)((l1 >> 37) & 1ull) << 3);
        way = livem == 0xFu ? (int)(((uint32_t)h >> 30) & 3u) : (__ffs((int)(~livem & 0xFu)) - 1);  // (the bucket index uses at most the hash's low 24 bits)
        lock = B + 31;
        held = rdl(d0, 31);
    } else {
        B = E.memo.big + (size_t)((uint32_t)h & E.memo.big_mask) * (QZ_MEMO_B_WAYS * QZ_MEMO_B_DW);
        const bool in = pos < 16;
        const uint32_t d = in ? B[(lane >> 5) * QZ_MEMO_B_DW + pos] : 0u;
        const uint64_t e0 = __ballot(pos >= 6 || d == kd);
        if ((uint32_t)e0 == 0xFFFFFFFFu || (uint32_t)(e0 >> 32) == 0xFFFFFFFFu) return 0;
        const uint64_t l0 = __ballot(pos == 5 && (d >> 16) == epoch);
        const uint32_t livem = (uint32_t)((l0 >> 5) & 1ull) | ((uint32_t)((l0 >> 37) & 1ull) << 1);
        way = livem == 0x3u ? (int)(((uint32_t)h >> 31) & 1u) : (__ffs((int)(~livem & 0x3u)) - 1);
        lock = B + 15;
        held = rdl(d, 15);
    }
    if (held == round_id) return 2;  // another wave has written this bucket in this round: skip (the memo is a cache)
    uint32_t got = 0u;
    if (lane == 0) got = atomicCAS(lock, held, round_id) == held ? 1u : 0u;
    if (rfl(got) == 0u) return 2;
    if (memo_is_small(bd)) {
        uint32_t* W = B + way * QZ_MEMO_S_DW;
        if (lane < 20) {
            uint32_t val;
            if (lane < 6) val = kd;
            else if (lane == 6) val = __float_as_uint(v);
            else if (lane == 7) val = mask[0];
            else val = __float_as_uint(prow[lane - 8]);
            W[lane] = val;
        }
    } else {
        uint32_t* W = B + way * QZ_MEMO_B_DW;
        for (int i = lane; i < 16 + QZ_N_ACT; i += 64) {
            uint32_t val = 0u;
            if (i < 6) val = memo_key_dword(i, hb, vb, mk);
            else if (i == 6) val = __float_as_uint(v);
            else if (i >= 8 && i < 13) val = mask[i - 8];
            else if (i >= 16) val = __float_as_uint(prow[i - 16]);
            if (!(way == 0 && i == 15)) W[i] = val;  // (dword 15 of way 0 is the lock)
        }
    }
    return 1;
}

// ---------------------------------------------------------------------------- asynchronous self-play
// k_advance: every board (a wavefront each) runs the loop of MCTS.get_move_probs (mcts.py:135-139) ON ITS OWN for as
// long as it can: descend -> leaf; a terminal leaf is backed up at once; a leaf whose evaluation is in the memo is
// expanded from the memo and backed up; any other leaf goes into the miss list (compacted by one atomic) and the
// board waits for the next launch, which starts by consuming the network's answer (expand + backup, exactly
// k_expand_backup's code).  A board that has done n_playout playouts stops; k_moves (the next round's first launch)
// plays its move -- kept out of this kernel because finish_move_board inlined here set the register count of the
// whole kernel (two waves per SIMD, or spills in the descent).  Per board the sequence of operations is exactly the
// lock-step engine's (k_select / k_expand_backup / k_finish_move): trees, pi, sampled moves and harvested tuples are
// bit-identical; only the interleaving BETWEEN boards differs.
//   max_iters   playouts a board may start per launch (1 = one playout per launch, the lock-step cadence)
//   budget      wall-clock limit in s_memrealtime ticks (100 MHz) after which a board starts no new playout
//   par         which of the two miss counters this round uses (rounds alternate: the tail of round r clears the
//               counter of round r + 1 while nobody reads it)
#ifdef QZ_ADV_STAMPS  // diagnostic build only (tests/hip/Makefile, benchmarks/advance_stamps.py): where a wavefront's time goes in k_advance
__device__ unsigned long long g_adv_stamps[4096][16];  // per board, accumulated over launches: cycles per phase + counts
__device__ unsigned long long g_adv_stamps3[4096][4];  // cycles inside the descents: replay rounds, walked levels (+ set-up), record commit
__device__ unsigned long long g_adv_stamps2[4096][4];  // levels confirmed by replay, replay rounds, rounds that confirmed < 8 levels
#define QZ_AS_MARK(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); as_acc[k] += now_ - as_t; if ((k) == 0 && now_ - as_t > as_max[0]) as_max[0] = now_ - as_t; if ((k) == 4 && now_ - as_t > as_max[1]) as_max[1] = now_ - as_t; as_t = now_; }
#define QZ_AS_COUNT(k, v) { as_acc[k] += (unsigned long long)(v); }
#else
#define QZ_AS_MARK(k)
#define QZ_AS_COUNT(k, v)
#endif
constexpr uint32_t ADV_LCAP = 320;  // levels of a descent mirrored in LDS (3.75 KB per wavefront: 32 wavefronts per CU fit in the 160 KB); deeper levels are read back from memory
#define QZ_ADV_ROT 2048u  // boards the first slot moves on per round (select_opts bit 3)
#ifndef QZ_ADV_WPB
#define QZ_ADV_WPB 1  // wavefronts (= boards) per workgroup of k_advance (4: rounds 3-4; A/B)
#endif
// ONE wavefront per workgroup: the chip hands a workgroup's slot to the next workgroup when ALL its wavefronts have left, and a
// board leaves the launch the moment it meets a leaf for the network -- a board whose mover still has walls after one playout.
// With four boards per workgroup such a slot stayed empty until its three neighbours were done too.
constexpr int ADV_WPB = QZ_ADV_WPB;
#ifndef QZ_ADV_WAVES_BIG
#define QZ_ADV_WAVES_BIG 7  // wavefronts per SIMD the build of k_advance for engines of more than 4,096 boards aims at (A/B: 8, 6)
#endif
#ifndef QZ_ADV_WAVES_SMALL
#define QZ_ADV_WAVES_SMALL 4  // wavefronts per SIMD the build of k_advance for engines of <= 4,096 boards aims at (A/B: 8 = one build for all sizes)
#endif
// A field of the engine descriptor FETCHED by a scalar load of its own from the kernel-argument segment (`kp_`; EngineDev is the
// first argument of k_advance / k_rows: offset 0) into the kernel's copy `E`.  A field read from the by-value argument is a
// sub-register of an 8- or 16-dword piece, which the register allocator keeps, spills and reloads WHOLE (eight or sixteen
// v_readlane for one pointer); an empty asm on a copy does not help, the coalescer joins the copy back into the piece.
#define QZ_KARG_P(T, f) { unsigned long long a_; asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a_) : "s"(kp_), "n"(offsetof(EngineDev, f))); E.f = (T*)(__attribute__((address_space(1))) T*)a_; }
#define QZ_KARG_S(f) { uint32_t a_; asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a_) : "s"(kp_), "n"(offsetof(EngineDev, f))); static_assert(sizeof(E.f) == 4, "32-bit field"); __builtin_memcpy(&E.f, &a_, 4); }
__device__ __forceinline__ void advance_board(EngineDev& E, const int max_iters, const unsigned int budget, const int par) {
    __shared__ uint32_t s_we[ADV_WPB][ADV_LCAP];
    __shared__ unsigned long long s_wb[ADV_WPB][ADV_LCAP];
    __shared__ uint32_t s_lc[ADV_WPB][LC_WORDS];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int w_ = __builtin_amdgcn_readfirstlane((int)blockIdx.x * ADV_WPB + wave);
    if (w_ >= E.n_boards) return;
    // select_opts bit 3, for engines of more boards than the chip holds wavefronts: ONE deadline for the launch -- the budget counts
    // from the start of the launch's first wavefront, not from each one's own -- so that a board that gets its slot in the middle
    // of the launch (when a board that met a leaf for the network has left) works until the same moment as the others instead of
    // stretching the launch by its own budget; and the boards take the first slots in turn (QZ_ADV_ROT further on every round:
    // a multiple of 8, so a board stays on its XCD and its tree in that XCD's L2).
    // (Only with a budget of 100 us or more: under a shorter one the boards behind the first slots would never start a playout.)
    const bool shared = (E.select_opts & 8) != 0 && budget >= 10000u && budget != 0xFFFFFFFFu;
    uint32_t seq = 0u;
    int b_ = w_;
    if (shared) {
        seq = rfl((uint32_t)E.miss_count[2]);
        b_ = (int)(((unsigned int)w_ + (seq % 4096u) * QZ_ADV_ROT) % (unsigned int)E.n_boards);
    }
    const int b = __builtin_amdgcn_readfirstlane(b_);  // in an SGPR: every per-board address below is scalar arithmetic
    // The engine descriptor arrives in the kernel-argument segment and is fetched in 16-dword pieces; left alone, a piece is ONE
    // value to the register allocator -- kept or spilled whole, and reloaded whole (sixteen v_readlane) wherever one field of
    // it is used.  The fields the loop uses are made values of their own here.
    // (A pointer goes through the asm as an integer and comes back as a pointer to GLOBAL memory -- address space 1 --
    // explicitly: left generic, every access through it becomes a FLAT instruction, whose wait is vmcnt(0) lgkmcnt(0).)
    // Round 6: "made values of their own" through an empty asm was not enough -- the register coalescer joins such a copy back into
    // the piece it was copied from, and the piece (E.edge_pool's: eight dwords) was still spilled and reloaded WHOLE, eight
    // v_readlane for one pointer, at six places of the descent (38 such places in the loop, 8 or 16 lanes each).  So every field the
    // loop uses is FETCHED by a scalar load of its own from the kernel-argument segment (EngineDev is the kernel's first argument:
    // offset 0), which no piece is ever part of; the fields only the prologue and the epilogue use are fetched again behind the loop
    // (QZ_KARG_COLD) so that nothing of the descriptor lives across it.
    const unsigned long long kp_ = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    QZ_KARG_P(Edge, edge_pool); QZ_KARG_P(uint32_t, path_edges); QZ_KARG_P(unsigned long long, path_blocks); QZ_KARG_P(uint32_t, memo.small);
    QZ_KARG_P(uint32_t, memo.big); QZ_KARG_P(uint32_t, free_tree); QZ_KARG_P(int, pool_words); QZ_KARG_P(unsigned long long, counters);
    QZ_KARG_P(uint64_t, root_hb); QZ_KARG_P(uint64_t, root_vb); QZ_KARG_P(uint64_t, root_meta);
    QZ_KARG_S(memo.small_mask); QZ_KARG_S(memo.big_mask); QZ_KARG_S(c_puct); QZ_KARG_S(n_playout); QZ_KARG_S(max_depth);
    QZ_KARG_S(select_opts); QZ_KARG_S(node_cap); QZ_KARG_S(edge_cap); QZ_KARG_S(fix_terminal_sign); QZ_KARG_S(tree_pool_pages);
    if (b == 0 && lane == 0) atomicAdd(&E.counters[QZ_C_ROUNDS], 1ull);
    // (the first launch: a board that is not playing leaves before the stamp -- if it is the first wavefront's, about one launch in
    // 250, that launch's wavefronts count the budget from their own starts.  Moving the test behind the stamp, as the second launch
    // has it, costs this build 59 more register-spill moves: left as it is.)
    if (rfl(E.status[b]) != QZ_PLAYING) return;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_it = t0;
    bool late = false;  // this wavefront's budget began before it did
    if (shared) {
        // the launch's first wavefront leaves (its start << 20 | round number) for the others; a wavefront that finds another
        // round's number there started in the launch's first microsecond: its own start will do.  (Read past the CU's L1: a line
        // fetched by an early wavefront of this CU would answer the late ones.)
        unsigned long long* const st = reinterpret_cast<unsigned long long*>(E.miss_count + 4);
        if (w_ == 0) {
            if (lane == 0) __hip_atomic_store(st, (t0 << 20) | (unsigned long long)(seq & 0xFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long sv = rfl64(__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if ((uint32_t)(sv & 0xFFFFFull) == (seq & 0xFFFFFu)) {
                constexpr unsigned long long M44 = (1ull << 44) - 1ull;
                unsigned long long first = (t0 & ~M44) | (sv >> 20);
                if (first > t0) first -= 1ull << 44;
                if (t0 - first < 0x40000000ull) {  // (a stamp of this round number from a million rounds ago is not this launch's)
                    t0 = first;
                    late = true;
                }
            }
        }
    }
    // select_opts bit 5: the boards on which neither player has a wall left are k_rows'
    // (qz_rows.h: sixteen lanes per board, four boards per wavefront), beside this launch
    if ((E.select_opts & 32) && ((rfl64(E.root_meta[b]) >> 16) & 0xFFFFull) == 0ull) return;
#ifndef QZ_BUDGET_PREDICT
#define QZ_BUDGET_PREDICT 1  // what a board expects its next playout to last: 0 = nothing, 1 = as long as its last one, 2 = the largest of its recent ones (a maximum that decays by a quarter per playout: measured no different from 1, 289.1 against 290.6 M playouts/s; nor is a margin of a quarter or a half of the last playout on top: 309.5 / 309.8 against 309.0 M, launches as long as before -- the launch's overrun of ~100 us is the extreme of ten thousand boards' playout times, not a misprediction of the typical one)
#endif
    unsigned int pred = 0u;
#ifdef QZ_ADV_STAMPS
    unsigned long long as_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, as_max[2] = {0, 0}, as_t = __builtin_amdgcn_s_memtime();
    const unsigned long long as_t0 = as_t;
#endif
    PathMirror PM{(lds_u32*)(uintptr_t)rfl((uint32_t)(uintptr_t)(lds_u32*)s_we[wave]), (lds_u64*)(uintptr_t)rfl((uint32_t)(uintptr_t)(lds_u64*)s_wb[wave]), ADV_LCAP, 0u};
    const PathMirror NOPM{(lds_u32*)nullptr, (lds_u64*)nullptr, 0u, 0u};
    // a move whose subtree copy is not done yet (k_compact, sliced at ITS budget): the board sits out.  A copy that found the
    // pool empty restarts the board from a fresh root in the SAME table half, whose pages k_round_tail is about to hand
    // back: nothing may be built there before
    if (rfl(E.reroot_pend[b]) != 0u || (rfl((uint32_t)E.release[b]) & 2u)) return;
    BoardRegs S = regs_load(E, b, lane, (lds_u32*)s_lc[wave]);
    const uint32_t epoch = rfl(*E.memo.epoch);
    uint32_t done = rfl(E.pl_done[b]), open_rounds = 0u;
    if ((S.root.cur == 1 ? S.root.w1 : S.root.w2) > 0) open_rounds = 1u;
    // The evaluation this board was waiting for (the leaf of the previous launch: its descent's path is in memory only) is
    // applied first -- TreeNode.expand + update_recursive with the network's answer -- then the loop: descend, resolve the
    // leaf (a terminal leaf's +-1, or its evaluation from the memo), apply at once.  (Round 3 carried the resolved leaf --
    // legal set, priors, value, path -- over the loop's back edge into ONE copy of the apply code: sixteen loop-carried
    // scalars more than the kernel has scalar registers for.)
    const uint32_t slot = rfl(E.pend_slot[b]);
    bool waiting = false;
    Board miss_leaf = S.root;
    uint32_t pedge = QZ_NONE, plen = 0u;
    if (slot != QZ_NONE) {
        const uint32_t* mk = E.miss_mask + (size_t)slot * 5;
        const float* const prow = E.miss_p + (size_t)slot * QZ_N_ACT;
        const double value = (double)E.miss_v[slot];
        pedge = rfl(E.leaf_pedge[b]);
        plen = rfl(E.path_len[b]);
        const unsigned long long blk = expand_node(E, S, lane, pedge, rfl(mk[0]), rfl(mk[1]), rfl(mk[2]), rfl(mk[3]), rfl(mk[4]), [&](int a) { return prow[a]; });
        note_expansion(E, S, b, lane, NOPM, plen, blk);
        backup_leaf(E, S, b, lane, NOPM, value, pedge, plen, 0u);
        done++;
        wave_sync();
    }
    QZ_AS_MARK(0)  // 0: launch prologue (state load, the pending evaluation)
    for (int it = 0; it < max_iters; it++) {
        // the lane / board indices of THIS iteration, opaque to the optimiser: without this every per-lane and per-board
        // address of the loop body is hoisted out of the loop as a 64-bit value -- dozens of them, more than there are
        // registers, so they went to scratch and came back through memory in the dependent chain of every playout
        int ln = lane, bb = b;
        asm volatile("" : "+v"(ln));
        asm volatile("" : "+s"(bb));
        // every loop-carried scalar is wave-uniform; said so explicitly at the top of each iteration, or the compiler keeps
        // them (and everything computed from them) in vector registers it does not have: the copies went to scratch, and a
        // scratch reload waits for s_waitcnt vmcnt(0), i.e. for every global store issued before it
        // the root position is read again for every descent (three scalar loads, issued here, used after the budget check): it
        // does not change during the launch -- moves are k_moves' -- and held in registers it was nine loop-carried scalars
        S.root = load_board(E.root_hb, E.root_vb, E.root_meta, bb);
        regs_uniform(S);
        done = rfl(done);
        PM.valid = rfl(PM.valid);
        if (done >= (uint32_t)E.n_playout) break;  // the move is k_moves' job (the next round's first launch)
        {   // no new playout once the budget is spent -- or would be overrun by a playout as long as this board's last one: boards
            // digging a long line take 100+ us per descent, and the launch ends with its LAST wave (profiles/round3: the
            // launches ran 0.29 ms over a 1 ms budget before this)
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            const unsigned int last = (unsigned int)(now - t_it);
            t_it = now;
            pred = rfl(pred);
            pred -= pred >> 2;
            if (QZ_BUDGET_PREDICT != 2 || last > pred) pred = last;
            if ((it > 0 || late) && (unsigned int)(now - t0) + (QZ_BUDGET_PREDICT ? pred : 0u) > budget) break;
        }
        Board leaf;
        uint32_t term;
        MemoProbe MP;  // the memo bucket of the leaf, requested the moment the leaf is known: in flight under the record commit
        select_core(E, S, bb, ln, PM, leaf, pedge, plen, term, [&](const Board& lf, bool) { MP = memo_probe_issue(E, lf, ln); });
        if (E.max_depth > 0 && plen > (uint32_t)E.max_depth) {  // (rare: its fields are fetched here)
            QZ_KARG_P(uint8_t, status); QZ_KARG_P(unsigned long long, drop_log); QZ_KARG_P(uint32_t, ply);
            if (drop_if_too_deep(E, bb, ln, plen)) break;
        }
        PM.valid = plen < ADV_LCAP ? plen : ADV_LCAP;
        wave_sync();  // the descent buffer (lane 0 / other lanes) before the backup reads it
        QZ_AS_MARK(4)  // 4: descent
        QZ_AS_COUNT(9, plen)
        if (term != 0u) {
            backup_leaf(E, S, bb, ln, PM, terminal_value(E, term), pedge, plen, term);
            done++;
            wave_sync();
            QZ_AS_MARK(2)  // 2: backup
            QZ_AS_COUNT(8, 1)
            continue;
        }
        MemoHit H;
        if (memo_probe_finish(E, epoch, leaf, ln, MP, H)) {
            lc_add(S, LC_HITS, 1u, ln);
            QZ_AS_MARK(5)  // 5: memo probe (hit)
            const float* const prow = H.p_row;  // priors as a row of 140 floats, or (nullptr) this lane's prior in p_lane (small-table hit)
            const float pl = H.p_lane;
            const unsigned long long blk = expand_node(E, S, ln, pedge, H.m0, H.m1, H.m2, H.m3, H.m4, [&](int a) { return prow ? prow[a] : pl; });
            note_expansion(E, S, bb, ln, PM, plen, blk);
            QZ_AS_MARK(1)  // 1: expansion
            backup_leaf(E, S, bb, ln, PM, (double)H.v, pedge, plen, 0u);
            done++;
            wave_sync();
            QZ_AS_MARK(2)  // 2: backup
            QZ_AS_COUNT(8, 1)
            continue;
        }
        QZ_AS_MARK(6)  // 6: memo probe (miss)
        // ---- a leaf for the network: recorded after the loop
        miss_leaf = leaf;
        waiting = true;
        break;
    }
    // what the epilogue writes to: fetched here, not carried through the loop
    QZ_KARG_S(n_boards);
    QZ_KARG_P(int, miss_count); QZ_KARG_P(uint64_t, miss_hb); QZ_KARG_P(uint64_t, miss_vb); QZ_KARG_P(uint64_t, miss_meta); QZ_KARG_P(uint32_t, pend_slot);
    QZ_KARG_P(uint32_t, leaf_pedge); QZ_KARG_P(uint32_t, path_len); QZ_KARG_P(uint32_t, rec_len); QZ_KARG_P(uint32_t, rec_stamp); QZ_KARG_P(uint32_t, rec_last);
    QZ_KARG_P(uint32_t, rec_clock); QZ_KARG_P(uint32_t, bc_playouts); QZ_KARG_P(uint32_t, bc_terminal); QZ_KARG_P(uint32_t, bc_overflow);
    QZ_KARG_P(uint32_t, bc_nonfinite); QZ_KARG_P(uint32_t, bc_maxdepth); QZ_KARG_P(uint32_t, bc_memo_hits); QZ_KARG_P(uint32_t, bc_evals);
    QZ_KARG_P(unsigned long long, bc_levels); QZ_KARG_P(unsigned long long, bc_scanned); QZ_KARG_P(unsigned long long, bc_expanded);
    QZ_KARG_P(uint32_t, root_N); QZ_KARG_P(uint32_t, root_ne); QZ_KARG_P(uint32_t, root_eoff); QZ_KARG_P(uint32_t, n_nodes); QZ_KARG_P(uint32_t, n_edges);
    QZ_KARG_P(uint32_t, tree_npages); QZ_KARG_P(uint32_t, pl_done); QZ_KARG_P(uint32_t, bc_open_rounds);
    if (waiting) {
        uint32_t s = 0u;
        if (lane == 0) s = (uint32_t)atomicAdd(E.miss_count + par, 1);
        s = rfl(s);
        // (a board has at most one leaf in the list, so a slot beyond n_boards means the counter was not cleared -- a host
        // that mixed up the two counters; the list must not be written past its end: the board forgets this descent and
        // repeats it in its next launch, the rules op and the network clamp the count they read)
        if (s >= (uint32_t)E.n_boards) {
            if (lane == 0) atomicAdd(&E.counters[QZ_C_MISS_OVERFLOW], 1ull);
            waiting = false;
        } else {
            if (lane == 0) {
                E.miss_hb[s] = miss_leaf.hb;
                E.miss_vb[s] = miss_leaf.vb;
                E.miss_meta[s] = pack_meta(miss_leaf);
                E.pend_slot[b] = s;
                E.leaf_pedge[b] = pedge;
                E.path_len[b] = plen;
            }
            {   // the path of this descent for the backup of the NEXT launch: the levels the LDS mirror holds, out to memory
                uint32_t* const path = E.path_edges + ((size_t)b * (QZ_PATH_RECS + 1) + QZ_PATH_RECS) * QZ_PATH_CAP;
                const uint32_t nm = plen < ADV_LCAP ? plen : ADV_LCAP;
                for (uint32_t i = (uint32_t)lane; i < nm; i += 64u) path[i] = PM.we[i];
            }
            lc_add(S, LC_EVALS, 1u, lane);
        }
    }
    regs_store(E, b, lane, S);
#ifdef QZ_ADV_STAMPS
    QZ_AS_MARK(7)  // 7: epilogue (miss record, state store)
    if (lane == 0 && b < 4096) {
        for (int k = 0; k < 10; k++) g_adv_stamps[b][k] += as_acc[k];
        for (int k = 0; k < 4; k++) g_adv_stamps3[b][k] += S.t_sel[k];
        g_adv_stamps2[b][0] += S.lc[LC_SPARE];
        g_adv_stamps2[b][1] += S.lc[14];
        g_adv_stamps2[b][2] += S.lc[15];
        const unsigned long long whole = __builtin_amdgcn_s_memtime() - as_t0;
        g_adv_stamps[b][10] += whole;
        g_adv_stamps[b][11] += 1ull;
        // launches that ran 15 % over their budget: how many, and what their prologue (a deferred compaction) and longest descent took
        if ((unsigned int)(__builtin_amdgcn_s_memrealtime() - t0) > budget + budget / 7u) {
            g_adv_stamps[b][12] += 1ull;
            g_adv_stamps2[b][3] += as_max[0];

            g_adv_stamps[b][15] += whole;
        }
        if (as_max[1] > g_adv_stamps[b][13]) g_adv_stamps[b][13] = as_max[1];  // longest single descent
        if (whole > g_adv_stamps[b][14]) g_adv_stamps[b][14] = whole;          // longest launch of this board
    }
#endif
    if (lane == 0) {
        E.pl_done[b] = done;
        if (!waiting && slot != QZ_NONE) E.pend_slot[b] = QZ_NONE;
        if (open_rounds) E.bc_open_rounds[b] += open_rounds;
    }
}

// Two builds of the loop.  k_advance<QZ_ADV_WAVES_BIG> for engines of more than 4,096 boards: the loop is bound by its instruction
// streams (vector pipes and scalar units ~75 % busy, a wavefront waiting for its turn to issue a third of its cycles), so registers
// that save instructions are worth more than the last wavefront slot: SEVEN wavefronts per SIMD (72 vector / 96 scalar registers
// allowed, 66 / 94 used, no scratch, 287 spill reloads in the loop) -- 7,168 boards at a time -- run 1.9 % more playouts per second
// than eight (64 / 80 registers, 12 bytes of scratch, 342 reloads; rounds 4-5's build), six 0.7 % fewer (profiles/round6/SUMMARY.md 7).
// k_advance<4>: 97 registers, for engines of up to 4,096 boards, where residency beyond four buys nothing.
// (EngineDev must stay the FIRST parameter: advance_board fetches its fields from offset 0 of the kernel-argument segment, QZ_KARG_*)
template <int W>
__global__ __launch_bounds__(64 * ADV_WPB) __attribute__((amdgpu_waves_per_eu(W, W))) void k_advance(EngineDev E, int max_iters, unsigned int budget, int par) {
    advance_board(E, max_iters, budget, par);
}

// k_moves: MCTSPlayer.choose_action's tail + one iteration of start_self_play's loop (finish_move_board) for every
// board that has done its n_playout playouts.  The move keeps the subtree in place while the tree is small; once the
// allocation cursor has passed compact_edges it compacts: the breadth-first copy of the kept subtree into fresh pages (the
// old half goes back to the pool in k_round_tail), by the same wavefront right after its move and in SLICES -- a copy that
// has not finished when `budget` (s_memrealtime ticks since the launch began) is spent saves its position in
// compact_state[b] and goes on in the next round's launch; the board sits out of k_advance meanwhile (reroot_pend).  Round 3
// ran the copies in k_advance's prologue: their registers and scratch were k_advance's.  POP-ONLY.
__device__ __forceinline__ void moves_board(EngineDev& E, const int b, const int lane, const unsigned int budget) {
    if (rfl(E.status[b]) != QZ_PLAYING || rfl((uint32_t)E.release[b]) != 0u) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t rp = rfl(E.reroot_pend[b]);
    if (rp == 0u) {
        if (rfl(E.pl_done[b]) < (uint32_t)E.n_playout) return;
        const bool compact = E.compact_edges <= 0 || rfl(E.n_edges[b]) >= rfl(E.compact_at[b]);
        const Board rb = unpack(0ull, 0ull, rfl64(E.root_meta[b]));
        if ((rb.cur == 1 ? rb.w1 : rb.w2) > 0 && lane == 0) E.bc_open_plies[b] += 1u;
        finish_move_board(E, b, lane, nullptr, nullptr, nullptr, compact ? 2 : 1);
        if (!compact) return;
        __threadfence();
        wave_sync();
        rp = rfl(__hip_atomic_load(E.reroot_pend + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));  // (lane 0 of this wave stored it)
        if (rp == 0u || rfl((uint32_t)__hip_atomic_load(E.status + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != QZ_PLAYING) return;
    }
    if (!wave_reroot(E, b, lane, rp == 1u ? QZ_NONE : rp - 2u, false, t0 + budget, E.compact_state + (size_t)b * 8)) {
        if (lane == 0) atomicAdd(&E.counters[QZ_C_COMPACT_SLICES], 1ull);
        return;  // the copy goes on in the next round
    }
    if (lane == 0) E.reroot_pend[b] = 0u;
}
__global__ __launch_bounds__(TPB) void k_moves(EngineDev E, unsigned int budget) {
    // (ONE wavefront per workgroup: beside the network's trunk -- qz_selfplay_round -- every SIMD's register file is full, and a
    // workgroup of four wavefronts needs room on all four SIMDs of a CU at once: it waited for the trunk's grid to run dry and the
    // round's tail waited 117-133 us for the moves; a one-wavefront workgroup takes the slot of any trunk wavefront that retires)
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (int)(blockDim.x >> 6) + wave);
    if (b >= E.n_boards) return;
    moves_board(E, b, lane, budget);
}
// After the network: (a) every evaluated leaf goes into the memo, (b) the OTHER miss counter is cleared for the next
// round, (c) k_release's work: trees replaced by a re-root go back to the pool, dropped games restart.  PUSH-ONLY.
// Grid (round 5): TAIL_SLOT_WAVES wavefronts stride over the miss list (device-side count), then one LANE per board looks
// whether its board has anything to hand back -- almost none has -- and the wavefront serves those that do one after the
// other.  (Round 4 launched a wavefront per slot of the list's CAPACITY and a wavefront per board: 2 x n_boards wavefronts
// that mostly read one word and left -- 66-89 us per round for a few microseconds of work.)
#ifndef QZ_TAIL_COMPACT
#define QZ_TAIL_COMPACT 1
#endif
constexpr int TAIL_SLOT_WAVES = 4096;
__device__ __forceinline__ void tail_board(EngineDev& E, const int b, const int lane) {
    const uint32_t rel = rfl((uint32_t)E.release[b]);
    const uint32_t half = rfl(E.tree_half[b]);
    if (rfl((uint32_t)E.status[b]) == QZ_ABORTED) {
        wave_free_tree_half(E, b, 0u, lane);
        wave_free_tree_half(E, b, 1u, lane);
        wave_free_traj(E, b, lane);
        if (lane == 0) reset_board_state(E, b);
        return;
    }
    if (rel & 1u) wave_free_tree_half(E, b, half ^ 1u, lane);
    if (rel & 2u) wave_free_tree_half(E, b, half, lane);
    if (rel && lane == 0) E.release[b] = 0;
}
__global__ __launch_bounds__(TPB) void k_round_tail(EngineDev E, int par) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int w = (int)blockIdx.x * WPB + wave;
#if QZ_TAIL_COMPACT
    if (w < TAIL_SLOT_WAVES) {
        int n = (int)rfl((uint32_t)E.miss_count[par]);
        n = n < E.n_boards ? n : E.n_boards;
        unsigned long long n_ins = 0ull, n_lck = 0ull;
        const uint32_t round_id = rfl((uint32_t)E.counters[QZ_C_ROUNDS]) | 0x80000000u;  // (this round's k_advance has counted itself; never 0)
        for (int sl = w; sl < n; sl += TAIL_SLOT_WAVES) {
            const int r = memo_insert(E, rfl64(E.miss_hb[sl]), rfl64(E.miss_vb[sl]), rfl64(E.miss_meta[sl]), E.miss_mask + (size_t)sl * 5,
                                      E.miss_p + (size_t)sl * QZ_N_ACT, E.miss_v[sl], lane, round_id);
            n_ins += r == 1 ? 1ull : 0ull;
            n_lck += r == 2 ? 1ull : 0ull;
        }
        if (lane == 0) {  // (one of QZ_C_SPREAD addresses each: not one address for thousands of wavefronts)
            if (n_ins) atomicAdd(&E.counters[QZ_C_COUNT + (w & (QZ_C_SPREAD - 1))], n_ins);
            if (n_lck) atomicAdd(&E.counters[QZ_C_COUNT + QZ_C_SPREAD + (w & (QZ_C_SPREAD - 1))], n_lck);
        }
        if (w == 0 && lane == 0) {
            E.miss_count[par ^ 1] = 0;
            E.miss_count[2] = (E.miss_count[2] + 1) & 0xFFFFF;  // rounds finished (k_advance's shared deadline, select_opts bit 3)
        }
        return;
    }
    const int b0 = (w - TAIL_SLOT_WAVES) * 64;
    if (b0 >= E.n_boards) return;
    const int bl = b0 + lane;
    const bool need = bl < E.n_boards && (E.release[bl] != 0 || E.status[bl] == QZ_ABORTED);
    uint64_t todo = __ballot(need);
    while (todo) {  // wave-uniform: the boards of this wavefront's 64 that have pages to hand back / a game to restart
        const int j = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1ull;
        tail_board(E, b0 + j, lane);
    }
#else
    if (w < E.n_boards) {
        const int n = (int)rfl((uint32_t)E.miss_count[par]);
        if (w < n) {
            const int r = memo_insert(E, rfl64(E.miss_hb[w]), rfl64(E.miss_vb[w]), rfl64(E.miss_meta[w]), E.miss_mask + (size_t)w * 5,
                                      E.miss_p + (size_t)w * QZ_N_ACT, E.miss_v[w], lane, rfl((uint32_t)E.counters[QZ_C_ROUNDS]) | 0x80000000u);
            if (lane == 0 && r) atomicAdd(&E.counters[r == 1 ? QZ_C_MEMO_INSERTS : QZ_C_MEMO_LOCKED], 1ull);
        }
        if (w == 0 && lane == 0) {
            E.miss_count[par ^ 1] = 0;
            E.miss_count[2] = (E.miss_count[2] + 1) & 0xFFFFF;  // rounds finished (k_advance's shared deadline, select_opts bit 3)
        }
        return;
    }
    const int b = w - E.n_boards;
    if (b >= E.n_boards) return;
    tail_board(E, b, lane);
#endif
}
// qz_memo_flush: the weights changed, every stored evaluation is dead
__global__ void k_memo_flush(EngineDev E) {
    uint32_t e = *E.memo.epoch + 1u;
    if (e >= 0xFFFFu) e = 0xFFFFu;  // (the host re-zeroes the tables before the epoch could wrap: qz_memo_flush)
    *E.memo.epoch = e;
}

// ---------------------------------------------------------------------------- random rollouts
// MCTS._evaluate_rollout (pure_mcts.py:81-103) for a batch of boards, one launch per iteration.
// The reference's loop: has_a_winner() first; at iteration limit-1 an unfinished game stops with
// "no winner" (value 0); otherwise a uniformly random legal action (the max over np.random.rand of
// the legal list, pure_mcts.py:7-10, 94-98) is played.  Here the winner test follows the move
// immediately (the same decision one iteration earlier, and move generation never sees a finished
// board, whose winning pawn may stand off the board).  `mask5` = actions() of the boards as they
// are now (this iteration's move-generation launch); value[b] = +1 if the winner is `player0[b]`
// (the side to move when the rollout began), -1 if the other, 0 without a winner.
__global__ __launch_bounds__(256) void k_rollout_step(uint64_t* hb, uint64_t* vb, uint64_t* meta, const uint32_t* __restrict__ mask5,
                                                      int n, const uint8_t* __restrict__ player0, uint8_t* __restrict__ done,
                                                      int8_t* __restrict__ value, int* __restrict__ n_done, uint64_t seed, int step,
                                                      int limit) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n || done[i]) return;
    uint32_t m[5];
#pragma unroll
    for (int k = 0; k < 5; k++) m[k] = mask5[(size_t)i * 5 + k];
    const int cnt = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]) + __popc(m[4]);
    if (step >= limit - 1 || cnt == 0) {
        // cnt == 0 on a live board: the reference's max() over an empty list raises; here: no winner
        value[i] = 0;
        done[i] = 1;
        atomicAdd(n_done, 1);
        return;
    }
    Board b = unpack(hb[i], vb[i], meta[i]);
    Philox ph{(uint32_t)seed, (uint32_t)(seed >> 32)};
    const uint4 r = ph((uint32_t)i, (uint32_t)step, 0x524F4C4Cu, 0u);
    int pick = (int)(((uint64_t)r.x * (uint64_t)cnt) >> 32);  // uniform in [0, cnt)
    int a = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int c = __popc(m[k]);
        if (pick >= 0 && pick < c) {
            uint32_t word = m[k];
            for (int j = 0; j < pick; j++) word &= word - 1u;  // drop the `pick` lowest set bits
            a = 32 * k + (__ffs((int)word) - 1);
            pick = -1;
        } else if (pick >= 0) {
            pick -= c;
        }
    }
    const bool won = apply_action(b, a);
    hb[i] = b.hb;
    vb[i] = b.vb;
    meta[i] = pack_meta(b);
    if (won) {
        value[i] = (int8_t)(winner_of(b) == (int)player0[i] ? 1 : -1);
        done[i] = 1;
        atomicAdd(n_done, 1);
    }
}
__global__ void k_rollout_begin(const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb, const uint64_t* __restrict__ meta, int n,
                                uint8_t* player0, uint8_t* done, int8_t* value, int* n_done) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const Board b = unpack(hb[i], vb[i], meta[i]);
    player0[i] = (uint8_t)b.cur;  // game.get_current_player() at the start (pure_mcts.py:82)
    const int w = winner_of(b);   // a rollout from a finished game ends at once (pure_mcts.py:84-87)
    done[i] = w != 0;
    value[i] = (int8_t)(w == 0 ? 0 : (w == b.cur ? 1 : -1));
    if (w != 0) atomicAdd(n_done, 1);
}

// ---------------------------------------------------------------------------- harvest
// exclusive prefix over finished boards (board order) -> tuple offsets / game ids
__global__ __launch_bounds__(1024) void k_harvest_scan(EngineDev E) {
    __shared__ uint32_t s_p[1024], s_g[1024];
    __shared__ uint32_t base_p, base_g;
    if (threadIdx.x == 0) {
        base_p = 0;
        base_g = 0;
    }
    __syncthreads();
    for (int start = 0; start < E.n_boards; start += 1024) {
        int b = start + (int)threadIdx.x;
        bool fin = b < E.n_boards && E.status[b] == QZ_FINISHED;
        uint32_t np = fin ? E.ply[b] : 0u, ng = fin ? 1u : 0u;
        s_p[threadIdx.x] = np;
        s_g[threadIdx.x] = ng;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t ap = 0, ag = 0;
            if ((int)threadIdx.x >= off) {
                ap = s_p[threadIdx.x - off];
                ag = s_g[threadIdx.x - off];
            }
            __syncthreads();
            s_p[threadIdx.x] += ap;
            s_g[threadIdx.x] += ag;
            __syncthreads();
        }
        if (fin) {
            E.harvest_off[b] = base_p + s_p[threadIdx.x] - np;
            E.harvest_gid[b] = base_g + s_g[threadIdx.x] - ng;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            base_p += s_p[1023];
            base_g += s_g[1023];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        E.counters[QZ_C_GAMES] += base_g;
        E.counters[QZ_C_PENDING_GAMES] = 0;
        E.counters[QZ_C_PENDING_PLIES] = 0;
    }
}

// quoridor.py:596-610: z = +1 where the recorded mover is the winner, else -1; then reset.
// PUSH-ONLY: the finished game's tree and trajectory pages go back to the pools.
__global__ __launch_bounds__(TPB) void k_harvest_copy(EngineDev E, uint64_t* t_hb, uint64_t* t_vb, uint64_t* t_meta,
                                                      float* __restrict__ t_pi, float* __restrict__ t_z,
                                                      int32_t* __restrict__ t_game, int32_t* __restrict__ g_board, long long cap) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    if (rfl(E.status[b]) != QZ_FINISHED) return;
    const uint32_t n = rfl(E.ply[b]), off = rfl(E.harvest_off[b]), gid = rfl(E.harvest_gid[b]);
    const int win = (int)rfl(E.winner[b]);
    if (g_board && lane == 0) g_board[gid] = (int32_t)b;
    const uint32_t* ptab = E.traj_ptab + (size_t)b * QZ_TRAJ_PT;
    uint32_t pgi = 0u, cur = 0u;
    for (uint32_t i = 0; i < n; i++) {
        const long long o = (long long)off + i;
        if (o >= cap) break;
        const uint32_t* page = E.traj_pool + (size_t)rfl(ptab[pgi]) * E.traj_page_dwords;
        if (cur >= E.traj_page_dwords || rfl(page[cur]) == QZ_TRAJ_SKIP) {  // the writer moved on to the next page here
            pgi++;
            cur = 0u;
            page = E.traj_pool + (size_t)rfl(ptab[pgi]) * E.traj_page_dwords;
        }
        const uint32_t* rec = page + cur;
        const int ne = (int)rfl(rec[0]);
        float* row = t_pi + (size_t)o * QZ_N_ACT;
        for (int a = lane; a < QZ_N_ACT; a += 64) row[a] = 0.f;
        wave_sync();
        const uint8_t* rec_act = reinterpret_cast<const uint8_t*>(rec + QZ_TRAJ_HDR + ne);
        for (int k = lane; k < ne; k += 64) row[rec_act[k]] = reinterpret_cast<const float*>(rec)[QZ_TRAJ_HDR + k];
        if (lane == 0) {
            const uint64_t m = (uint64_t)rec[6] | ((uint64_t)rec[7] << 32);
            t_hb[o] = (uint64_t)rec[2] | ((uint64_t)rec[3] << 32);
            t_vb[o] = (uint64_t)rec[4] | ((uint64_t)rec[5] << 32);
            t_meta[o] = m;
            const int mover = (int)((m >> 32) & 0xFF);
            t_z[o] = (mover == win) ? 1.0f : -1.0f;
            if (t_game) t_game[o] = (int32_t)gid;
        }
        cur += QZ_TRAJ_HDR + (uint32_t)ne + (((uint32_t)ne + 3u) >> 2);
    }
    wave_sync();
    wave_free_tree_half(E, b, 0u, lane);
    wave_free_tree_half(E, b, 1u, lane);
    wave_free_traj(E, b, lane);
    if (lane == 0) reset_board_state(E, b);
}

// free lists <- 0..n-1 (engine creation)
__global__ void k_pool_init(EngineDev E) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < E.tree_pool_pages) E.free_tree[i] = (uint32_t)(E.tree_pool_pages - 1 - i);  // low pages are popped first
    if (i < E.traj_pool_pages) E.free_traj[i] = (uint32_t)(E.traj_pool_pages - 1 - i);
    if (i == 0) {
        E.pool_words[QZ_P_TREE_TOP] = E.tree_pool_pages;
        E.pool_words[QZ_P_TREE_LOW] = E.tree_pool_pages;
        E.pool_words[QZ_P_TRAJ_TOP] = E.traj_pool_pages;
        E.pool_words[QZ_P_TRAJ_LOW] = E.traj_pool_pages;
    }
}

__global__ void k_sqrt_table(double* out, int n) {  // self-test helper: device sqrt(double(i))
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) out[i] = sqrt_count((uint32_t)i);  // (what the descents use)
}

#include "qz_rows.h"

}  // namespace

// ============================================================================ launchers
namespace qzl {

static inline dim3 wave_grid(int n) { return dim3((unsigned)((n + WPB - 1) / WPB)); }

#ifndef QZ_NBE
#define QZ_NBE 16
#endif
constexpr int NBE = QZ_NBE;  // boards per encoder group (8 and 32 measured at 32,768 boards: see DESIGN 9.2)

template <int NB>
static void launch_masks_enc(const PoolHand* hands, int n, uint32_t* mask5, const uint64_t* hb,
                             const uint64_t* vb, const uint64_t* meta, const uint8_t* terminal, float* planes, int enc_tile0,
                             int n_enc_groups, hipStream_t s) {
    const int n_mask_groups = mask5 ? (n + NB - 1) / NB : 0;
    if (n_mask_groups + n_enc_groups == 0) return;
    hipLaunchKernelGGL((k_pool_masks_enc<NB, NBE>), dim3((unsigned)(n_mask_groups + n_enc_groups)), dim3(256), 0, s, hands, n,
                       mask5, n_mask_groups, enc_tile0, hb, vb, meta, terminal, planes);
}

size_t movegen_scratch_bytes(int n) { return (size_t)n * sizeof(PoolHand); }  // 184 B per board (round 3: 1,522)

hipError_t movegen_encode(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n, uint32_t* mask5,
                          float* planes, const uint8_t* terminal, void* scratch, const RulesOpts& ro, hipStream_t s, const int* n_dev) {
    if (n <= 0) return hipSuccess;
    if (n_dev && planes) return hipErrorInvalidValue;  // a device-side count: k_wave_rules, legal sets only (no encoder tiles)
    // Small batches are latency-bound: one launch, a wavefront per board, no hand-off through
    // HBM (k_wave_rules).  From ~8k boards on the chip is saturated and the pooled two-launch
    // pipeline, which packs lanes better, wins.
    if (n_dev || (ro.variant >= 2 && ro.variant <= 6) || (ro.variant == 0 && n < 8192)) {
        const int n_enc_groups = planes ? (n + NBE - 1) / NBE : 0;
        // boards per wavefront: on bench trees (late-game boards, many without walls left) one board per
        // wavefront measured 29.1 us vs 33.0 (two) / 34.6 (four) at 4,096 boards; on the synthetic
        // S-mid set two were slightly ahead (40.4 vs 42.9 us).  The in-situ number decides.
        const int G = n_dev ? 1 : (ro.variant == 2 ? 2 : (ro.variant == 4 ? 4 : 1));
        const int n_mg_groups = mask5 ? (n + WPB * G - 1) / (WPB * G) : 0;
        dim3 grid((unsigned)(n_mg_groups + n_enc_groups));
        // one board per wavefront: base paths on nine lanes per player (qz_path_rows.h); variant 5 = the same kernel
        // with one search per lane, kept as its A/B and parity partner
        // the planes go out as streaming (non-temporal) stores: they leave the L2 while the searching wavefronts are still
        // busy instead of in the end-of-kernel write-back (in situ 23.1 -> 21.8 us); variant 6 = ordinary stores (A/B)
        if (G == 1 && ro.variant != 5) hipLaunchKernelGGL((k_wave_rules<NBE, 1, true>), grid, dim3(256), 0, s, hb, vb, meta, n, terminal, mask5, planes, n_mg_groups, ro.detour_wave | (ro.variant == 6 ? 0 : 0x100), n_dev);
        else if (G == 1) hipLaunchKernelGGL((k_wave_rules<NBE, 1, false>), grid, dim3(256), 0, s, hb, vb, meta, n, terminal, mask5, planes, n_mg_groups, ro.detour_wave, n_dev);
        else if (G == 4) hipLaunchKernelGGL((k_wave_rules<NBE, 4, false>), grid, dim3(256), 0, s, hb, vb, meta, n, terminal, mask5, planes, n_mg_groups, ro.detour_wave, n_dev);
        else hipLaunchKernelGGL((k_wave_rules<NBE, 2, false>), grid, dim3(256), 0, s, hb, vb, meta, n, terminal, mask5, planes, n_mg_groups, ro.detour_wave, n_dev);
        return hipGetLastError();
    }
    PoolHand* hands = reinterpret_cast<PoolHand*>(scratch);
    // Encoder tiles are split over the two launches: enc_split_pct percent ride beside the path search (a latency-bound
    // dependent chain of ~27 us that leaves issue slots and the memory pipe idle), the rest in the mask groups' launch.
    // Dependent launches (round 4's sweep at 32,768 boards, S-open / S-mid / S-dense, us): 40 % 67.0 / 74.9 / 79.1, 50 % 66.6 / 72.3 /
    // 76.8, 60 % 67.5 / 71.9 / 78.6, 70 % 67.8 / 73.7 / 80.5.
    const int enc_total = planes ? (n + NBE - 1) / NBE : 0;
    const int enc_a = mask5 ? (enc_total * ro.enc_split_pct) / 100 : 0;
    const int enc_b = enc_total - enc_a;
    const int nbt = ro.variant >= 8 ? ro.variant : (n >= 16384 ? 24 : (n >= 8192 ? 16 : 8));
    if (mask5) {
        const int n_path_groups = (2 * n + 255) / 256;
        hipLaunchKernelGGL((k_pool_paths_enc<NBE>), dim3((unsigned)(n_path_groups + enc_a)), dim3(256), 0, s, hb, vb, meta, n,
                           terminal, hands, n_path_groups, planes, ro.detour_pooled);
    }
    if (nbt >= 32) launch_masks_enc<32>(hands, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else if (nbt >= 24) launch_masks_enc<24>(hands, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else if (nbt >= 16) launch_masks_enc<16>(hands, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else if (nbt >= 12) launch_masks_enc<12>(hands, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else launch_masks_enc<8>(hands, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    return hipGetLastError();
}
hipError_t step(uint64_t* hb, uint64_t* vb, uint64_t* meta, const uint8_t* action, int n, uint8_t* done, uint8_t* winner,
                hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hb, vb, meta, action, n, done, winner);
    return hipGetLastError();
}
hipError_t select(const EngineDev& E, hipStream_t s) {
    hipLaunchKernelGGL(k_select, wave_grid(E.n_boards), dim3(TPB), 0, s, E);
    return hipGetLastError();
}
hipError_t expand_backup(const EngineDev& E, const float* p, const float* v, hipStream_t s) {
    hipLaunchKernelGGL(k_expand_backup, wave_grid(E.n_boards), dim3(TPB), 0, s, E, p, v);
    return hipGetLastError();
}
hipError_t expand_backup_select(const EngineDev& E, const float* p, const float* v, hipStream_t s) {
    hipLaunchKernelGGL(k_expand_backup_select, wave_grid(E.n_boards), dim3(TPB), 0, s, E, p, v);
    return hipGetLastError();
}
hipError_t root_pi(const EngineDev& E, double* pi, int32_t* visits, hipStream_t s) {
    hipLaunchKernelGGL(k_root_pi, wave_grid(E.n_boards), dim3(TPB), 0, s, E, pi, visits);
    return hipGetLastError();
}
hipError_t root_children(const EngineDev& E, int32_t* visits, double* q, float* prior, int32_t* root_visits, hipStream_t s) {
    hipLaunchKernelGGL(k_root_children, wave_grid(E.n_boards), dim3(TPB), 0, s, E, visits, q, prior, root_visits);
    return hipGetLastError();
}
hipError_t update_with_move(const EngineDev& E, const uint8_t* moves, hipStream_t s) {
    hipLaunchKernelGGL(k_update_with_move, wave_grid(E.n_boards), dim3(TPB), 0, s, E, moves);
    hipLaunchKernelGGL(k_release, wave_grid(E.n_boards), dim3(TPB), 0, s, E);
    return hipGetLastError();
}
hipError_t finish_move(const EngineDev& E, const uint8_t* forced, float* pi_out, uint8_t* move_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finish_move, wave_grid(E.n_boards), dim3(TPB), 0, s, E, forced, pi_out, move_out);
    hipLaunchKernelGGL(k_release, wave_grid(E.n_boards), dim3(TPB), 0, s, E);
    return hipGetLastError();
}
// Subtree copies get 1/32 of the round's budget per round (31 us of a millisecond): the moves' launch runs beside / right
// after the network (qz_selfplay_round) and the round's tail waits for it, and the launch lasts at least as long as its
// copies are allowed to -- there is always a board somewhere whose copy is a 1,000-level line (one level per memory round
// trip, ~1 ms in all).  Measured at 8,192 boards: with a quarter the launch ALWAYS ran to its budget (262 us; the trunk
// beside it 191), with an eighth it lasted 200 us of which ~100 stuck out behind the network.  The board concerned sits out
// some thirty rounds instead (a dozen boards of 8,192 at any time); an ordinary compaction (a few thousand edges) still
// fits one slice.
static unsigned int compact_budget(unsigned int budget_ticks) {
    return budget_ticks == 0xFFFFFFFFu ? budget_ticks : (budget_ticks / 32u > 0u ? budget_ticks / 32u : 1u);
}
// one wavefront that sleeps: holds a stream back for `ticks` of s_memrealtime (100 MHz) without taking anything from the chip
__global__ __launch_bounds__(64) void k_hold(unsigned int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
hipError_t advance(const EngineDev& E, int max_iters, unsigned int budget_ticks, int auto_finish, int par, hipStream_t s) {
    if (auto_finish) hipLaunchKernelGGL(k_moves, wave_grid(E.n_boards), dim3(TPB), 0, s, E, compact_budget(budget_ticks));
    if (E.select_opts & 32) {
        // A/B (advance_lanes): k_rows' wavefronts placed BEFORE this launch's, so that they find contiguous registers
        static const int hold_us = getenv("QZ_ADV_HOLD_US") ? atoi(getenv("QZ_ADV_HOLD_US")) : 0;
        if (hold_us > 0) hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, s, (unsigned int)hold_us * 100u);
    }
    const dim3 adv_grid((unsigned)((E.n_boards + ADV_WPB - 1) / ADV_WPB)), adv_block(64 * ADV_WPB);
    if (E.n_boards > 4096 * QZ_ADV_WAVES_SMALL / 4 || (E.select_opts & 4)) hipLaunchKernelGGL(k_advance<QZ_ADV_WAVES_BIG>, adv_grid, adv_block, 0, s, E, max_iters, budget_ticks, par);
    else hipLaunchKernelGGL(k_advance<QZ_ADV_WAVES_SMALL>, adv_grid, adv_block, 0, s, E, max_iters, budget_ticks, par);
    return hipGetLastError();
}
// k_rows (qz_rows.h): the boards without walls, four per wavefront, beside k_advance's launch for the others (select_opts bit 5)
hipError_t advance_lanes(const EngineDev& E, int max_iters, unsigned int budget_ticks, int par, hipStream_t s) {
    {
        // four wavefronts per SIMD (k_rows<4>: 117 registers, no scratch) = 16 boards per SIMD.  (A/B: QZ_ROWS_WEU 3 / 2)
        static const int weu = getenv("QZ_ROWS_WEU") ? atoi(getenv("QZ_ROWS_WEU")) : 4;
        hipError_t me = hipMemsetAsync(E.rows_list, 0, 2 * sizeof(uint32_t), s);  // (the list is rebuilt for every launch: a board must never be listed twice)
        if (me != hipSuccess) return me;
        hipLaunchKernelGGL(k_rows_scout, dim3((unsigned)((E.n_boards + 63) / 64)), dim3(64), 0, s, E);
        // As many wavefronts as find room AT ONCE -- every wavefront of the grid must start at the launch's beginning: the deadline
        // counts from a wavefront's own start --; the boards beyond 4 x that are taken from the queue by rows whose boards left.
        // The chip holds 4,096 wavefronts of 128 registers (benchmarks/hip/residency_census.hip), but this launch starts beside
        // k_advance's for the boards that still have walls: ten thousand 64-register wavefronts come and go in its first tens of
        // microseconds, a register allocation is contiguous, and two 64-register holes left by two of them are not room for a k_rows
        // wavefront -- a grid of 3,584 or 4,096 leaves some hundred wavefronts waiting until the first ones END (launches of 5.9-6.2 ms
        // instead of 3.3).  3,072 (three per SIMD) always find room.  With k_advance's stream held back 40 us (QZ_ADV_HOLD_US, below)
        // 3,840 fit, for +1.5 % (profiles/round6/rows_hold/).  (A/B: QZ_ROWS_WAVES)
        static const int cap_env = getenv("QZ_ROWS_WAVES") ? atoi(getenv("QZ_ROWS_WAVES")) : 0;
        const int cap = cap_env > 0 ? cap_env : (weu >= 3 ? 3072 : 2048);
        const int need = (E.n_boards + rows::NR - 1) / rows::NR;
        const dim3 g((unsigned)(need < cap ? need : cap));
        if (weu >= 4) hipLaunchKernelGGL(k_rows<4>, g, dim3(64), 0, s, E, max_iters, budget_ticks, par);
        else if (weu == 3) hipLaunchKernelGGL(k_rows<3>, g, dim3(64), 0, s, E, max_iters, budget_ticks, par);
        else hipLaunchKernelGGL(k_rows<2>, g, dim3(64), 0, s, E, max_iters, budget_ticks, par);
    }
    return hipGetLastError();
}
// the moves of the boards that have done their playouts + the subtree copies they leave (and the slices earlier moves left)
hipError_t moves(const EngineDev& E, unsigned int budget_ticks, hipStream_t s) {
    hipLaunchKernelGGL(k_moves, dim3((unsigned)E.n_boards), dim3(64), 0, s, E, compact_budget(budget_ticks));
    return hipGetLastError();
}
hipError_t round_tail(const EngineDev& E, int par, hipStream_t s) {
#if QZ_TAIL_COMPACT
    hipLaunchKernelGGL(k_round_tail, wave_grid(TAIL_SLOT_WAVES + (E.n_boards + 63) / 64), dim3(TPB), 0, s, E, par);
#else
    hipLaunchKernelGGL(k_round_tail, wave_grid(2 * E.n_boards), dim3(TPB), 0, s, E, par);
#endif
    return hipGetLastError();
}
hipError_t memo_flush(const EngineDev& E, hipStream_t s) {
    hipLaunchKernelGGL(k_memo_flush, dim3(1), dim3(1), 0, s, E);
    return hipGetLastError();
}
hipError_t reset(const EngineDev& E, int reset_boards, hipStream_t s) {
    hipLaunchKernelGGL(k_reset, wave_grid(E.n_boards), dim3(TPB), 0, s, E, reset_boards);
    return hipGetLastError();
}
hipError_t pool_init(const EngineDev& E, hipStream_t s) {
    const int n = E.tree_pool_pages > E.traj_pool_pages ? E.tree_pool_pages : E.traj_pool_pages;
    hipLaunchKernelGGL(k_pool_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, E);
    return hipGetLastError();
}
hipError_t harvest(const EngineDev& E, uint64_t* t_hb, uint64_t* t_vb, uint64_t* t_meta, float* t_pi, float* t_z,
                   int32_t* t_game, int32_t* g_board, long long cap, hipStream_t s) {
    hipLaunchKernelGGL(k_harvest_scan, dim3(1), dim3(1024), 0, s, E);
    hipLaunchKernelGGL(k_harvest_copy, wave_grid(E.n_boards), dim3(TPB), 0, s, E, t_hb, t_vb, t_meta, t_pi, t_z, t_game, g_board, cap);
    return hipGetLastError();
}
hipError_t rollout_begin(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n, uint8_t* player0, uint8_t* done,
                         int8_t* value, int* n_done, hipStream_t s) {
    hipError_t e = hipMemsetAsync(n_done, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_rollout_begin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hb, vb, meta, n, player0, done, value, n_done);
    return hipGetLastError();
}
hipError_t rollout_step(uint64_t* hb, uint64_t* vb, uint64_t* meta, const uint32_t* mask5, int n, const uint8_t* player0, uint8_t* done,
                        int8_t* value, int* n_done, uint64_t seed, int step, int limit, hipStream_t s) {
    hipLaunchKernelGGL(k_rollout_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hb, vb, meta, mask5, n, player0, done, value,
                       n_done, seed, step, limit);
    return hipGetLastError();
}
#ifdef QZ_RULES_STAMPS
extern "C" int qzt_rules_stamps_read(void* stamps, void* enc, void* rt) {  // [4096][16] u32, [512][2] u64, [4096][2] u64
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_rules_stamps), sizeof(g_rules_stamps));
    if (e == hipSuccess) e = hipMemcpyFromSymbol(enc, HIP_SYMBOL(g_rules_enc), sizeof(g_rules_enc));
    if (e == hipSuccess) e = hipMemcpyFromSymbol(rt, HIP_SYMBOL(g_rules_rt), sizeof(g_rules_rt));
    return (int)e;
}
#endif
#ifdef QZ_ADV_STAMPS
extern "C" int qzt_advance_stamps3_read(void* host_out, int clear) {  // [4096 boards][4] u64
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_adv_stamps3), sizeof(g_adv_stamps3));
    if (e == hipSuccess && clear) {
        static unsigned long long zeros[4096][4];
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_adv_stamps3), zeros, sizeof(zeros));
    }
    return (int)e;
}
extern "C" int qzt_advance_stamps2_read(void* host_out, int clear) {  // [4096 boards][4] u64
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_adv_stamps2), sizeof(g_adv_stamps2));
    if (e == hipSuccess && clear) {
        static unsigned long long zeros[4096][4];
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_adv_stamps2), zeros, sizeof(zeros));
    }
    return (int)e;
}
extern "C" int qzt_advance_stamps_read(void* host_out, int clear) {  // [4096 boards][16] u64
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_adv_stamps), sizeof(g_adv_stamps));
    if (e == hipSuccess && clear) {
        static unsigned long long zeros[4096][16];
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_adv_stamps), zeros, sizeof(zeros));
    }
    return (int)e;
}
#endif
#ifdef QZ_ROWS_STAMPS
extern "C" int qzt_rows_stamps_read(void* host_out, int clear) {  // [24] u64: cycles per section of k_rows, summed over wavefronts and launches
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rows_stamps), sizeof(g_rows_stamps));
    if (e == hipSuccess && clear) {
        static unsigned long long zeros[24];
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamps), zeros, sizeof(zeros));
    }
    return (int)e;
}
#endif
#ifdef QZ_SELECT_STAMPS
extern "C" int qzt_select_stamps_read(void* host_out) {  // [64 launches (playout counter & 63)][4096 boards][8] u32
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sel_stamps), sizeof(g_sel_stamps));
}
#endif
hipError_t sqrt_table(double* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_sqrt_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n);
    return hipGetLastError();
}

}  // namespace qzl

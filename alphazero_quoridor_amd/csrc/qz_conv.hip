// qz_conv.hip -- the trunk convolution of the leaf evaluator on the matrix cores, at fp32 accuracy.
//
// One launch = conv3x3(64 -> 64, pad 1, no bias) + the reference's per-leaf BatchNorm (training
// mode on a batch of one: statistics over the 81 positions of every (leaf, channel),
// policy_value_net.py:20-48,154) [+ residual] [+ ReLU] on channels-last fp32 activations
// x[B][81][64] -> out[B][81][64].  It replaces MIOpen's fp32 implicit GEMM (195 us at B = 4,096:
// 125 of the 157 TFLOP/s the f32 MFMA can do), MIOpen's zero-fill launch (14 us) and the separate
// normalisation pass (26-38 us) of every trunk layer.
//
// fp32 accuracy on the fp16 matrix pipe (16x the f32 MFMA rate): every operand is split in two
// halves, v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (a 22-bit significand: the split loses
// <= 2^-22 |v|, or 2^-25 absolute where lo is an fp16 subnormal), and
//     x*w  ~=  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi          (three MFMAs, fp32 accumulation)
// drops only x_lo*w_lo (<= 2^-22 |x w|).  Error per product ~3 x 2^-22 -- the size of the fp32
// rounding noise of a 576-term dot product; measured against F.conv2d in tests/test_gpu_conv.py.
// The weights are pre-scaled by a power of two (so that w_lo is an fp16 normal) and the
// accumulator is scaled back exactly before the statistics.
//
// Mapping (implicit GEMM, M = positions, N = 64 output channels, K = 9 taps x 64 input channels):
//   workgroup = 256 threads = CS (2) leaves: M = 162 rows in 6 tiles of 32 (rows 162..191 idle)
//   wave w: output channels 32 (w & 1) .. +31, M tiles 3 (w >> 1) .. +2: nine MFMAs
//           (v_mfma_f32_32x32x16_f16) per 16-channel K step, 324 per layer
//   A (activations): staged once per workgroup in LDS as fp16 hi / lo images [leaf][82 rows][64 ch],
//           row 81 all zero (the padding taps point there), 144-byte rows so that 16 consecutive
//           rows cover all 64 banks: a fragment is one conflict-free ds_read_b128 per lane
//   B (weights): [part hi|lo][tap][k chunk][channel out][16 channels in] fp16, 147 KB per layer,
//           read straight from L2 (one coalesced global_load_dwordx4 per fragment)
//   epilogue: two-pass mean / variance over a leaf's 81 rows: in registers, across the two lane
//           halves by shuffle, across the two wave groups through 2 KB of LDS
// No MFMA-free fallback and no library call: this file is plain HIP for gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int CS = 2;        // leaves per workgroup
constexpr int NPOS = 81;
constexpr int ROWS = 82;     // 81 positions + one all-zero row per leaf
constexpr int RSTR = 72;     // halfs per LDS row (144 B)
constexpr int RV = RSTR / 8; // ... in 16-byte vectors
constexpr int C = 64;
constexpr int ZERO_ROW = NPOS;  // of leaf 0

struct ConvShared {
    half8 a_hi[CS * ROWS * RV];  // 16-byte vectors: a fragment is ONE ds_read_b128
    half8 a_lo[CS * ROWS * RV];
    float red[2][2][CS][C];  // [pass][wave group][leaf][channel]
};

// stage CS leaves of x (fp32, [81][64] each) as fp16 hi / lo images
__device__ __forceinline__ void stage_input(ConvShared& sm, const float* __restrict__ x, long long b0, long long n, int tid) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int i = tid; i < CS * NPOS * 16; i += 256) {
        const int s = i / (NPOS * 16), rem = i - s * (NPOS * 16);
        const int p = rem >> 4, c4 = rem & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b0 + s < n) v = x4[((b0 + s) * NPOS + p) * 16 + c4];
        half4 hi, lo;
        hi[0] = (_Float16)v.x;
        hi[1] = (_Float16)v.y;
        hi[2] = (_Float16)v.z;
        hi[3] = (_Float16)v.w;
        lo[0] = (_Float16)(v.x - (float)hi[0]);
        lo[1] = (_Float16)(v.y - (float)hi[1]);
        lo[2] = (_Float16)(v.z - (float)hi[2]);
        lo[3] = (_Float16)(v.w - (float)hi[3]);
        const int o = (s * ROWS + p) * RSTR + c4 * 4;
        *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(sm.a_hi) + o) = hi;
        *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(sm.a_lo) + o) = lo;
    }
    if (tid < CS * 16) {  // the zero rows
        const int s = tid >> 4, c4 = tid & 15;
        half4 z;
        z[0] = z[1] = z[2] = z[3] = (_Float16)0.f;
        const int o = (s * ROWS + NPOS) * RSTR + c4 * 4;
        *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(sm.a_hi) + o) = z;
        *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(sm.a_lo) + o) = z;
    }
}

// w16: [2][9][4][64][16] fp16 (hi part, then lo part)
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_conv3x3_norm(const float* __restrict__ x, const _Float16* __restrict__ w16,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ residual, float* __restrict__ out, long long n,
                                                      const float* __restrict__ inv_scale_p, float eps) {
    __shared__ ConvShared sm;
    const float inv_scale = *inv_scale_p;  // device memory: a captured launch follows the weights (LeafEvaluator.refresh)
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nt = wave & 1, mg = wave >> 1;
    const long long b0 = (long long)blockIdx.x * CS;

    stage_input(sm, x, b0, n, tid);

    // LDS offsets (in 16-byte vectors) of this lane's A rows: [tile][tap]; rows outside the board /
    // past the last leaf read the zero row
    int rowoff[3][9];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int m = 32 * (3 * mg + t) + r;
        const bool live = m < CS * NPOS;
        const int s = m >= NPOS ? 1 : 0, p = m - NPOS * s;
        const int y = p / 9, xx = p - 9 * y;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int yy = y + tap / 3 - 1, x2 = xx + tap % 3 - 1;
            const bool ok = live && (unsigned)yy < 9u && (unsigned)x2 < 9u;
            rowoff[t][tap] = (ok ? (s * ROWS + yy * 9 + x2) : ZERO_ROW) * RV + h;
        }
    }
    floatx16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    __syncthreads();

    // B fragments come straight from L2: this lane's 16 bytes of fragment f = (part, k step) sit at
    // wb[f * 128] (in 16-byte vectors).  They are requested TWO k steps (18 MFMAs, ~600 cycles) ahead
    // into a ring of three register sets, so the matrix pipe never waits for an L2 round trip.
    const half8* wb = reinterpret_cast<const half8*>(w16) + (size_t)(32 * nt + r) * 2 + h;
    constexpr int PARTV = 9 * 4 * C * 2;  // 16-byte vectors per part (hi | lo)
    half8 bh[3], bl[3];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        bh[k] = wb[k * (C * 2)];
        bl[k] = wb[PARTV + k * (C * 2)];
    }
#pragma unroll
    for (int k = 0; k < 36; k++) {  // k = 4 tap + (16-channel chunk)
        if (k + 2 < 36) {
            bh[(k + 2) % 3] = wb[(k + 2) * (C * 2)];
            bl[(k + 2) % 3] = wb[PARTV + (k + 2) * (C * 2)];
        }
        asm volatile("" ::: "memory");  // the prefetch stays ahead of this step's LDS reads
        const int tap = k >> 2, kc = k & 3;
        const half8 b_hi = bh[k % 3], b_lo = bl[k % 3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const half8 a_hi = sm.a_hi[rowoff[t][tap] + 2 * kc];
            const half8 a_lo = sm.a_lo[rowoff[t][tap] + 2 * kc];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[t], 0, 0, 0);
        }
    }

    // ---- epilogue: this lane owns channel co, rows m(t, i) = 32 (3 mg + t) + (i & 3) + 8 (i >> 2) + 4 h.
    // Branch-free on purpose: selects for the statistics, and the 16 residual loads / 16 stores of a
    // tile are issued as batches (element (leaf s, position p, channel co) of the output sits at
    // (b0 * 81 + m) * 64 + co whichever leaf row m belongs to).
    const int co = 32 * nt + r;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = 32 * (3 * mg + t) + (i & 3) + 8 * (i >> 2) + 4 * h;
            const float y = acc[t][i] * inv_scale;  // exact: a power of two
            acc[t][i] = y;
            s0 += m < NPOS ? y : 0.f;
            s1 += (m >= NPOS && m < 2 * NPOS) ? y : 0.f;
        }
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    if (h == 0) {
        sm.red[0][mg][0][co] = s0;
        sm.red[0][mg][1][co] = s1;
    }
    __syncthreads();
    const float mean0 = (sm.red[0][0][0][co] + sm.red[0][1][0][co]) * (1.0f / 81.0f);
    const float mean1 = (sm.red[0][0][1][co] + sm.red[0][1][1][co]) * (1.0f / 81.0f);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = 32 * (3 * mg + t) + (i & 3) + 8 * (i >> 2) + 4 * h;
            const float d0 = acc[t][i] - mean0, d1 = acc[t][i] - mean1;
            q0 += m < NPOS ? d0 * d0 : 0.f;
            q1 += (m >= NPOS && m < 2 * NPOS) ? d1 * d1 : 0.f;
        }
    q0 += __shfl_xor(q0, 32, 64);
    q1 += __shfl_xor(q1, 32, 64);
    if (h == 0) {
        sm.red[1][mg][0][co] = q0;
        sm.red[1][mg][1][co] = q1;
    }
    __syncthreads();
    const float g = gamma[co], bt = beta[co];
    const float k0 = g / sqrtf((sm.red[1][0][0][co] + sm.red[1][1][0][co]) * (1.0f / 81.0f) + eps);
    const float k1 = g / sqrtf((sm.red[1][0][1][co] + sm.red[1][1][1][co]) * (1.0f / 81.0f) + eps);
    const size_t obase = (size_t)b0 * NPOS * C + (size_t)co;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int tile = 3 * mg + t;  // wave-uniform
        const int m0 = 32 * tile + 4 * h;
        if (tile < 5 && b0 + CS <= n) {  // every row of the tile is a real position of a real leaf
            float rv[16];
            if (RES) {
#pragma unroll
                for (int i = 0; i < 16; i++) rv[i] = residual[obase + (size_t)(m0 + (i & 3) + 8 * (i >> 2)) * C];
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = m0 + (i & 3) + 8 * (i >> 2);
                float v = (acc[t][i] - (m >= NPOS ? mean1 : mean0)) * (m >= NPOS ? k1 : k0) + bt;
                if (RES) v += rv[i];
                if (RELU) v = fmaxf(v, 0.f);
                out[obase + (size_t)m * C] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = m0 + (i & 3) + 8 * (i >> 2);
                const int sidx = m >= NPOS ? 1 : 0;
                if (m < 2 * NPOS && b0 + sidx < n) {
                    float v = (acc[t][i] - (sidx ? mean1 : mean0)) * (sidx ? k1 : k0) + bt;
                    if (RES) v += residual[obase + (size_t)m * C];
                    if (RELU) v = fmaxf(v, 0.f);
                    out[obase + (size_t)m * C] = v;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// k_trunk: the WHOLE residual trunk (2 n_blocks layers) in one launch.  A workgroup of TWO waves
// keeps ONE leaf on the CU for all layers:
//   * wave w owns output channels 32 w .. 32 w + 31 of all 81 positions (three 32-row tiles, rows
//     81..95 idle), so the per-leaf statistics never leave the wave: registers + one shuffle
//     between the lane halves, no LDS, no barrier;
//   * the next layer's input never leaves LDS: the epilogue converts the normalised activations to
//     fp16 hi / lo in registers, pairs of neighbouring channels meet by one DPP move and go
//     straight into the two images as packed 32-bit stores;
//   * the residual of a block never leaves the registers: the lane that owns output element
//     (position, channel) of one layer owns it in every layer, so the block input it needs two
//     layers later is a value it produced itself.
// Two barriers per layer (both waves are done reading the images / the new images are complete),
// between two waves only.  HBM traffic: the first layer's input once, the last layer's output
// once: 170 MB per 4,096 leaves instead of 2.1 GB for ten separate launches -- whose workgroups
// all start together, stage together (HBM burst, matrix pipe idle), multiply together (HBM idle)
// and store together: the layer-by-layer route spends half its time in those bursts.
constexpr int MAX_TRUNK_LAYERS = 16;
struct TrunkArgs {
    const _Float16* w16[MAX_TRUNK_LAYERS];
    const float* gamma[MAX_TRUNK_LAYERS];
    const float* beta[MAX_TRUNK_LAYERS];
    const float* inv_scale;  // [dev] [n_layers] (+ the head stage's at [n_layers]): 1 / the power-of-two scale of each layer's weight image
};
struct TrunkShared {
    half8 a_hi[ROWS * RV];  // one leaf: 81 positions + the all-zero row, 144-byte rows
    half8 a_lo[ROWS * RV];
    float red6[2][2][32];   // head stage: [pass][wave][head channel]
    // input stage: tables and the board's pixel codes (its raw fp32 output [81][64] lies in the image memory)
    float wd[4 * 9 * 64];
    float s9[9 * 64];
    uint8_t code[96];
};
// The head stage (optional): the merged 64 -> 6 head convolution (policy_value_net.py:64-65,69-70:
// conv2 = value channels 0..3, conv3 = policy channels 4..5) + bn2 / bn3 per leaf + ReLU on the
// last layer's activations while they are still in LDS, as one more implicit GEMM with a 32-column
// B tile of which 6 columns are real.  feat [n][486] (c * 81 + pos) feeds k_head_fc (qz_nn.hip).
struct HeadArgs {
    const _Float16* w6;   // [2][9][4][32][16] fp16 hi | lo of W6 * scale, columns 6..31 zero
    const float* gamma6;  // [6]
    const float* beta6;   // [6]
    float* feat;          // [n][486] out; nullptr = no head stage (its 1 / scale is TrunkArgs.inv_scale[n_layers])
    int role_shift;       // head stage: workgroups with bit `role_shift` of their index set give the two-tile share to wave 1 (>= 31: never)
};

// The input stage (optional): the first layer conv1(state(board)) + bn1 per leaf + ReLU computed from the packed
// board and three tables instead of being read from HBM -- k_input_layer's arithmetic (qz_nn.hip: <= 3 all-ones planes
// by border class, the empty wall grid, <= 22 non-zero pixels) in k_trunk's register layout, followed by the ordinary
// layer epilogue (statistics, ReLU, hand-over into the LDS images).
struct InputArgs {
    const uint64_t *hb, *vb, *meta;  // packed boards; hb == nullptr: no input stage, the trunk input is read from x
    const uint8_t* terminal;         // or nullptr
    const float *hot9, *base0, *wd;  // [21][9][64], [81][64], [4][9][64] (qz_nn_input_layer)
    const float *gamma0, *beta0;     // bn1
};
// SPLIT = true: every product as three MFMAs on split fp16 operands (fp32 accuracy, the parity mode, the default).
// SPLIT = false: ONE MFMA per product on the hi halves only (fp16 operands, fp32 accumulation) -- the labelled
// throughput mode (`nn_precision="fp16"`, bench.py --nn-dtype fp16): a third of the matrix work, p / v within ~1e-3
// of the reference instead of 1e-5 (tests/test_gpu_conv.py states the bound); never the default.
template <bool FROM_BOARD, bool SPLIT = true>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_trunk(const float* __restrict__ x, float* __restrict__ out, TrunkArgs A, int n_layers, float eps, HeadArgs H, InputArgs I,
                                               const int* __restrict__ n_live  // or nullptr: only the first *n_live leaves are evaluated (the engine's miss list)
#ifdef QZ_TRUNK_STAMPS
                                               , unsigned long long* stamps  // [workgroup][wave][8]: diagnostic build only
#endif
                                               ) {
    __shared__ TrunkShared sm;
    // one workgroup per leaf (a grid of a few persistent workgroups striding over the list measured 3 % slower: profiles/round5/SUMMARY.md 6);
    // the count of live leaves may sit on the device (the engine's miss list): workgroups beyond it leave at once
    const int n_lv = n_live ? __builtin_amdgcn_readfirstlane(*n_live) : 0x7FFFFFFF;
  {
    const unsigned int leaf = blockIdx.x;
    if ((int)leaf >= n_lv) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, nt = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int co = 32 * nt + r;
    const size_t obase = (size_t)leaf * NPOS * C + (size_t)co;
#ifdef QZ_TRUNK_STAMPS
    unsigned long long t_stage = 0, t_loop = 0, t_stat = 0, t_hand = 0, t_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_mark, r_begin = __builtin_amdgcn_s_memrealtime();
#define QZ_STAMP(acc_var) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_var += now_ - t_mark; t_mark = now_; }
#else
#define QZ_STAMP(acc_var)
#endif
    constexpr bool from_board = FROM_BOARD;
    float* const s_wd = sm.wd;
    float* const s_s9 = sm.s9;
    uint8_t* const s_code = sm.code;
    float* const s_raw = reinterpret_cast<float*>(sm.a_hi);  // input stage: conv1 before bn1, fp32 [81][64] (20,736 of the images' 23,616 B)
    bool term = false;
    if (!from_board) {   // stage the leaf (fp32 [81][64]) as fp16 hi / lo images
        const float4* x4 = reinterpret_cast<const float4*>(x) + (size_t)leaf * NPOS * 16;
        _Float16* ih = reinterpret_cast<_Float16*>(sm.a_hi);
        _Float16* il = reinterpret_cast<_Float16*>(sm.a_lo);
        for (int i = tid; i < NPOS * 16; i += 128) {
            const int p = i >> 4, c4 = i & 15;
            const float4 v = x4[i];
            half4 hi, lo;
            hi[0] = (_Float16)v.x;
            hi[1] = (_Float16)v.y;
            hi[2] = (_Float16)v.z;
            hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)(v.x - (float)hi[0]);
            lo[1] = (_Float16)(v.y - (float)hi[1]);
            lo[2] = (_Float16)(v.z - (float)hi[2]);
            lo[3] = (_Float16)(v.w - (float)hi[3]);
            *reinterpret_cast<half4*>(ih + p * RSTR + c4 * 4) = hi;
            *reinterpret_cast<half4*>(il + p * RSTR + c4 * 4) = lo;
        }
        if (tid < 16) {
            half4 z;
            z[0] = z[1] = z[2] = z[3] = (_Float16)0.f;
            *reinterpret_cast<half4*>(ih + NPOS * RSTR + tid * 4) = z;
            *reinterpret_cast<half4*>(il + NPOS * RSTR + tid * 4) = z;
        }
    } else {
        // packed board, include/qz_abi.h: meta = p1 i8 | p2 i8 | walls1 u8 | walls2 u8 | current player u8
        const uint64_t m = I.meta[leaf], bhb = I.hb[leaf], bvb = I.vb[leaf];
        const int p1 = (int)(int8_t)(m & 0xFF), p2 = (int)(int8_t)((m >> 8) & 0xFF);
        const int w1 = (int)((m >> 16) & 0xFF), w2 = (int)((m >> 24) & 0xFF), cur = (int)((m >> 32) & 0xFF);
        term = I.terminal ? (I.terminal[leaf] != 0) : false;
        for (int q = tid; q < 576; q += 128) reinterpret_cast<float4*>(s_wd)[q] = reinterpret_cast<const float4*>(I.wd)[q];
        {
            const int wm = cur == 1 ? w1 : w2, wo = cur == 1 ? w2 : w1;
            int im = wm - 1, io = wo - 1;  // Python index -1 -> last plane (quoridor.py:79-80)
            if (im < 0) im += 10;
            if (io < 0) io += 10;
            const int h0 = im, h1 = 10 + io;
            for (int e = tid; e < 576; e += 128) {
                const int cls = e >> 6, c = e & 63;
                float v = I.hot9[(h0 * 9 + cls) * 64 + c] + I.hot9[(h1 * 9 + cls) * 64 + c];
                if (cur == 2) v += I.hot9[(20 * 9 + cls) * 64 + c];
                s_s9[e] = v;
            }
        }
        if (tid < 81) {
            const int rr = tid / 9, cc = tid - 9 * rr;
            uint32_t code = 0u;
            if (rr < 8 && cc < 8) {
                const int ix = 8 * rr + cc;
                code = (uint32_t)((bhb >> ix) & 1ull) | ((uint32_t)((bvb >> ix) & 1ull) << 1);
            }
            int pm = cur == 1 ? p1 : p2, po = cur == 1 ? p2 : p1;
            if (pm < 0) pm += 81;
            if (po < 0) po += 81;
            code |= (tid == pm ? 4u : 0u) | (tid == po ? 8u : 0u);
            s_code[tid] = (uint8_t)code;
        }
        __syncthreads();
        // thread = (channel quad, positions p = (tid >> 4) + 8 k): the additions in k_input_layer's order
        const int cq = tid & 15;
#pragma unroll 1
        for (int p = tid >> 4; p < NPOS; p += 8) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!term) {
                const int y = p / 9, xx = p - 9 * y;
                const int cls = (y == 0 ? 0 : (y == 8 ? 2 : 1)) * 3 + (xx == 0 ? 0 : (xx == 8 ? 2 : 1));
                const float4 s4 = reinterpret_cast<const float4*>(s_s9)[cls * 16 + cq], b4 = reinterpret_cast<const float4*>(I.base0)[p * 16 + cq];
                a = make_float4(s4.x + b4.x, s4.y + b4.y, s4.z + b4.z, s4.w + b4.w);
                for (int dy = -1; dy <= 1; dy++) {
                    const int ny = y + dy;
                    if (ny < 0 || ny > 8) continue;
                    for (int dx = -1; dx <= 1; dx++) {
                        const int nx = xx + dx;
                        if (nx < 0 || nx > 8) continue;
                        const uint32_t code = s_code[ny * 9 + nx];
                        if (code == 0u) continue;
                        const int tap = (dy + 1) * 3 + (dx + 1);
#pragma unroll
                        for (int kind = 0; kind < 4; kind++)
                            if ((code >> kind) & 1u) {
                                const float4 w4 = reinterpret_cast<const float4*>(s_wd)[(kind * 9 + tap) * 16 + cq];
                                a.x += w4.x; a.y += w4.y; a.z += w4.z; a.w += w4.w;
                            }
                    }
                }
            }
            reinterpret_cast<float4*>(s_raw)[p * 16 + cq] = a;
        }
    }
    // this lane's A rows per tile (row m = 32 t + r): vector offset of the position and which of the 9 taps
    // stay on the board (the others, and rows 81..95, read the zero row)
    int rbase[3];
    uint32_t vmask[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int m = 32 * t + r;
        const int y = m / 9, xx = m - 9 * y;
        rbase[t] = m * RV + h;
        uint32_t vm = 0u;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int yy = y + tap / 3 - 1, x2 = xx + tap % 3 - 1;
            if (m < NPOS && (unsigned)yy < 9u && (unsigned)x2 < 9u) vm |= 1u << tap;
        }
        vmask[t] = vm;
    }
    // the block input at this lane's own output elements (= the trunk input for the first block);
    // output element i of tile t is row 32 t + 4 h + (i & 3) + 8 (i >> 2), channel co
    floatx16 resid[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = 32 * t + 4 * h + (i & 3) + 8 * (i >> 2);
            resid[t][i] = (m < NPOS && !from_board) ? x[obase + (size_t)m * C] : 0.f;
        }
    __syncthreads();
    QZ_STAMP(t_stage)

    constexpr int PARTV = 9 * 4 * C * 2;
#pragma unroll 1
    for (int l = from_board ? -1 : 0; l < n_layers; l++) {  // l = -1: the input stage takes the place of the matrix loop
        floatx16 acc[3];
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
        if (l < 0) {
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int m = 32 * t + 4 * h + (i & 3) + 8 * (i >> 2);
                    if (m < NPOS) acc[t][i] = s_raw[m * 64 + co];
                }
        } else {
        const half8* wb = reinterpret_cast<const half8*>(A.w16[l]) + (size_t)(32 * nt + r) * 2 + h;
        // B ring of three register sets, filled two k steps ahead (the loop of k_conv3x3_norm, fully unrolled)
        half8 bh[3], bl[3];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            bh[k] = wb[k * (C * 2)];
            bl[k] = wb[PARTV + k * (C * 2)];
        }
        // A fragments: the reads of the NEXT tile are issued before the MFMAs of this one (two register sets), so the
        // ~120-cycle LDS latency runs under 96 cycles of matrix work instead of in front of it; the scheduling fences
        // keep the compiler from folding the two sets back into one
        int ro[9][3];
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int delta = ((tap / 3 - 1) * 9 + (tap % 3 - 1)) * RV;
#pragma unroll
            for (int t = 0; t < 3; t++) ro[tap][t] = ((vmask[t] >> tap) & 1u) ? rbase[t] + delta : ZERO_ROW * RV + h;
        }
        half8 ah[2], al[2];
        ah[0] = sm.a_hi[ro[0][0]];
        al[0] = sm.a_lo[ro[0][0]];
#pragma unroll
        for (int k = 0; k < 36; k++) {
            const int tap = k >> 2, kc = k & 3;
            if (k + 2 < 36) {
                bh[(k + 2) % 3] = wb[(k + 2) * (C * 2)];
                bl[(k + 2) % 3] = wb[PARTV + (k + 2) * (C * 2)];
            }
            const half8 b_hi = bh[k % 3], b_lo = bl[k % 3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int cur = (3 * k + t) & 1, nxt = cur ^ 1;
                if (t < 2) {
                    ah[nxt] = sm.a_hi[ro[tap][t + 1] + 2 * kc];
                    al[nxt] = sm.a_lo[ro[tap][t + 1] + 2 * kc];
                } else if (k + 1 < 36) {
                    ah[nxt] = sm.a_hi[ro[(k + 1) >> 2][0] + 2 * ((k + 1) & 3)];
                    al[nxt] = sm.a_lo[ro[(k + 1) >> 2][0] + 2 * ((k + 1) & 3)];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], b_hi, acc[t], 0, 0, 0);
                if (SPLIT) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], b_lo, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], b_hi, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        QZ_STAMP(t_loop)
        // ---- per-leaf statistics of channel co over the 81 rows: all of them live in this wave
        const float inv_scale = l >= 0 ? A.inv_scale[l] : 1.0f;
        float s0 = 0.f;
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float y = acc[t][i] * inv_scale;  // exact: a power of two
                acc[t][i] = y;
                s0 += m < NPOS ? y : 0.f;
            }
        s0 += __shfl_xor(s0, 32, 64);
        const float mean = s0 * (1.0f / 81.0f);
        float q0 = 0.f;
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float d = acc[t][i] - mean;
                q0 += m < NPOS ? d * d : 0.f;
            }
        q0 += __shfl_xor(q0, 32, 64);
        const float bt = (l >= 0 ? A.beta[l] : I.beta0)[co];
        const float kn = (l >= 0 ? A.gamma[l] : I.gamma0)[co] / sqrtf(q0 * (1.0f / 81.0f) + eps);
        QZ_STAMP(t_stat)
        const bool second = (l & 1) != 0;  // conv2 of a block: + block input, result = next block input (the input stage: 0 + its output)
        const bool last = l == n_layers - 1;
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float v = (acc[t][i] - mean) * kn + bt;
                if (second) v += resid[t][i];
                v = fmaxf(v, 0.f);
                if (second) resid[t][i] = v;
                acc[t][i] = v;
            }
        if (last) {
            if (out) {
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int m = 32 * t + 4 * h + (i & 3) + 8 * (i >> 2);
                        if (m < NPOS) out[obase + (size_t)m * C] = acc[t][i];
                    }
            }
            if (!H.feat) break;  // else: the trunk output goes to the images once more, for the head stage
        }
        __syncthreads();  // both waves are done reading this layer's input images
        if (l < 0 && tid < 16) {  // (the input stage kept its raw output where the zero rows belong)
            half4 z;
            z[0] = z[1] = z[2] = z[3] = (_Float16)0.f;
            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(sm.a_hi) + NPOS * RSTR + tid * 4) = z;
            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(sm.a_lo) + NPOS * RSTR + tid * 4) = z;
        }
        // hand-over: v -> (hi, lo) fp16; the lane pair (co even, co + 1) swaps one packed word by DPP, the even
        // lane stores both hi halves into a_hi, the odd lane both lo halves into a_lo: one 32-bit store each
        // (two 16-bit stores per element and no DPP: 4.8 k instead of 5.8 k cycles here, but the MFMA loop of the other wave
        // gets 0.4 k longer: no gain)
        uint32_t* img = reinterpret_cast<uint32_t*>((lane & 1) ? static_cast<void*>(sm.a_lo) : static_cast<void*>(sm.a_hi));
        const int cpair = (co & ~1) >> 1;
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = 32 * t + 4 * h + (i & 3) + 8 * (i >> 2);
                const float v = acc[t][i];
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                const uint32_t mine = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
                const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);  // lane ^ 1
                // even lane: (my hi, partner's hi); odd lane: (partner's lo, my lo) -- channel co & ~1 in the low half
                const uint32_t word = (lane & 1) ? ((other >> 16) | (mine & 0xFFFF0000u)) : ((mine & 0xFFFFu) | (other << 16));
                if (m < NPOS) img[m * (RSTR / 2) + cpair] = word;
            }
        __syncthreads();  // the new images are complete before anybody reads them
        QZ_STAMP(t_hand)
    }
    if (H.feat) {
        // ---- head stage.  M tiles: wave 0 takes rows 0..63 (two tiles), wave 1 rows 64..80 (one); lane (r, h)
        // ends up with head channel r (r < 6 are real) of its tiles' rows
        // (the wave with two tiles has twice the matrix work: alternating that role between the waves from workgroup to
        // workgroup spreads it over the SIMDs of a CU)
        const int role = nt ^ (int)((H.role_shift < 31) ? ((leaf >> H.role_shift) & 1u) : 0u);
        const bool two = __builtin_amdgcn_readfirstlane(role) == 0;
        const int rb0 = role ? rbase[2] : rbase[0], rb1 = rbase[1];
        const uint32_t vm0 = role ? vmask[2] : vmask[0], vm1 = vmask[1];
        constexpr int P6 = 36 * 32 * 2;
        const half8* wb6 = reinterpret_cast<const half8*>(H.w6) + (size_t)r * 2 + h;
        floatx16 ha0, ha1;
#pragma unroll
        for (int i = 0; i < 16; i++) ha0[i] = ha1[i] = 0.f;
        half8 bh[3], bl[3];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            bh[k] = wb6[k * 64];
            bl[k] = wb6[P6 + k * 64];
        }
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int delta = ((tap / 3 - 1) * 9 + (tap % 3 - 1)) * RV;
            const int ro0 = ((vm0 >> tap) & 1u) ? rb0 + delta : ZERO_ROW * RV + h;
            const int ro1 = ((vm1 >> tap) & 1u) ? rb1 + delta : ZERO_ROW * RV + h;
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const int k = 4 * tap + kc;
                if (k + 2 < 36) {
                    bh[(k + 2) % 3] = wb6[(k + 2) * 64];
                    bl[(k + 2) % 3] = wb6[P6 + (k + 2) * 64];
                }
                asm volatile("" ::: "memory");
                const half8 b_hi = bh[k % 3], b_lo = bl[k % 3];
                {
                    const half8 a_hi = sm.a_hi[ro0 + 2 * kc];
                    const half8 a_lo = sm.a_lo[ro0 + 2 * kc];
                    ha0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, ha0, 0, 0, 0);
                    if (SPLIT) {
                        ha0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, ha0, 0, 0, 0);
                        ha0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, ha0, 0, 0, 0);
                    }
                }
                if (two) {
                    const half8 a_hi = sm.a_hi[ro1 + 2 * kc];
                    const half8 a_lo = sm.a_lo[ro1 + 2 * kc];
                    ha1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, ha1, 0, 0, 0);
                    if (SPLIT) {
                        ha1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, ha1, 0, 0, 0);
                        ha1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, ha1, 0, 0, 0);
                    }
                }
            }
        }
        // per-leaf statistics of head channel r over the 81 rows: this wave's rows, the lane halves by shuffle,
        // the two waves through LDS
        const int row0 = (role ? 64 : 0) + 4 * h;
        const float is6 = A.inv_scale[n_layers];
        float s0 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = row0 + (i & 3) + 8 * (i >> 2);
            ha0[i] *= is6;
            ha1[i] *= is6;
            s0 += m < NPOS ? ha0[i] : 0.f;
            if (two) s0 += ha1[i];  // rows 32..63
        }
        s0 += __shfl_xor(s0, 32, 64);
        if (h == 0) sm.red6[0][role][r] = s0;
        __syncthreads();  // also: both waves are done reading the images
        const float mean = (sm.red6[0][0][r] + sm.red6[0][1][r]) * (1.0f / 81.0f);
        float q0 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = row0 + (i & 3) + 8 * (i >> 2);
            const float d0 = ha0[i] - mean, d1 = ha1[i] - mean;
            q0 += m < NPOS ? d0 * d0 : 0.f;
            if (two) q0 += d1 * d1;
        }
        q0 += __shfl_xor(q0, 32, 64);
        if (h == 0) sm.red6[1][role][r] = q0;
        __syncthreads();
        float* fb = reinterpret_cast<float*>(sm.a_hi);  // 486 floats; the images are dead
        if (r < 6) {
            const float var = (sm.red6[1][0][r] + sm.red6[1][1][r]) * (1.0f / 81.0f);
            const float kn = H.gamma6[r] / sqrtf(var + eps), bt = H.beta6[r];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = row0 + (i & 3) + 8 * (i >> 2);
                if (m < NPOS) fb[r * NPOS + m] = fmaxf((ha0[i] - mean) * kn + bt, 0.f);
                if (two) fb[r * NPOS + m + 32] = fmaxf((ha1[i] - mean) * kn + bt, 0.f);
            }
        }
        __syncthreads();
        float* fo = H.feat + (size_t)leaf * (6 * NPOS);
        for (int i = tid; i < 6 * NPOS; i += 128) fo[i] = fb[i];
    }
#ifdef QZ_TRUNK_STAMPS
    if (lane == 0) {
        unsigned long long* o = stamps + ((size_t)leaf * 2 + nt) * 8;
        o[0] = t_stage; o[1] = t_loop; o[2] = t_stat; o[3] = t_hand; o[4] = __builtin_amdgcn_s_memtime() - t_begin; o[5] = t_begin;
        o[6] = __builtin_amdgcn_s_memrealtime() - r_begin; o[7] = __smid();  // o[4] / o[6] x 100 MHz = the clock this wave ran at
    }
#endif
  }
}

}  // namespace

namespace qzl {
hipError_t conv3x3_norm(const float* x, const void* w16, const float* gamma, const float* beta, const float* residual, float* out,
                        long long n, const float* inv_scale, int relu, float eps, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + CS - 1) / CS));
    const _Float16* w = reinterpret_cast<const _Float16*>(w16);
    if (residual && relu) hipLaunchKernelGGL((k_conv3x3_norm<true, true>), grid, dim3(256), 0, s, x, w, gamma, beta, residual, out, n, inv_scale, eps);
    else if (residual) hipLaunchKernelGGL((k_conv3x3_norm<true, false>), grid, dim3(256), 0, s, x, w, gamma, beta, residual, out, n, inv_scale, eps);
    else if (relu) hipLaunchKernelGGL((k_conv3x3_norm<false, true>), grid, dim3(256), 0, s, x, w, gamma, beta, residual, out, n, inv_scale, eps);
    else hipLaunchKernelGGL((k_conv3x3_norm<false, false>), grid, dim3(256), 0, s, x, w, gamma, beta, residual, out, n, inv_scale, eps);
    return hipGetLastError();
}
// the input stage's arguments as the C ABI hands them over (qz_abi.hip declares the same struct)
struct TrunkInput {
    const uint64_t *hb, *vb, *meta;
    const uint8_t* terminal;
    const float *hot9, *base0, *wd, *gamma0, *beta0;
};
// feat != nullptr (fused route only): [w6 fp16, gamma6, beta6, feat out] + inv_scale6: the head stage runs in the same
// launch and the trunk output is NOT written back to x.  in != nullptr (fused route only): the first layer is computed
// from the packed boards in the same launch and x is not read (may be NULL)
hipError_t trunk(float* x, float* tmp, long long n, int n_blocks, const void* const* w16, const float* const* gamma, const float* const* beta,
                 const float* inv_scale /*[dev]*/, float eps, int fused, hipStream_t s, const void* w6 = nullptr, const float* gamma6 = nullptr,
                 const float* beta6 = nullptr, float* feat = nullptr, const TrunkInput* in = nullptr, const int* n_live = nullptr, int single_product = 0) {
    if (n <= 0 || n_blocks <= 0) return hipSuccess;
    const int nl = 2 * n_blocks;
    if ((feat || in || n_live) && !(fused && nl <= MAX_TRUNK_LAYERS)) return hipErrorInvalidValue;
    if (fused && nl <= MAX_TRUNK_LAYERS) {  // one persistent launch: activations stay on the CU
        TrunkArgs A;
        for (int l = 0; l < nl; l++) {
            A.w16[l] = reinterpret_cast<const _Float16*>(w16[l]);
            A.gamma[l] = gamma[l];
            A.beta[l] = beta[l];
        }
        for (int l = nl; l < MAX_TRUNK_LAYERS; l++) {
            A.w16[l] = nullptr;
            A.gamma[l] = A.beta[l] = nullptr;
        }
        A.inv_scale = inv_scale;
#ifdef QZ_TRUNK_STAMPS
        return hipErrorNotSupported;  // the diagnostic build launches the kernel itself (tests/hip/qz_conv_stamps.hip)
#else
        HeadArgs H;
        H.w6 = reinterpret_cast<const _Float16*>(w6);
        H.gamma6 = gamma6;
        H.beta6 = beta6;
        H.feat = feat;
        H.role_shift = 0;  // odd workgroups swap the roles: 787 -> 779 us for 4,096 leaves (shifts 1..8 and none measured: 783..791)
        InputArgs I = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (in) I = InputArgs{in->hb, in->vb, in->meta, in->terminal, in->hot9, in->base0, in->wd, in->gamma0, in->beta0};
        if (single_product && !in) return hipErrorInvalidValue;  // (the throughput mode exists on the route from the packed boards only)
        if (in && single_product) hipLaunchKernelGGL((k_trunk<true, false>), dim3((unsigned)n), dim3(128), 0, s, x, feat ? nullptr : x, A, nl, eps, H, I, n_live);
        else if (in) hipLaunchKernelGGL((k_trunk<true, true>), dim3((unsigned)n), dim3(128), 0, s, x, feat ? nullptr : x, A, nl, eps, H, I, n_live);
        else hipLaunchKernelGGL((k_trunk<false, true>), dim3((unsigned)n), dim3(128), 0, s, x, feat ? nullptr : x, A, nl, eps, H, I, n_live);
        return hipGetLastError();
#endif
    }
    for (int b = 0; b < n_blocks; b++) {
        // y = relu(bn1(conv1(x)));  x = relu(bn2(conv2(y)) + x)   (policy_value_net.py:33-48)
        hipError_t e = conv3x3_norm(x, w16[2 * b], gamma[2 * b], beta[2 * b], nullptr, tmp, n, inv_scale + 2 * b, 1, eps, s);
        if (e != hipSuccess) return e;
        e = conv3x3_norm(tmp, w16[2 * b + 1], gamma[2 * b + 1], beta[2 * b + 1], x, x, n, inv_scale + 2 * b + 1, 1, eps, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace qzl

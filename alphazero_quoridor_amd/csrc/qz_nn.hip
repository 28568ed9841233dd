// qz_nn.hip -- glue kernel for the leaf evaluator (the network itself stays in PyTorch-ROCm /
// MIOpen): the reference evaluates every leaf with BatchNorm in TRAINING mode on a batch of
// one (policy_value_net.py:154, no .eval() anywhere), i.e. every (sample, channel) 9x9 plane
// is normalised with its own mean / biased variance.  Stock PyTorch reaches that only through
// instance_norm -> batch_norm over B*C pseudo-channels plus separate residual-add and ReLU
// passes (~45 % of the step on MI355X).  This kernel does  out = act(gamma*(x-mean)*rstd + beta
// [+ residual])  in ONE pass: a 256-thread workgroup stages 64 planes (20.7 KB, 16-byte
// coalesced loads) in LDS, four lanes reduce each plane (two-pass mean / variance, fp32), and
// the result leaves with 16-byte coalesced stores.  HBM-bound: 2 x 324 B per plane (+324 B
// with a residual).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int PL = 81;            // elements per plane (9x9)
constexpr int PPB = 64;           // planes per workgroup
constexpr int NF = PL * PPB;      // 5184 floats = 1296 float4 per workgroup

template <bool HAS_RES, bool RELU>
__global__ __launch_bounds__(256) void k_instnorm_act(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ res,
                                                      float* __restrict__ out, long long n_planes, int C, float eps) {
    __shared__ float s_x[NF];
    __shared__ float s_scale[PPB], s_shift[PPB];
    const int tid = (int)threadIdx.x;
    const long long p0 = (long long)blockIdx.x * PPB;
    const int np = (int)((n_planes - p0) < PPB ? (n_planes - p0) : PPB);
    const int nf = np * PL;
    const float* xin = x + p0 * PL;
    const bool full = np == PPB;  // a full tile starts 16-byte aligned and holds 1296 float4
    if (full) {
        const float4* x4 = reinterpret_cast<const float4*>(xin);
        for (int q = tid; q < NF / 4; q += 256) reinterpret_cast<float4*>(s_x)[q] = x4[q];
    } else {
        for (int e = tid; e < nf; e += 256) s_x[e] = xin[e];
    }
    __syncthreads();
    // four lanes per plane: mean, then biased variance around it (what BatchNorm training uses)
    {
        const int pl = tid >> 2, sub = tid & 3;
        float s = 0.f;
        if (pl < np)
            for (int i = sub; i < PL; i += 4) s += s_x[pl * PL + i];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        const float mean = s * (1.0f / PL);
        float v = 0.f;
        if (pl < np)
            for (int i = sub; i < PL; i += 4) {
                float d = s_x[pl * PL + i] - mean;
                v += d * d;
            }
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if (sub == 0 && pl < np) {
            const float rstd = 1.0f / sqrtf(v * (1.0f / PL) + eps);
            const int c = (int)((p0 + pl) % C);
            const float g = gamma[c] * rstd;
            s_scale[pl] = g;
            s_shift[pl] = beta[c] - mean * g;
        }
    }
    __syncthreads();
    float* o = out + p0 * PL;
    const float* r = HAS_RES ? res + p0 * PL : nullptr;
    if (full) {
        for (int q = tid; q < NF / 4; q += 256) {
            const int e = q * 4;
            float4 xv = reinterpret_cast<const float4*>(s_x)[q];
            float in[4] = {xv.x, xv.y, xv.z, xv.w};
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (HAS_RES) {
                float4 t = reinterpret_cast<const float4*>(r)[q];
                rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w;
            }
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int pl = (e + j) / PL;
                float t = in[j] * s_scale[pl] + s_shift[pl];
                if (HAS_RES) t += rv[j];
                y[j] = RELU ? fmaxf(t, 0.f) : t;
            }
            reinterpret_cast<float4*>(o)[q] = make_float4(y[0], y[1], y[2], y[3]);
        }
    } else {
        for (int e = tid; e < nf; e += 256) {
            const int pl = e / PL;
            float t = s_x[e] * s_scale[pl] + s_shift[pl];
            if (HAS_RES) t += r[e];
            o[e] = RELU ? fmaxf(t, 0.f) : t;
        }
    }
}


// Channels-last variant: x is the NHWC memory of a logical [B, C, 9, 9] tensor, i.e.
// [B][81][C].  MIOpen's NHWC fp32 convolutions are ~20 % faster than the NCHW ones here, so the
// evaluator keeps activations channels-last end to end.  A workgroup takes S = max(1, 5184/(81 C))
// consecutive samples (C = 64 -> one sample = 20.7 KB tile), wave w sums positions w, w+4, ...
// of every (sample, channel) pair (lane = pair: conflict-free LDS reads), partial sums meet in LDS.
template <bool HAS_RES, bool RELU>
__global__ __launch_bounds__(256) void k_instnorm_act_nhwc(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res,
                                                           float* __restrict__ out, long long n_samples, int C, int S, float eps) {
    __shared__ float s_x[NF];
    __shared__ float s_red[4][64];
    __shared__ float s_scale[64], s_shift[64];
    const int tid = (int)threadIdx.x, sub = tid >> 6, pr = tid & 63;
    const long long s0 = (long long)blockIdx.x * S;
    const int ns = (int)((n_samples - s0) < S ? (n_samples - s0) : S);
    const int per = PL * C;        // floats per sample
    const int nf = ns * per;
    const long long base = s0 * per;
    const bool vec = ((nf & 3) == 0) && ((base & 3) == 0);
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        for (int q = tid; q < nf / 4; q += 256) reinterpret_cast<float4*>(s_x)[q] = x4[q];
    } else {
        for (int e = tid; e < nf; e += 256) s_x[e] = x[base + e];
    }
    __syncthreads();
    const int pairs = ns * C;
    const int sm = pr / C, c = pr - sm * C;
    float part = 0.f;
    if (pr < pairs)
        for (int i = sub; i < PL; i += 4) part += s_x[sm * per + i * C + c];
    s_red[sub][pr] = part;
    __syncthreads();
    const float mean = (s_red[0][pr] + s_red[1][pr] + s_red[2][pr] + s_red[3][pr]) * (1.0f / PL);
    __syncthreads();
    part = 0.f;
    if (pr < pairs)
        for (int i = sub; i < PL; i += 4) {
            float d = s_x[sm * per + i * C + c] - mean;
            part += d * d;
        }
    s_red[sub][pr] = part;
    __syncthreads();
    if (sub == 0 && pr < pairs) {
        const float var = (s_red[0][pr] + s_red[1][pr] + s_red[2][pr] + s_red[3][pr]) * (1.0f / PL);
        const float g = gamma[c] / sqrtf(var + eps);
        s_scale[pr] = g;
        s_shift[pr] = beta[c] - mean * g;
    }
    __syncthreads();
    float* o = out + base;
    const float* r = HAS_RES ? res + base : nullptr;
    if (vec) {
        for (int q = tid; q < nf / 4; q += 256) {
            const int e = q * 4;
            float4 xv = reinterpret_cast<const float4*>(s_x)[q];
            float in[4] = {xv.x, xv.y, xv.z, xv.w};
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (HAS_RES) {
                float4 t = reinterpret_cast<const float4*>(r)[q];
                rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w;
            }
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ee = e + j;
                const int ss = ee / per;
                const int cc = (ee - ss * per) % C;
                const int p2 = ss * C + cc;
                float t = in[j] * s_scale[p2] + s_shift[p2];
                if (HAS_RES) t += rv[j];
                y[j] = RELU ? fmaxf(t, 0.f) : t;
            }
            reinterpret_cast<float4*>(o)[q] = make_float4(y[0], y[1], y[2], y[3]);
        }
    } else {
        for (int e = tid; e < nf; e += 256) {
            const int ss = e / per;
            const int cc = (e - ss * per) % C;
            const int p2 = ss * C + cc;
            float t = s_x[e] * s_scale[p2] + s_shift[p2];
            if (HAS_RES) t += r[e];
            o[e] = RELU ? fmaxf(t, 0.f) : t;
        }
    }
}

// C = 64 (every trunk layer: 11 of the 12 normalisations of a forward pass), one sample per
// workgroup, no LDS tile: float4 q = tid + 256 k of the 81 x 64 sample always holds channels
// 4 (tid & 15) .. +3, so a thread keeps its 5-6 float4 in registers, sums them, meets the other
// 15 threads of its channel quad through two lane shuffles (xor 16, 32) and one 1 KB LDS exchange
// between the four wavefronts; the centred second pass costs no memory traffic.
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 quad_reduce(float4 s, float (*red)[64], int wave, int lane, int cq) {
    s.x += __shfl_xor(s.x, 16); s.y += __shfl_xor(s.y, 16); s.z += __shfl_xor(s.z, 16); s.w += __shfl_xor(s.w, 16);
    s.x += __shfl_xor(s.x, 32); s.y += __shfl_xor(s.y, 32); s.z += __shfl_xor(s.z, 32); s.w += __shfl_xor(s.w, 32);
    if (lane < 16) reinterpret_cast<float4*>(red[wave])[lane] = s;
    __syncthreads();
    float4 t = reinterpret_cast<const float4*>(red[0])[cq];
    t = f4_add(t, reinterpret_cast<const float4*>(red[1])[cq]);
    t = f4_add(t, reinterpret_cast<const float4*>(red[2])[cq]);
    t = f4_add(t, reinterpret_cast<const float4*>(red[3])[cq]);
    return t;
}
template <bool HAS_RES, bool RELU>
__global__ __launch_bounds__(256) void k_instnorm_act_nhwc64(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ res,
                                                             float* __restrict__ out, float eps) {
    __shared__ __attribute__((aligned(16))) float s_a[4][64];
    __shared__ __attribute__((aligned(16))) float s_b[4][64];
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63, cq = tid & 15;
    const size_t base = (size_t)blockIdx.x * (PL * 64);
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const bool tail = tid < 16;  // 1296 float4 per sample = 5 * 256 + 16
    float4 v[6];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = x4[tid + 256 * k];
    v[5] = tail ? x4[tid + 1280] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r[6];
    if (HAS_RES) {
        const float4* r4 = reinterpret_cast<const float4*>(res + base);
#pragma unroll
        for (int k = 0; k < 5; k++) r[k] = r4[tid + 256 * k];
        r[5] = tail ? r4[tid + 1280] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 s = v[0];
#pragma unroll
    for (int k = 1; k < 6; k++) s = f4_add(s, v[k]);
    s = quad_reduce(s, s_a, wave, lane, cq);
    const float4 mean = make_float4(s.x * (1.0f / PL), s.y * (1.0f / PL), s.z * (1.0f / PL), s.w * (1.0f / PL));
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < 5 || tail) {
            const float dx = v[k].x - mean.x, dy = v[k].y - mean.y, dz = v[k].z - mean.z, dw = v[k].w - mean.w;
            q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
        }
    }
    q = quad_reduce(q, s_b, wave, lane, cq);
    const float4 g4 = reinterpret_cast<const float4*>(gamma)[cq], b4 = reinterpret_cast<const float4*>(beta)[cq];
    float4 sc, sh;
    sc.x = g4.x / sqrtf(q.x * (1.0f / PL) + eps); sh.x = b4.x - mean.x * sc.x;
    sc.y = g4.y / sqrtf(q.y * (1.0f / PL) + eps); sh.y = b4.y - mean.y * sc.y;
    sc.z = g4.z / sqrtf(q.z * (1.0f / PL) + eps); sh.z = b4.z - mean.z * sc.z;
    sc.w = g4.w / sqrtf(q.w * (1.0f / PL) + eps); sh.w = b4.w - mean.w * sc.w;
    float4* o4 = reinterpret_cast<float4*>(out + base);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < 5 || tail) {
            float4 y;
            y.x = v[k].x * sc.x + sh.x; y.y = v[k].y * sc.y + sh.y; y.z = v[k].z * sc.z + sh.z; y.w = v[k].w * sc.w + sh.w;
            if (HAS_RES) { y.x += r[k].x; y.y += r[k].y; y.z += r[k].z; y.w += r[k].w; }
            if (RELU) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
            o4[tid + 256 * k] = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Input layer from the BOARD: conv1(state(board)) [+ per-leaf normalisation] + ReLU without
// reading the 26 x 9 x 9 planes.  state() (quoridor.py:58-131) is almost entirely structure:
//   planes 5..25  at most three of them are all-ones, the rest zero  -> a 3x3 convolution of a
//                 constant plane only depends on the border class of the output position
//                 (9 classes): table hot9[plane-5][class][c];
//   plane 0       ones on the 8x8 intersection grid except where a wall stands -> the all-empty
//                 image is one fixed table base0[pos][c]; every wall takes W[:,0] away again;
//   planes 1, 2   V / H walls, planes 3, 4 the two pawns: <= 22 non-zero pixels.
// So out[pos][c] = sum of <=3 hot9 rows + base0[pos][c] + sum over the non-zero pixels in the 3x3
// neighbourhood of pos of wd[kind][tap][c], kind = {H wall: W2-W0, V wall: W1-W0, mover pawn: W3,
// other pawn: W4}.  ~20 k multiply-free adds per leaf instead of 2.4 MFLOP, no 8.4 KB read; the
// tables are built from the layer's weights by the evaluator (policy_value_net.py refresh()).
// One leaf per workgroup, thread = (4 channels, positions p = (tid >> 4) + 16 k) exactly like
// k_instnorm_act_nhwc64, whose register-resident normalisation follows in the same kernel.
template <bool NORM>
__global__ __launch_bounds__(256) void k_input_layer(const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                     const uint64_t* __restrict__ meta, const uint8_t* __restrict__ terminal,
                                                     const float* __restrict__ hot9, const float* __restrict__ base0,
                                                     const float* __restrict__ wd, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ out, float eps) {
    __shared__ __attribute__((aligned(16))) float s_wd[4 * 9 * 64];
    __shared__ __attribute__((aligned(16))) float s_s9[9 * 64];
    __shared__ __attribute__((aligned(16))) float s_a[4][64];
    __shared__ __attribute__((aligned(16))) float s_b[4][64];
    __shared__ uint8_t s_code[96];
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63, cq = tid & 15;
    const size_t b = blockIdx.x;
    // packed board, include/qz_abi.h: meta = p1 i8 | p2 i8 | walls1 u8 | walls2 u8 | current player u8
    struct {
        uint64_t hb, vb;
        int p1, p2, w1, w2, cur;
    } bd;
    {
        const uint64_t m = meta[b];
        bd.hb = hb[b];
        bd.vb = vb[b];
        bd.p1 = (int)(int8_t)(m & 0xFF);
        bd.p2 = (int)(int8_t)((m >> 8) & 0xFF);
        bd.w1 = (int)((m >> 16) & 0xFF);
        bd.w2 = (int)((m >> 24) & 0xFF);
        bd.cur = (int)((m >> 32) & 0xFF);
    }
    const bool term = terminal ? (terminal[b] != 0) : false;
    // the base rows only depend on the thread: issue their loads before anything waits on the board
    const bool tail = tid < 16;
    float4 base[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int p = (tid >> 4) + 16 * k;
        base[k] = (k < 5 || tail) ? reinterpret_cast<const float4*>(base0)[p * 16 + cq] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int q = tid; q < 576; q += 256) reinterpret_cast<float4*>(s_wd)[q] = reinterpret_cast<const float4*>(wd)[q];
    {
        const int wm = bd.cur == 1 ? bd.w1 : bd.w2, wo = bd.cur == 1 ? bd.w2 : bd.w1;
        int im = wm - 1, io = wo - 1;  // Python index -1 -> last plane (quoridor.py:79-80)
        if (im < 0) im += 10;
        if (io < 0) io += 10;
        const int h0 = im, h1 = 10 + io;
        const bool h2 = bd.cur == 2;
        for (int e = tid; e < 576; e += 256) {
            const int cls = e >> 6, c = e & 63;
            float v = hot9[(h0 * 9 + cls) * 64 + c] + hot9[(h1 * 9 + cls) * 64 + c];
            if (h2) v += hot9[(20 * 9 + cls) * 64 + c];
            s_s9[e] = v;
        }
    }
    if (tid < 81) {
        const int r = tid / 9, c = tid - 9 * r;
        uint32_t code = 0u;
        if (r < 8 && c < 8) {
            const int ix = 8 * r + c;
            code = (uint32_t)((bd.hb >> ix) & 1ull) | ((uint32_t)((bd.vb >> ix) & 1ull) << 1);
        }
        int pm = bd.cur == 1 ? bd.p1 : bd.p2, po = bd.cur == 1 ? bd.p2 : bd.p1;
        if (pm < 0) pm += 81;
        if (po < 0) po += 81;
        code |= (tid == pm ? 4u : 0u) | (tid == po ? 8u : 0u);
        s_code[tid] = (uint8_t)code;
    }
    __syncthreads();
    float4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int p = (tid >> 4) + 16 * k;
        if ((k < 5 || tail) && !term) {
            const int y = p / 9, x = p - 9 * y;
            const int cls = (y == 0 ? 0 : (y == 8 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == 8 ? 2 : 1));
            acc = f4_add(reinterpret_cast<const float4*>(s_s9)[cls * 16 + cq], base[k]);
            for (int dy = -1; dy <= 1; dy++) {
                const int ny = y + dy;
                if (ny < 0 || ny > 8) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    const int nx = x + dx;
                    if (nx < 0 || nx > 8) continue;
                    const uint32_t code = s_code[ny * 9 + nx];
                    if (code == 0u) continue;
                    const int tap = (dy + 1) * 3 + (dx + 1);
#pragma unroll
                    for (int kind = 0; kind < 4; kind++)
                        if ((code >> kind) & 1u) acc = f4_add(acc, reinterpret_cast<const float4*>(s_wd)[(kind * 9 + tap) * 16 + cq]);
                }
            }
        }
        v[k] = acc;
    }
    const float4 b4 = reinterpret_cast<const float4*>(beta)[cq];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = b4;
    if (NORM) {
        float4 s = v[0];
#pragma unroll
        for (int k = 1; k < 6; k++) s = f4_add(s, v[k]);
        s = quad_reduce(s, s_a, wave, lane, cq);
        const float4 mean = make_float4(s.x * (1.0f / PL), s.y * (1.0f / PL), s.z * (1.0f / PL), s.w * (1.0f / PL));
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (k < 5 || tail) {
                const float dx = v[k].x - mean.x, dy = v[k].y - mean.y, dz = v[k].z - mean.z, dw = v[k].w - mean.w;
                q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
            }
        }
        q = quad_reduce(q, s_b, wave, lane, cq);
        const float4 g4 = reinterpret_cast<const float4*>(gamma)[cq];
        sc.x = g4.x / sqrtf(q.x * (1.0f / PL) + eps); sh.x = b4.x - mean.x * sc.x;
        sc.y = g4.y / sqrtf(q.y * (1.0f / PL) + eps); sh.y = b4.y - mean.y * sc.y;
        sc.z = g4.z / sqrtf(q.z * (1.0f / PL) + eps); sh.z = b4.z - mean.z * sc.z;
        sc.w = g4.w / sqrtf(q.w * (1.0f / PL) + eps); sh.w = b4.w - mean.w * sc.w;
    }
    float4* o4 = reinterpret_cast<float4*>(out + b * (size_t)(PL * 64));
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < 5 || tail) {
            float4 y;
            y.x = fmaxf(v[k].x * sc.x + sh.x, 0.f); y.y = fmaxf(v[k].y * sc.y + sh.y, 0.f);
            y.z = fmaxf(v[k].z * sc.z + sh.z, 0.f); y.w = fmaxf(v[k].w * sc.w + sh.w, 0.f);
            o4[tid + 256 * k] = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Both heads in one kernel: trunk output t [n][81][64] (NHWC) -> p [n][140], v [n]
//   h  = conv3x3(t) with the merged 64->6 head weights (value head channels 0..3, policy 4..5)
//        policy_value_net.py:64-65,69-70 (conv2 / conv3), bn2 / bn3 per leaf (or a folded bias), ReLU
//   v  = tanh(fc2(fc1(h[0:4] flattened c*81+pos)))                      policy_value_net.py:88-90
//   p  = exp(log_softmax(fc3(h[4:6] flattened)))                        policy_value_net.py:92-93,155
// Replaces a 94 us library convolution (2.3 GFLOP at 24 TFLOP/s), its zero-fill, a normalisation
// pass, three small GEMMs, softmax / exp / tanh and the reshape copies between them.
// Three leaves per workgroup: thread = (leaf, position) holds the six output channels, the 3x3x64
// weights are wave-uniform (scalar loads, v_pk_fma_f32 with the activation broadcast), the three
// staged [81][64] tiles use a 68-float row stride so the 16-byte LDS reads of consecutive
// positions do not collide.  fc1 runs on wavefronts 0-1 and fc3 on wavefronts 2-3 with
// pre-transposed weights (coalesced), each weight read once for the three leaves.
typedef float f2_t __attribute__((ext_vector_type(2)));
constexpr int HB = 6;     // leaves per workgroup: thread = (position, group of three leaves), 162 of 192 threads
constexpr int HT = 192;   // threads per workgroup; with 78 KB of LDS two workgroups share a CU, so one's
                          // latency-bound phases (staging, fully connected layers) run under the other's convolution
constexpr int XS = 36;    // row stride of a staged half tile (32 input channels + 4 pad), floats
constexpr int FS = 488;   // feature stride per leaf (486 used; 16-byte aligned rows)
template <int NL>
struct HeadFcT {          // lives in the tile memory once the convolutions are done
    float h1[NL][128];
    float logit[NL][144];
};
using HeadFc = HeadFcT<HB>;
constexpr int HBF = 3;    // leaves per workgroup of k_head_fc: half the multiply-adds per wavefront of the six-leaf form (the
                          // kernel is bound by its slowest wavefront's ~3,000 dependent FMAs + nine weight round trips, not by traffic)
// fc1 + fc2 + tanh and fc3 + softmax of HB leaves whose normalised head features (c * 81 + pos order,
// FS floats apart) are in LDS: the second half of k_head, and all of k_head_fc
template <int NL>
__device__ __forceinline__ void head_fc_tail(const float* __restrict__ s_f, HeadFcT<NL>& fc, int nb, long long s0, int tid, int wave, int lane,
                                             const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w2,
                                             const float* __restrict__ b2, const float* __restrict__ w3t, const float* __restrict__ b3,
                                             float* __restrict__ p_out, float* __restrict__ v_out) {
    if (wave < 2) {  // fc1: 324 -> 128, thread = output
        const int j = tid;
        float a1[NL];
#pragma unroll
        for (int s = 0; s < NL; s++) a1[s] = 0.f;
        // 36 coalesced weight loads in flight per trip (the loop is latency-bound, not issue-bound: every trip waits one
        // L2 round trip; 12 per trip made k_head_fc 25 us for 4,096 leaves)
        for (int i = 0; i < 4 * PL; i += 36) {
            float w[36];
#pragma unroll
            for (int k = 0; k < 36; k++) w[k] = w1t[(i + k) * 128 + j];
#pragma unroll
            for (int s = 0; s < NL; s++) {
#pragma unroll
                for (int k4 = 0; k4 < 9; k4++) {
                    const float4 f = *reinterpret_cast<const float4*>(&s_f[s * FS + i + 4 * k4]);
                    a1[s] = __builtin_fmaf(w[4 * k4 + 0], f.x, a1[s]);
                    a1[s] = __builtin_fmaf(w[4 * k4 + 1], f.y, a1[s]);
                    a1[s] = __builtin_fmaf(w[4 * k4 + 2], f.z, a1[s]);
                    a1[s] = __builtin_fmaf(w[4 * k4 + 3], f.w, a1[s]);
                }
            }
        }
        const float bb = b1[j];
#pragma unroll
        for (int s = 0; s < NL; s++) fc.h1[s][j] = a1[s] + bb;
    } else {  // fc3: 162 -> 140 on the third wavefront, thread = outputs j, j + 64, j + 128
        const int j = tid - 128, j1 = j + 64, j2 = j + 128;
        const bool three = j2 < 140;
        float a3[NL], a3b[NL], a3c[NL];
#pragma unroll
        for (int s = 0; s < NL; s++) a3[s] = a3b[s] = a3c[s] = 0.f;
        for (int i = 0; i < 2 * PL; i += 18) {  // 162 = 9 * 18: up to 54 loads in flight per trip
            float wa[18], wb[18], wc[18];
#pragma unroll
            for (int k = 0; k < 18; k++) {
                wa[k] = w3t[(i + k) * 140 + j];
                wb[k] = w3t[(i + k) * 140 + j1];
                wc[k] = three ? w3t[(i + k) * 140 + j2] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < NL; s++) {
#pragma unroll
                for (int k2 = 0; k2 < 9; k2++) {
                    const float2 f = *reinterpret_cast<const float2*>(&s_f[s * FS + 4 * PL + i + 2 * k2]);
                    a3[s] = __builtin_fmaf(wa[2 * k2], f.x, a3[s]);
                    a3[s] = __builtin_fmaf(wa[2 * k2 + 1], f.y, a3[s]);
                    a3b[s] = __builtin_fmaf(wb[2 * k2], f.x, a3b[s]);
                    a3b[s] = __builtin_fmaf(wb[2 * k2 + 1], f.y, a3b[s]);
                    a3c[s] = __builtin_fmaf(wc[2 * k2], f.x, a3c[s]);
                    a3c[s] = __builtin_fmaf(wc[2 * k2 + 1], f.y, a3c[s]);
                }
            }
        }
        const float ba = b3[j], bb = b3[j1], bc = three ? b3[j2] : 0.f;
#pragma unroll
        for (int s = 0; s < NL; s++) {
            fc.logit[s][j] = a3[s] + ba;
            fc.logit[s][j1] = a3b[s] + bb;
            if (three) fc.logit[s][j2] = a3c[s] + bc;
        }
    }
    __syncthreads();
    for (int s = wave; s < nb; s += HT / 64) {  // a wavefront finishes a leaf
        float part = fc.h1[s][lane] * w2[lane] + fc.h1[s][lane + 64] * w2[lane + 64];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) v_out[s0 + s] = tanhf(part + b2[0]);
        const float x0 = fc.logit[s][lane], x1 = fc.logit[s][lane + 64];
        const bool has2 = lane + 128 < 140;
        const float x2 = has2 ? fc.logit[s][lane + 128] : -INFINITY;
        float m = fmaxf(fmaxf(x0, x1), x2);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float e = expf(x0 - m) + expf(x1 - m) + (has2 ? expf(x2 - m) : 0.f);
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        const float lse = m + logf(e);
        float* po = p_out + (size_t)(s0 + s) * 140;
        po[lane] = expf(x0 - lse);
        po[lane + 64] = expf(x1 - lse);
        if (has2) po[lane + 128] = expf(x2 - lse);
    }
}

template <bool NORM>
__global__ __launch_bounds__(HT) void k_head(const float* __restrict__ t, long long n, const float* __restrict__ w6k,
                                              const float* __restrict__ gamma6, const float* __restrict__ beta6,
                                              const float* __restrict__ w1t, const float* __restrict__ b1,
                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                              const float* __restrict__ w3t, const float* __restrict__ b3,
                                              float* __restrict__ p_out, float* __restrict__ v_out, float eps) {
    __shared__ __attribute__((aligned(16))) float s_x[HB * PL * XS];   // 70 KB: six half tiles; later features + HeadFc
    __shared__ __attribute__((aligned(16))) float s_w[9 * 32 * 6];     // the 3x3 weights of the current half
    float* const s_f = s_x;                                            // normalised features, c*81+pos order (tiles are dead by then)
    __shared__ float s_part[HB * 6][4];
    __shared__ float s_stat[HB * 6][2];
    static_assert(sizeof(HeadFc) + sizeof(float) * HB * FS <= sizeof(float) * HB * PL * XS, "features + fc scratch must fit in the tile memory");
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long s0 = (long long)blockIdx.x * HB;
    const int nb = (int)((n - s0) < HB ? (n - s0) : HB);
    const int grp = tid / PL, pos = tid - PL * grp;   // conv role: leaves 3 grp .. 3 grp + 2 at one position
    const bool active = tid < (HB / 3) * PL;
    f2_t acc[3][3];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int k = 0; k < 3; k++) acc[s][k] = f2_t{0.f, 0.f};
    // ---- 3x3 convolution 64 -> 6, input channels in two halves of 32 (so nine leaves fit in LDS)
    for (int half = 0; half < 2; half++) {
        if (half) __syncthreads();
        // float4 q = (leaf, position, channel quad of the half); eight loads in flight per thread
        for (int q0 = tid; q0 < HB * PL * 8; q0 += HT * 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + HT * k;
                const int sl = q / (PL * 8), rem = q - sl * (PL * 8), p = rem >> 3, c4 = rem & 7;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < HB * PL * 8 && sl < nb)
                    v[k] = reinterpret_cast<const float4*>(t + (size_t)(s0 + sl) * (PL * 64) + p * 64 + half * 32)[c4];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + HT * k;
                const int sl = q / (PL * 8), rem = q - sl * (PL * 8), p = rem >> 3, c4 = rem & 7;
                if (q < HB * PL * 8) *reinterpret_cast<float4*>(&s_x[(sl * PL + p) * XS + c4 * 4]) = v[k];
            }
        }
        for (int q = tid; q < 9 * 32 * 6 / 4; q += HT) {  // [tap][32 channels of this half][6]
            const int tap = q / 48, r4 = q - tap * 48;
            reinterpret_cast<float4*>(s_w)[q] = reinterpret_cast<const float4*>(w6k + (tap * 64 + half * 32) * 6)[r4];
        }
        __syncthreads();
        if (active) {
            const int y = pos / 9, x = pos - 9 * y;
            for (int ky = 0; ky < 3; ky++) {
                const int ny = y + ky - 1;
                if (ny < 0 || ny > 8) continue;
                for (int kx = 0; kx < 3; kx++) {
                    const int nx = x + kx - 1;
                    if (nx < 0 || nx > 8) continue;
                    const float* xrow = &s_x[((3 * grp) * PL + ny * 9 + nx) * XS];
                    const float* wt = &s_w[(ky * 3 + kx) * (32 * 6)];
#pragma unroll 2
                    for (int c4 = 0; c4 < 8; c4++) {
                        float w[24];
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            const float4 wv = reinterpret_cast<const float4*>(wt + c4 * 24)[k];
                            w[4 * k] = wv.x; w[4 * k + 1] = wv.y; w[4 * k + 2] = wv.z; w[4 * k + 3] = wv.w;
                        }
#pragma unroll
                        for (int sl = 0; sl < 3; sl++) {
                            const float4 xv = *reinterpret_cast<const float4*>(xrow + sl * (PL * XS) + 4 * c4);
                            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                // plain v_fma_f32 on purpose: packed fp32 VALU ops (v_pk_fma_f32) were measured to
                                // return wrong values on MI355X while another wave's MFMAs run on the same SIMD
                                // (benchmarks/diag_head.py, diag_victims.py); the library is built with
                                // -fno-slp-vectorize for the same reason
#pragma unroll
                                for (int k = 0; k < 3; k++) {
                                    acc[sl][k].x = __builtin_fmaf(w[6 * j + 2 * k], xs[j], acc[sl][k].x);
                                    acc[sl][k].y = __builtin_fmaf(w[6 * j + 2 * k + 1], xs[j], acc[sl][k].y);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();  // every thread is done with the tiles: their memory becomes the feature buffer
    if (active) {
#pragma unroll
        for (int sl = 0; sl < 3; sl++) {
            float* feat = &s_f[(3 * grp + sl) * FS];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                feat[(2 * k) * PL + pos] = acc[sl][k].x;
                feat[(2 * k + 1) * PL + pos] = acc[sl][k].y;
            }
        }
    }
    __syncthreads();
    // ---- per (leaf, channel) statistics over the 81 positions: 4 partial sums per pair, two passes
    const int pr = tid >> 2, sub = tid & 3;
    const bool stat = pr < nb * 6;
    const float* hv = &s_f[(stat ? pr / 6 : 0) * FS + (stat ? pr % 6 : 0) * PL];
    float mean = 0.f;
    if (NORM) {
        float sum = 0.f;
        if (stat)
            for (int i = sub; i < PL; i += 4) sum += hv[i];
        if (stat) s_part[pr][sub] = sum;
        __syncthreads();
        if (stat) mean = (s_part[pr][0] + s_part[pr][1] + s_part[pr][2] + s_part[pr][3]) * (1.0f / PL);
        __syncthreads();
        float q = 0.f;
        if (stat)
            for (int i = sub; i < PL; i += 4) {
                const float d = hv[i] - mean;
                q += d * d;
            }
        if (stat) s_part[pr][sub] = q;
        __syncthreads();
    }
    if (stat && sub == 0) {
        const int c = pr % 6;
        float scale = 1.0f, shift = beta6[c];
        if (NORM) {
            const float var = (s_part[pr][0] + s_part[pr][1] + s_part[pr][2] + s_part[pr][3]) * (1.0f / PL);
            scale = gamma6[c] / sqrtf(var + eps);
            shift = beta6[c] - mean * scale;
        }
        s_stat[pr][0] = scale;
        s_stat[pr][1] = shift;
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int sl = 0; sl < 3; sl++) {
            const int s = 3 * grp + sl;
            if (s < nb) {
                float* feat = &s_f[s * FS];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    feat[(2 * k) * PL + pos] = fmaxf(acc[sl][k].x * s_stat[s * 6 + 2 * k][0] + s_stat[s * 6 + 2 * k][1], 0.f);
                    feat[(2 * k + 1) * PL + pos] = fmaxf(acc[sl][k].y * s_stat[s * 6 + 2 * k + 1][0] + s_stat[s * 6 + 2 * k + 1][1], 0.f);
                }
            }
        }
    }
    __syncthreads();
    head_fc_tail(s_f, *reinterpret_cast<HeadFc*>(s_x + HB * FS), nb, s0, tid, wave, lane, w1t, b1, w2, b2, w3t, b3, p_out, v_out);
}

// The fully connected half alone: feat [n][486] = the normalised, rectified head convolution
// (value channels 0..3, policy 4..5; c * 81 + pos) written by k_trunk's head stage (qz_conv.hip).
__global__ __launch_bounds__(HT) void k_head_fc(const float* __restrict__ feat, long long n, const float* __restrict__ w1t,
                                                const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                const float* __restrict__ w3t, const float* __restrict__ b3, float* __restrict__ p_out,
                                                float* __restrict__ v_out, const int* __restrict__ n_live) {
    __shared__ __attribute__((aligned(16))) float s_f[HBF * FS];
    __shared__ HeadFcT<HBF> fc;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long s0 = (long long)blockIdx.x * HBF;
    if (n_live) {  // only the first *n_live leaves exist (the engine's miss list)
        const long long nl = (long long)__builtin_amdgcn_readfirstlane(*n_live);
        n = nl < n ? nl : n;
        if (s0 >= n) return;
    }
    const int nb = (int)((n - s0) < HBF ? (n - s0) : HBF);
    for (int i = tid; i < HBF * 243; i += HT) {  // 486 floats per leaf as 243 float2
        const int sl = i / 243, k = i - sl * 243;
        float2 v = make_float2(0.f, 0.f);
        if (sl < nb) v = reinterpret_cast<const float2*>(feat + (size_t)(s0 + sl) * 486)[k];
        *reinterpret_cast<float2*>(&s_f[sl * FS + 2 * k]) = v;
    }
    __syncthreads();
    head_fc_tail(s_f, fc, nb, s0, tid, wave, lane, w1t, b1, w2, b2, w3t, b3, p_out, v_out);
}

}  // namespace

namespace qzl {
hipError_t instnorm_act(const float* x, const float* gamma, const float* beta, const float* res, float* out,
                        long long n_planes, int C, int relu, float eps, hipStream_t s) {
    if (n_planes <= 0) return hipSuccess;
    dim3 grid((unsigned)((n_planes + PPB - 1) / PPB)), block(256);
    if (res && relu) hipLaunchKernelGGL((k_instnorm_act<true, true>), grid, block, 0, s, x, gamma, beta, res, out, n_planes, C, eps);
    else if (res) hipLaunchKernelGGL((k_instnorm_act<true, false>), grid, block, 0, s, x, gamma, beta, res, out, n_planes, C, eps);
    else if (relu) hipLaunchKernelGGL((k_instnorm_act<false, true>), grid, block, 0, s, x, gamma, beta, res, out, n_planes, C, eps);
    else hipLaunchKernelGGL((k_instnorm_act<false, false>), grid, block, 0, s, x, gamma, beta, res, out, n_planes, C, eps);
    return hipGetLastError();
}
hipError_t instnorm_act_nhwc(const float* x, const float* gamma, const float* beta, const float* res, float* out,
                             long long n_samples, int C, int relu, float eps, hipStream_t s) {
    if (n_samples <= 0) return hipSuccess;
    if (C < 1 || C > 64) return hipErrorInvalidValue;
    const bool aligned16 = (((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)res) & 15) == 0;
    if (C == 64 && aligned16) {
        dim3 grid((unsigned)n_samples), block(256);
        if (res && relu) hipLaunchKernelGGL((k_instnorm_act_nhwc64<true, true>), grid, block, 0, s, x, gamma, beta, res, out, eps);
        else if (res) hipLaunchKernelGGL((k_instnorm_act_nhwc64<true, false>), grid, block, 0, s, x, gamma, beta, res, out, eps);
        else if (relu) hipLaunchKernelGGL((k_instnorm_act_nhwc64<false, true>), grid, block, 0, s, x, gamma, beta, res, out, eps);
        else hipLaunchKernelGGL((k_instnorm_act_nhwc64<false, false>), grid, block, 0, s, x, gamma, beta, res, out, eps);
        return hipGetLastError();
    }
    int S = NF / (PL * C);
    if (S < 1) S = 1;
    if (S * C > 64) S = 64 / C;
    if (S > 1 && (S & 1)) S -= 1;  // even sample count keeps every tile 16-byte aligned
    dim3 grid((unsigned)((n_samples + S - 1) / S)), block(256);
    if (res && relu) hipLaunchKernelGGL((k_instnorm_act_nhwc<true, true>), grid, block, 0, s, x, gamma, beta, res, out, n_samples, C, S, eps);
    else if (res) hipLaunchKernelGGL((k_instnorm_act_nhwc<true, false>), grid, block, 0, s, x, gamma, beta, res, out, n_samples, C, S, eps);
    else if (relu) hipLaunchKernelGGL((k_instnorm_act_nhwc<false, true>), grid, block, 0, s, x, gamma, beta, res, out, n_samples, C, S, eps);
    else hipLaunchKernelGGL((k_instnorm_act_nhwc<false, false>), grid, block, 0, s, x, gamma, beta, res, out, n_samples, C, S, eps);
    return hipGetLastError();
}
hipError_t input_layer(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, const uint8_t* terminal, long long n,
                       const float* hot9, const float* base0, const float* wd, const float* gamma, const float* beta, float* out,
                       float eps, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    dim3 grid((unsigned)n), block(256);
    if (gamma) hipLaunchKernelGGL((k_input_layer<true>), grid, block, 0, s, hb, vb, meta, terminal, hot9, base0, wd, gamma, beta, out, eps);
    else hipLaunchKernelGGL((k_input_layer<false>), grid, block, 0, s, hb, vb, meta, terminal, hot9, base0, wd, gamma, beta, out, eps);
    return hipGetLastError();
}
hipError_t head(const float* t, long long n, const float* w6k, const float* gamma6, const float* beta6, const float* w1t, const float* b1,
                const float* w2, const float* b2, const float* w3t, const float* b3, float* p_out, float* v_out, float eps, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + HB - 1) / HB)), block(HT);
    if (gamma6) hipLaunchKernelGGL((k_head<true>), grid, block, 0, s, t, n, w6k, gamma6, beta6, w1t, b1, w2, b2, w3t, b3, p_out, v_out, eps);
    else hipLaunchKernelGGL((k_head<false>), grid, block, 0, s, t, n, w6k, gamma6, beta6, w1t, b1, w2, b2, w3t, b3, p_out, v_out, eps);
    return hipGetLastError();
}
hipError_t head_fc(const float* feat, long long n, const float* w1t, const float* b1, const float* w2, const float* b2, const float* w3t,
                   const float* b3, float* p_out, float* v_out, hipStream_t s, const int* n_live) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_head_fc, dim3((unsigned)((n + HBF - 1) / HBF)), dim3(HT), 0, s, feat, n, w1t, b1, w2, b2, w3t, b3, p_out, v_out, n_live);
    return hipGetLastError();
}
}  // namespace qzl

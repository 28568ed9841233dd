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
}  // namespace qzl

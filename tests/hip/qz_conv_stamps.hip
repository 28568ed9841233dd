// qz_conv_stamps.hip -- TEST-ONLY diagnostic build of the fused trunk kernel with s_memtime stamps
// (cycles a wave spends staging / in the MFMA loop / in the statistics / in the hand-over), see
// benchmarks/trunk_stamps.py.  The product library is built WITHOUT the stamps.
#define QZ_TRUNK_STAMPS 1
#include "../../alphazero_quoridor_amd/csrc/qz_conv.hip"

extern "C" int qzt_trunk_stamps(float* x, long long n, int n_layers, const void* const* w16, const float* const* gamma, const float* const* beta,
                                const float* inv_scale /*[dev]*/, unsigned long long* stamps, void* stream) {
    TrunkArgs A;
    for (int l = 0; l < MAX_TRUNK_LAYERS; l++) {
        A.w16[l] = l < n_layers ? reinterpret_cast<const _Float16*>(w16[l]) : nullptr;
        A.gamma[l] = l < n_layers ? gamma[l] : nullptr;
        A.beta[l] = l < n_layers ? beta[l] : nullptr;
    }
    A.inv_scale = inv_scale;
    // QZ_STAMPS_PAD_LDS=<bytes>: extra dynamic LDS per workgroup, to measure a wave that has its SIMD for itself (96 KB -> one workgroup per CU)
    const size_t pad = getenv("QZ_STAMPS_PAD_LDS") ? (size_t)atol(getenv("QZ_STAMPS_PAD_LDS")) : 0;
    hipLaunchKernelGGL((k_trunk<false, true>), dim3((unsigned)n), dim3(128), pad, (hipStream_t)stream, x, x, A, n_layers, 1e-5f, HeadArgs{nullptr, nullptr, nullptr, nullptr, 31},
                       InputArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, nullptr, stamps);
    return (int)hipGetLastError();
}

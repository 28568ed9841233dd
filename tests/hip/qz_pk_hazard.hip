// qz_pk_hazard.hip -- TEST-ONLY (tests/test_gpu_determinism.py): the smallest kernel pair around the packed-fp32 hazard.
// Built WITHOUT the product's "-target-feature -packed-fp32-ops" (the assembler refuses v_pk_fma_f32 with it), into its
// own libqz_pk_hazard.so; nothing of the product links against it.
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------------------------------
// The packed-fp32 hazard, as small as it gets (profiles/round2/packed_fp32_next_to_mfma.txt).
// The kernels of the product that returned wrong values next to another wave's MFMAs all fed v_pk_fma_f32 from
// registers that an LDS read had just written; a victim iterating v_pk_fma_f32 on registers alone was never disturbed.
// This pair does exactly that and nothing else: per iteration one ds_read_b64 of a weight pair, then the FMA on it --
// as ONE v_pk_fma_f32 (PACKED) or as two v_fma_f32 -- both spelled in inline assembly, so the instruction mix does not
// depend on compiler flags.  Same arithmetic, same rounding: the two variants must agree bit for bit, alone and next
// to any neighbour.
typedef float pk2_t __attribute__((ext_vector_type(2)));
template <bool PACKED>
__global__ __launch_bounds__(256) void k_lds_fma_victim(float* out, int iters) {
    __shared__ pk2_t w[1024];
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < 1024; i += 256) w[i] = pk2_t{1.0f + 1e-4f * (float)(i % 37), 1.0f - 1e-4f * (float)(i % 41)};
    __syncthreads();
    pk2_t a = {0.001f * (float)((blockIdx.x * 256 + tid) % 977) + 0.5f, 0.25f};
    const pk2_t c = {0.0003f, -0.0002f};
    for (int it = 0; it < iters; it++) {
        const pk2_t b = w[(tid * 3 + it * 7) & 1023];  // ds_read_b64: the FMA's operand is an LDS result
        pk2_t d;
        if (PACKED) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        } else {
            float d0, d1;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d0) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
            d = pk2_t{d0, d1};
        }
        a = d * 0.5f + 0.25f;  // (keeps the recurrence bounded; identical in both variants)
    }
    out[(size_t)blockIdx.x * 256 + tid] = a[0] + a[1];
}
extern "C" int qzt_lds_fma_victim(float* out, int blocks, int iters, int packed, void* stream) {
    if (packed) hipLaunchKernelGGL((k_lds_fma_victim<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL((k_lds_fma_victim<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}

// MFMA lane-map probes (unit-test support): one 32x32 tile through the fragment maps the
// GEMM/attention kernels assume, so a wrong map shows up in isolation.
#include "common.h"

// A[32][16] f16 (k contiguous), Bt[32][16] f16 (row n, k contiguous) -> C[32][32] f32 = A @ Bt^T
__global__ void k_probe_mfma_f16(const f16 *__restrict__ A, const f16 *__restrict__ Bt, float *__restrict__ Cm)
{
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    f16x8 a = *(const f16x8 *)(A + r * 16 + 8 * h);
    f16x8 b = *(const f16x8 *)(Bt + r * 16 + 8 * h);
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        int row = (q & 3) + 8 * (q >> 2) + 4 * h;
        Cm[row * 32 + r] = c[q];
    }
}

// A[32][2] f32, Bt[32][2] f32 -> C = A @ Bt^T through v_mfma_f32_32x32x2_f32
__global__ void k_probe_mfma_f32(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ Cm)
{
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 + h], Bt[r * 2 + h], c, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        int row = (q & 3) + 8 * (q >> 2) + 4 * h;
        Cm[row * 32 + r] = c[q];
    }
}

extern "C" int32_t ctx_probe_mfma(int32_t which, const void *A, const void *Bt, float *Cm, ctx_stream_t stream)
{
    CTX_REQUIRE(A && Bt && Cm, "probe: null pointer");
    if (which == 0)
        hipLaunchKernelGGL(k_probe_mfma_f16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const f16 *)A, (const f16 *)Bt, Cm);
    else
        hipLaunchKernelGGL(k_probe_mfma_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float *)A, (const float *)Bt, Cm);
    CTX_CHECK_LAUNCH("probe");
    return CTX_OK;
}

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

// which = 2: ds_read_b64_tr_b16 as attention.hip uses it.  A = f16 tile [8 rows][32 cols]; lane l (q = (l&15)>>2, p = l&3,
// a = (l>>4)&1, h = l>>5) points at row 4h + q, columns 16a + 4p .. +3 and must receive column 16a + (l&15) of rows 4h .. 4h+3.
typedef short probe_s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k_probe_tr16(const f16 *__restrict__ A, float *__restrict__ Cm)
{
    __shared__ __attribute__((aligned(16))) f16 t[8 * 32];
    for (int i = threadIdx.x; i < 256; i += 64) t[i] = A[i];
    __syncthreads();
    const int l = threadIdx.x, q = (l & 15) >> 2, p = l & 3, a = (l >> 4) & 1, h = l >> 5;
    const f16 *ptr = t + (4 * h + q) * 32 + 16 * a + 4 * p;
    probe_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) probe_s16x4 *)ptr);
    f16x4 w = __builtin_bit_cast(f16x4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) Cm[l * 4 + j] = (float)w[j];
}

extern "C" int32_t ctx_probe_mfma(int32_t which, const void *A, const void *Bt, float *Cm, ctx_stream_t stream)
{
    CTX_REQUIRE(A && Bt && Cm, "probe: null pointer");
    if (which == 2)
        hipLaunchKernelGGL(k_probe_tr16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const f16 *)A, Cm);
    else if (which == 0)
        hipLaunchKernelGGL(k_probe_mfma_f16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const f16 *)A, (const f16 *)Bt, Cm);
    else
        hipLaunchKernelGGL(k_probe_mfma_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float *)A, (const float *)Bt, Cm);
    CTX_CHECK_LAUNCH("probe");
    return CTX_OK;
}

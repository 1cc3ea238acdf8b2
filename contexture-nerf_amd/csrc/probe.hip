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

// ------------------------------------------------------------------------------------------------
// Staging-rate probe (tools/probe_stage.py): how many bytes per second a CU moves out of L2 by LDS-DMA
// (global_load_lds_dwordx4), by plain 16-byte loads into registers, or by both at once.  One workgroup per CU, every wave
// issues U 1-KiB wave-instructions per turn and lets the newest U stay in flight.  Workgroups of one XCD walk a
// 2 MiB window (L2-resident, larger than L1) in 64 KiB regions.
typedef const __attribute__((address_space(1))) void *pr_gptr_t;
typedef __attribute__((address_space(3))) void *pr_lptr_t;
typedef unsigned int pr_u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ __launch_bounds__(1024) void k_probe_stage(const char *__restrict__ src, int iters, unsigned *__restrict__ sink, int shared_region)
{
    extern __shared__ __attribute__((aligned(16))) char plds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int region = shared_region ? 0 : ((blockIdx.x >> 3) & 31);
    const char *base = src + (size_t)region * 65536 + lane * 16;
    char *dst = plds + wave * (2 * U * 1024);
    pr_u32x4 acc = {0, 0, 0, 0};
    pr_u32x4 va[U], vb[U];
    constexpr int UD = MODE == 0 ? U : (MODE == 1 ? 0 : U / 2);     // DMA pieces per turn; the rest go to registers
    int off = wave * U * 1024;
    const int step = nw * U * 1024;
#pragma unroll
    for (int i = 0; i < U; ++i) vb[i] = (pr_u32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const char *p = base + ((off + i * 1024) & 65535);
            if (i < UD) __builtin_amdgcn_global_load_lds((pr_gptr_t)p, (pr_lptr_t)(dst + i * 1024), 16, 0, 0);
            else va[i] = *(const pr_u32x4 *)p;
        }
        off += step;
#pragma unroll
        for (int i = UD; i < U; ++i) acc ^= vb[i];
        if (UD == U) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const char *p = base + ((off + i * 1024) & 65535);
            if (i < UD) __builtin_amdgcn_global_load_lds((pr_gptr_t)p, (pr_lptr_t)(dst + (U + i) * 1024), 16, 0, 0);
            else vb[i] = *(const pr_u32x4 *)p;
        }
        off += step;
#pragma unroll
        for (int i = UD; i < U; ++i) acc ^= va[i];
        if (UD == U) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U) : "memory");
    }
#pragma unroll
    for (int i = UD; i < U; ++i) acc ^= vb[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[0] = acc[0];
}

// mode 0: LDS-DMA only, 1: register loads only, 2: half and half.  Returns milliseconds for `iters` turns (U KiB per wave and turn),
// negative on error.  src must hold >= 2 MiB.
extern "C" float ctx_probe_stage(int32_t mode, int32_t waves, int32_t u, int32_t iters, int32_t shared_region, const void *src, void *sink,
                                 ctx_stream_t stream)
{
    if (!src || !sink || waves < 1 || waves > 16 || (u != 4 && u != 8) || mode < 0 || mode > 2 || iters < 2) {
        ctx_set_error("probe_stage: bad argument");
        return -1.f;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)waves * 2 * u * 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), lds, s, (const char *)src, 2, (unsigned *)sink, shared_region);
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), lds, s, (const char *)src, iters, (unsigned *)sink, shared_region);
        (void)hipEventRecord(e1, s);
    };
    if (u == 4) { if (mode == 0) go(k_probe_stage<0, 4>); else if (mode == 1) go(k_probe_stage<1, 4>); else go(k_probe_stage<2, 4>); }
    else { if (mode == 0) go(k_probe_stage<0, 8>); else if (mode == 1) go(k_probe_stage<1, 8>); else go(k_probe_stage<2, 8>); }
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ctx_set_error("probe_stage: %s", hipGetErrorString(e)); return -2.f; }
    return ms;
}

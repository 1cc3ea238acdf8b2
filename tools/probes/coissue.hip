// Does one wave's MFMA stream overlap another wave's VALU / transcendental / LDS-read stream on the SAME SIMD of gfx950?
// Workgroup of 512 threads = 8 waves, waves w and w+4 share SIMD w%4 (one workgroup per CU, grid 256).  Wave role by half:
//   A-only: waves 0-3 run stream A, waves 4-7 idle;  B-only likewise;  A|B: waves 0-3 run A while waves 4-7 run B.
// Perfect overlap: t(A|B) = max(t(A), t(B)); none: t(A) + t(B).  Also the single-stream rates at 1 and 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 2048
enum { S_NONE = 0, S_MFMA, S_FMA, S_EXP, S_LDS, S_MFMA_DEP, S_MIX2, S_MIX4, S_MIX6, S_MIXE2, S_MIXE4, S_SLOT, S_SLOT_NOMFMA };
template <int S>
__device__ __forceinline__ float stream(float seed, const float *lds)
{
    float acc = 0.f;
    if (S == S_MFMA || S == S_MFMA_DEP) {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) c[i][q] = seed;
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)seed; b[j] = (_Float16)(seed + 1.f); }
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[S == S_MFMA_DEP ? 0 : i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[S == S_MFMA_DEP ? 0 : i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) acc += c[i][0];
    }
    if (S == S_SLOT || S == S_SLOT_NOMFMA) {
        // the attention inner slot: one MFMA, then exp2(fma(s, c, -m)) of two scores and their packed fp16 pair (dependent chain)
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) c[i][q] = seed;
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)seed; b[j] = (_Float16)(seed + 1.f); }
        float x[8], y[4];
        for (int i = 0; i < 8; ++i) x[i] = seed + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (S == S_SLOT) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
                float t0, t1;
                asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(t0) : "v"(x[2 * i]), "v"(seed));
                asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(t1) : "v"(x[2 * i + 1]), "v"(seed));
                asm volatile("v_exp_f32 %0, %0" : "+v"(t0));
                asm volatile("v_exp_f32 %0, %0" : "+v"(t1));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(y[i]) : "v"(t0), "v"(t1));
            }
        }
        for (int i = 0; i < 4; ++i) acc += c[i][0] + y[i];
    } else if (S >= S_MIX2) {
        // same wave: each MFMA followed by NV independent VALU (fma) or transcendental (exp) instructions
        constexpr int NV = S == S_MIX2 || S == S_MIXE2 ? 2 : (S == S_MIX4 || S == S_MIXE4 ? 4 : 6);
        constexpr bool EXP = S == S_MIXE2 || S == S_MIXE4;
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) c[i][q] = seed;
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)seed; b[j] = (_Float16)(seed + 1.f); }
        float x[24];
        for (int i = 0; i < 24; ++i) x[i] = seed + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i * NV + v]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i * NV + v]) : "v"(seed));
                }
            }
        }
        for (int i = 0; i < 4; ++i) acc += c[i][0];
        for (int i = 0; i < 24; ++i) acc += x[i];
    }
    if (S == S_FMA || S == S_EXP) {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = seed + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (S == S_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(seed));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            }
        }
        for (int i = 0; i < 16; ++i) acc += x[i];
    }
    if (S == S_LDS) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 s = {0, 0, 0, 0};
        const f4 *p = (const f4 *)lds + (threadIdx.x & 63);
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { f4 v = p[64 * (i & 7)]; asm volatile("" : "+v"(v)); s += v; }
        }
        acc = s[0] + s[1] + s[2] + s[3];
    }
    return acc;
}
template <int A, int B, int C3 = S_NONE>
__global__ __launch_bounds__(768) void k(float *out, float seed)
{
    __shared__ float lds[64 * 4 * 8 + 64];
    for (int i = threadIdx.x; i < 64 * 4 * 8; i += 768) lds[i] = seed;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float r = wave < 4 ? stream<A>(seed, lds) : (wave < 8 ? stream<B>(seed, lds) : stream<C3>(seed, lds));
    out[blockIdx.x * 768 + threadIdx.x] = r;
}
template <int A, int B, int C3 = S_NONE> static double run(float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B, C3>), dim3(256), dim3(768), 0, 0, out, 0.5f);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<A, B, C3>), dim3(256), dim3(768), 0, 0, out, 0.5f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / 5;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 768 * 4);
    const char *nm[] = {"none", "mfma x4 indep", "v_fma_f32 x16", "v_exp_f32 x16", "ds_read_b128 x16", "mfma dependent"};
    double t1[6];
    t1[1] = run<S_MFMA, S_NONE>(out); t1[2] = run<S_FMA, S_NONE>(out); t1[3] = run<S_EXP, S_NONE>(out); t1[4] = run<S_LDS, S_NONE>(out);
    t1[5] = run<S_MFMA_DEP, S_NONE>(out);
    for (int i = 1; i < 6; ++i) printf("one wave per SIMD  %-18s %8.1f us   (%.2f ns per instruction)\n", nm[i], t1[i], t1[i] * 1e3 / ITER / (i == 1 || i == 5 ? 4 : 16));
    printf("two waves per SIMD, same stream:   mfma %8.1f  fma %8.1f  exp %8.1f  lds %8.1f us\n", run<S_MFMA, S_MFMA>(out), run<S_FMA, S_FMA>(out),
           run<S_EXP, S_EXP>(out), run<S_LDS, S_LDS>(out));
    printf("mfma | fma    %8.1f us   (max %8.1f, sum %8.1f)\n", run<S_MFMA, S_FMA>(out), t1[1] > t1[2] ? t1[1] : t1[2], t1[1] + t1[2]);
    printf("mfma | exp    %8.1f us   (max %8.1f, sum %8.1f)\n", run<S_MFMA, S_EXP>(out), t1[1] > t1[3] ? t1[1] : t1[3], t1[1] + t1[3]);
    printf("mfma | lds    %8.1f us   (max %8.1f, sum %8.1f)\n", run<S_MFMA, S_LDS>(out), t1[1] > t1[4] ? t1[1] : t1[4], t1[1] + t1[4]);
    printf("fma  | exp    %8.1f us   (max %8.1f, sum %8.1f)\n", run<S_FMA, S_EXP>(out), t1[2] > t1[3] ? t1[2] : t1[3], t1[2] + t1[3]);
    printf("fma  | lds    %8.1f us   (max %8.1f, sum %8.1f)\n", run<S_FMA, S_LDS>(out), t1[2] > t1[4] ? t1[2] : t1[4], t1[2] + t1[4]);
    printf("mfma dep | exp %7.1f us   (max %8.1f, sum %8.1f)\n", run<S_MFMA_DEP, S_EXP>(out), t1[5] > t1[3] ? t1[5] : t1[3], t1[5] + t1[3]);
    printf("same wave, one per SIMD: mfma + 2 fma %8.1f   + 4 fma %8.1f   + 6 fma %8.1f   + 2 exp %8.1f   + 4 exp %8.1f us  (mfma alone %8.1f)\n",
           run<S_MIX2, S_NONE>(out), run<S_MIX4, S_NONE>(out), run<S_MIX6, S_NONE>(out), run<S_MIXE2, S_NONE>(out), run<S_MIXE4, S_NONE>(out), t1[1]);
    printf("same wave, two per SIMD: mfma + 2 fma %8.1f   + 4 fma %8.1f   + 6 fma %8.1f   + 2 exp %8.1f   + 4 exp %8.1f us  (mfma alone %8.1f)\n",
           run<S_MIX2, S_MIX2>(out), run<S_MIX4, S_MIX4>(out), run<S_MIX6, S_MIX6>(out), run<S_MIXE2, S_MIXE2>(out), run<S_MIXE4, S_MIXE4>(out), run<S_MFMA, S_MFMA>(out));
    printf("attention slot (mfma, 2 fma, 2 exp, cvt_pk; 4 slots per trip): 1 wave/SIMD %8.1f   2 waves %8.1f   3 waves %8.1f us;  without the mfma: 1 wave %8.1f  2 waves %8.1f  3 waves %8.1f;  mfma alone x3 waves %8.1f\n",
           run<S_SLOT, S_NONE>(out), run<S_SLOT, S_SLOT>(out), run<S_SLOT, S_SLOT, S_SLOT>(out), run<S_SLOT_NOMFMA, S_NONE>(out), run<S_SLOT_NOMFMA, S_SLOT_NOMFMA>(out),
           run<S_SLOT_NOMFMA, S_SLOT_NOMFMA, S_SLOT_NOMFMA>(out), run<S_MFMA, S_MFMA, S_MFMA>(out));
    printf("phase-separated waves: slot-without-mfma | mfma | mfma+slot: %8.1f us\n", run<S_SLOT_NOMFMA, S_MFMA, S_SLOT>(out));
    return 0;
}

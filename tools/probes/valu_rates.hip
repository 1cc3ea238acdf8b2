// VALU issue-rate probe for gfx950: wave-instructions per clock of the ops the attention softmax is made of.
// One wave per SIMD (grid = 256 CUs x 4 waves), 16 independent chains per op, ITER trips; time by hipEvents, clocks from
// the fma line (v_fma_f32 is full rate: 4 clk per wave64 instruction).  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define NCH 16
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    float x[NCH];
    f2 p[NCH];
    for (int i = 0; i < NCH; ++i) { x[i] = a + threadIdx.x * 1e-6f + i; p[i] = (f2){x[i], x[i] + 1.f}; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"((f2){a, a}), "v"((f2){b, b}));
            if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"((f2){a, a}));
            if (OP == 6) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1));
            if (OP == 7) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));
            if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            if (OP == 9) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
            if (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"((f2){a, a}));
            if (OP == 11) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
            if (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) s += x[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> static double run(float *out, const char *name, double ref_ns)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, 0.999f, 1e-3f);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, 0.999f, 1e-3f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ns = ms * 1e6 / 5 / ((double)ITER * NCH);
    printf("%-18s %7.3f ns per wave instruction  (%.2f x v_fma_f32 = %.1f clk at 4 clk per fma)\n", name, ns, ref_ns > 0 ? ns / ref_ns : 1.0, ref_ns > 0 ? 4.0 * ns / ref_ns : 4.0);
    return ns;
}
int main()
{
    float *out; hipMalloc(&out, 256 * 256 * 4);
    double r = run<0>(out, "v_fma_f32", 0);
    run<1>(out, "v_exp_f32", r); run<2>(out, "v_pk_fma_f32", r); run<3>(out, "v_max3_f32", r); run<4>(out, "v_cvt_pk_f16_f32", r);
    run<5>(out, "v_pk_mul_f32", r); run<6>(out, "v_ldexp_f32", r); run<7>(out, "v_floor_f32", r); run<8>(out, "v_rcp_f32", r);
    run<9>(out, "v_exp_f16", r); run<10>(out, "v_pk_add_f32", r); run<11>(out, "v_cvt_i32_f32", r); run<12>(out, "v_lshl_add_u32", r);
    return 0;
}

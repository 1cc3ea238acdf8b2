// LDS read throughput per CU on gfx950 by instruction: ds_read_b128, ds_read_b64, ds_read_b64_tr_b16 (conflict-free addresses),
// 4 / 8 / 12 waves per CU, 16 independent reads in flight per wave.  Build: hipcc -O3 --offload-arch=gfx950 lds_rates.hip -o lds_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 2048
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(768) void k(float *out, int stride)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // byte address of this lane's datum inside a 1 KB (b128) / 512 B (b64) row block; `stride` bytes between the 16 reads
    const unsigned base = (OP == 0 ? lane * 16 : lane * 8);
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < ITER; ++it) {
        f4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned ad = base + (unsigned)(i * stride);
            if (OP == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(ad));
            if (OP == 1) { f2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(ad)); v[i] = (f4){t[0], t[1], 0, 0}; }
            if (OP == 2) { f2 t; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t) : "v"(ad)); v[i] = (f4){t[0], t[1], 0, 0}; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
        acc += v[0] + v[15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int OP> static void run(float *out, const char *name, int bytes_per_lane)
{
    for (int threads = 256; threads <= 768; threads += 256) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 65536, 0, out, 1024);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 65536, 0, out, 1024);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 5, n = (double)ITER * 16 * (threads / 64);
        printf("%-20s %2d waves per CU: %8.1f us  %6.2f ns per wave instruction per CU  %6.1f B/ns per CU\n", name, threads / 64, us, us * 1e3 / n, n * 64 * bytes_per_lane / (us * 1e3));
    }
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 768 * 4);
    (void)hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>(out, "ds_read_b128", 16); run<1>(out, "ds_read_b64", 8); run<2>(out, "ds_read_b64_tr_b16", 8);
    return 0;
}

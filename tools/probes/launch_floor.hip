// Per-kernel floor on an in-order stream: back-to-back launches of (a) an empty kernel, (b) a kernel that writes 12 MB, (c) a kernel
// that reads 12 MB, 1000 launches each, wall time per launch by hipEvents.  Build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_write(f4 *y, unsigned n) { unsigned i = blockIdx.x * 256u + threadIdx.x; if (i < n) y[i] = (f4){1.f, 2.f, 3.f, (float)i}; }
__global__ __launch_bounds__(256) void k_read(const f4 *x, float *out, unsigned n)
{
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    f4 v = i < n ? x[i] : (f4){0, 0, 0, 0};
    if (v[0] == 123456.f) out[0] = v[1];
}
template <class F> static double per_launch_us(F f, int n = 1000)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) f();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / n;
}
int main()
{
    const unsigned n = 12u * 1024 * 1024 / 16;
    f4 *buf; float *out; (void)hipMalloc(&buf, (size_t)n * 16); (void)hipMalloc(&out, 4); (void)hipMemset(buf, 0, (size_t)n * 16);
    printf("empty kernel, 1 block of 64:            %6.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }));
    printf("empty kernel, 1024 blocks of 256:       %6.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0); }));
    printf("write 12 MB (3072 blocks):              %6.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_write, dim3((n + 255) / 256), dim3(256), 0, 0, buf, n); }));
    printf("read 12 MB (3072 blocks):               %6.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_read, dim3((n + 255) / 256), dim3(256), 0, 0, buf, out, n); }));
    printf("write 12 MB then read it (pair):        %6.2f us per pair\n", per_launch_us([&] { hipLaunchKernelGGL(k_write, dim3((n + 255) / 256), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(k_read, dim3((n + 255) / 256), dim3(256), 0, 0, buf, out, n); }));
    return 0;
}

// What paces a GroupNorm-apply-shaped streaming kernel on gfx950?  y = f(x) over [B][HW][C] fp16 with 16-byte accesses, variants:
//   0 copy, flat grid, 1 chunk per thread (torch-like)        1 copy, AU chunks per thread in flight, contiguous run per block
//   5 = 1 + the table prologue only   6 = modulo + math, table values from registers   7 = LDS table reads, index without modulo
//   2 = 1 + per-channel scale/shift from an LDS table (i % c8n)   3 = 2 + SiLU        4 = 2 but table index without the modulo (c8n | 256)
// Build: hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VAR, int AU>
__global__ __launch_bounds__(256) void k(const f16 *__restrict__ x, f16 *__restrict__ y, unsigned total, int c8n, const float *__restrict__ tab)
{
    extern __shared__ float s_ab[];
    const int C = c8n * 8;
    if (VAR >= 2 && VAR != 6) {
        for (int c = threadIdx.x; c < 2 * C; c += 256) s_ab[c] = tab[c];
        __syncthreads();
    }
    if (VAR == 0) {
        const unsigned i = blockIdx.x * 256u + threadIdx.x;
        if (i < total) *(f16x8 *)(y + (size_t)i * 8) = *(const f16x8 *)(x + (size_t)i * 8);
        return;
    }
    const unsigned per = 256u * AU;
    const unsigned beg = blockIdx.x * per;
    f16x8 v[AU];
#pragma unroll
    for (int u = 0; u < AU; ++u) v[u] = *(const f16x8 *)(x + (size_t)min(beg + 256u * u + threadIdx.x, total - 1u) * 8);
#pragma unroll
    for (int u = 0; u < AU; ++u) {
        const unsigned i = beg + 256u * u + threadIdx.x;
        if (i >= total) continue;
        f16x8 o = v[u];
        if (VAR >= 2 && VAR != 5) {
            const int c0 = (VAR == 7 ? (int)((threadIdx.x + 3 * u) & 31) : (int)(i % (unsigned)c8n)) * 8;
            f32x4 a0, a1, b0, b1;
            if (VAR == 6) { const float cf = (float)c0; a0 = (f32x4){cf, cf, cf, cf}; a1 = a0; b0 = a0; b1 = a0; }
            else { a0 = *(const f32x4 *)(s_ab + c0); a1 = *(const f32x4 *)(s_ab + c0 + 4); b0 = *(const f32x4 *)(s_ab + C + c0); b1 = *(const f32x4 *)(s_ab + C + c0 + 4); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (float)v[u][j] * (j < 4 ? a0[j] : a1[j - 4]) + (j < 4 ? b0[j] : b1[j - 4]);
                if (VAR == 3) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
                o[j] = (f16)t;
            }
        }
        *(f16x8 *)(y + (size_t)i * 8) = o;
    }
}
template <int VAR, int AU> static void run(const f16 *x, f16 *y, unsigned total, int c8n, const float *tab, const char *name)
{
    const unsigned nb = VAR == 0 ? (total + 255) / 256 : (total + 256 * AU - 1) / (256 * AU);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<VAR, AU>), dim3(nb), dim3(256), 2 * c8n * 8 * 4, 0, x, y, total, c8n, tab);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<VAR, AU>), dim3(nb), dim3(256), 2 * c8n * 8 * 4, 0, x, y, total, c8n, tab);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %-58s %7.1f us  %5.2f TB/s\n", name, ms * 1e3 / 20, 2.0 * total * 16 / (ms * 1e-3 / 20) / 1e12);
}
int main()
{
    const int shapes[3][3] = {{2, 9216, 320}, {12, 9216, 320}, {2, 2304, 640}};
    for (auto &sh : shapes) {
        const int B = sh[0], HW = sh[1], C = sh[2], c8n = C / 8;
        const unsigned total = (unsigned)B * HW * c8n;
        f16 *x, *y; float *tab;
        (void)hipMalloc(&x, (size_t)total * 16); (void)hipMalloc(&y, (size_t)total * 16); (void)hipMalloc(&tab, 2 * C * 4);
        (void)hipMemset(x, 0, (size_t)total * 16); (void)hipMemset(tab, 0, 2 * C * 4);
        printf("B %d HW %d C %d (%.1f MB in, same out)\n", B, HW, C, total * 16 / 1e6);
        run<0, 1>(x, y, total, c8n, tab, "copy, one 16-byte chunk per thread");
        run<1, 4>(x, y, total, c8n, tab, "copy, 4 chunks per thread in flight");
        run<1, 12>(x, y, total, c8n, tab, "copy, 12 chunks per thread in flight");
        run<2, 4>(x, y, total, c8n, tab, "scale/shift from an LDS table (i % c8n), 4 chunks");
        run<2, 12>(x, y, total, c8n, tab, "scale/shift from an LDS table (i % c8n), 12 chunks");
        run<3, 12>(x, y, total, c8n, tab, "the same + SiLU, 12 chunks");
        run<3, 4>(x, y, total, c8n, tab, "the same + SiLU, 4 chunks");
        run<5, 4>(x, y, total, c8n, tab, "copy after the table prologue (global -> LDS, barrier), 4");
        run<6, 4>(x, y, total, c8n, tab, "modulo + math, table values from registers, 4");
        run<7, 4>(x, y, total, c8n, tab, "LDS table reads + math, index without the modulo, 4");
        (void)hipFree(x); (void)hipFree(y); (void)hipFree(tab);
    }
    return 0;
}

// UV back-projection scatter without global float atomics: the backward of kaolin's texture_mapping (= grid_sample, bilinear,
// align_corners=False, padding 'border'; reference call site src/models/render.py:135 under autograd, src/training/trainer.py:866)
// and the direct UV scatter of north_star's "torch-scatter UV back-projection".
//
// The atomics form (geometry.hip: k_texmap_bwd) is bound by the chip's float-atomic rate (~1.3 TB/s of added bytes: 749 us for
// 7 views @1200^2).  Here pixels are first BINNED by atlas tile (32 x 32 texels); a workgroup then owns one tile, accumulates
// the bilinear contributions of its pixel list in an LDS-resident tile and writes every texel once with a plain store.
//
//   plan (depends on uv / mask only: built once per raster and reused by every backward of the SDS loop)
//     k_sb_count   per-workgroup LDS histogram of (pixel, tile) entries -> tile counts
//     k_sb_scan    exclusive scan -> tile offsets, chunk list (a tile's list is cut into chunks of <= SB_CH entries)
//     k_sb_fill    pixel ids into the tile lists (one global reservation per (workgroup, tile))
//   scatter (per call)
//     k_sb_zero    zero the int64 accumulators of multi-chunk tiles only
//     k_sb_accum   one workgroup per chunk: LDS tile of int64 FIXED-POINT sums (2^-32 units), ds_add_u64
//     k_sb_finish  multi-chunk tiles: int64 accumulator -> float, added to grad_tex
// Sums are integers, so the result does not depend on the order in which pixels, lanes or chunks arrive: bit-reproducible run
// to run, and closer to the exact sum than a float accumulation.  A pixel whose 2 x 2 footprint straddles tiles is listed in
// each tile it touches (1 + ~2/32 entries per pixel on average) and each tile adds only the taps that fall inside it.
#include "common.h"
#include "kernels.h"

#define SB_TS 32                  // tile side (texels)
#define SB_CH 8192                // entries per chunk (measured at 7 x 1200^2 onto 1024^2: 4096 -> 74.9 us, 8192 -> 70, 16384 -> 78.1)
#define SB_MAXC 4
#define SB_FIX 4294967296.0f      // 2^32

struct SbHeader {                 // first 64 bytes of the plan
    int32_t ntx, ntiles, B, T;
    int64_t HW, cap;              // entries capacity
    int32_t nchunks, nmulti;
    int64_t total;
};
// plan layout (bytes): header 64 | counts i32[ntiles] | offsets i64[ntiles+1] | cursor i64[ntiles] | nch i32[ntiles] |
//                      chunk_tile i32[maxchunks] | chunk_beg i64[maxchunks] | chunk_n i32[maxchunks] | entries u32[cap]
struct SbPlan {
    SbHeader *h; int32_t *counts; int64_t *offsets; int64_t *cursor; int32_t *nch; int32_t *ctile; int64_t *cbeg; int32_t *cn; uint32_t *entries;
    int64_t maxchunks;
};
static inline size_t sb_align(size_t x) { return (x + 255) / 256 * 256; }
static SbPlan sb_map(void *plan, int ntiles, int64_t cap)
{
    SbPlan p; char *c = (char *)plan;
    p.maxchunks = ntiles + cap / SB_CH + 1;
    p.h = (SbHeader *)c; c += 256;
    p.counts = (int32_t *)c; c += sb_align((size_t)ntiles * 4);
    p.offsets = (int64_t *)c; c += sb_align((size_t)(ntiles + 1) * 8);
    p.cursor = (int64_t *)c; c += sb_align((size_t)ntiles * 8);
    p.nch = (int32_t *)c; c += sb_align((size_t)ntiles * 4);
    p.ctile = (int32_t *)c; c += sb_align((size_t)p.maxchunks * 4);
    p.cbeg = (int64_t *)c; c += sb_align((size_t)p.maxchunks * 8);
    p.cn = (int32_t *)c; c += sb_align((size_t)p.maxchunks * 4);
    p.entries = (uint32_t *)c;
    return p;
}

__device__ __forceinline__ float sb_src_index(float g, int size)
{
    float c = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
    return fminf((float)(size - 1), fmaxf(c, 0.0f));
}
// the (up to 4) distinct tiles touched by the pixel's valid taps; returns their count
__device__ __forceinline__ int sb_tiles(float u, float v, int T, int ntx, int (&tiles)[4])
{
    const float ix = sb_src_index(u * 2.0f - 1.0f, T), iy = sb_src_index((1.0f - v) * 2.0f - 1.0f, T);
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), x1 = x0 + 1, y1 = y0 + 1;
    const bool bx1 = x1 < T, by1 = y1 < T;                          // x0, y0 are in range after the border clamp
    const int tx0 = x0 / SB_TS, ty0 = y0 / SB_TS, tx1 = bx1 ? x1 / SB_TS : tx0, ty1 = by1 ? y1 / SB_TS : ty0;
    int n = 0;
    tiles[n++] = ty0 * ntx + tx0;
    if (tx1 != tx0) tiles[n++] = ty0 * ntx + tx1;
    if (ty1 != ty0) {
        tiles[n++] = ty1 * ntx + tx0;
        if (tx1 != tx0) tiles[n++] = ty1 * ntx + tx1;
    }
    return n;
}

#define SB_PIX_PER_WG 4096
__global__ __launch_bounds__(256) void k_sb_count(const float *__restrict__ uv, const int64_t *__restrict__ mask_idx, int64_t N, int T, int ntx,
                                                  int ntiles, int32_t *__restrict__ counts)
{
    extern __shared__ int32_t s_hist[];
    for (int i = threadIdx.x; i < ntiles; i += 256) s_hist[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * SB_PIX_PER_WG, p1 = min(N, p0 + SB_PIX_PER_WG);
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        if (mask_idx && mask_idx[p] < 0) continue;
        const float2 q = *(const float2 *)(uv + p * 2);
        int t[4];
        const int n = sb_tiles(q.x, q.y, T, ntx, t);
        for (int k = 0; k < n; ++k) atomicAdd(&s_hist[t[k]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += 256)
        if (s_hist[i]) atomicAdd(&counts[i], s_hist[i]);
}

// one workgroup: exclusive scan of the counts, cursors, chunk list
__global__ __launch_bounds__(1024) void k_sb_scan(SbHeader *h, const int32_t *__restrict__ counts, int64_t *__restrict__ offsets,
                                                  int64_t *__restrict__ cursor, int32_t *__restrict__ nch, int32_t *__restrict__ ctile,
                                                  int64_t *__restrict__ cbeg, int32_t *__restrict__ cn, int ntiles)
{
    __shared__ int64_t s_sum[1024], s_chk[1024];
    const int per = (ntiles + 1023) / 1024, t0 = threadIdx.x * per, t1 = min(ntiles, t0 + per);
    int64_t s = 0, c = 0;
    for (int t = t0; t < t1; ++t) { s += counts[t]; c += (counts[t] + SB_CH - 1) / SB_CH; }
    s_sum[threadIdx.x] = s; s_chk[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t a = 0, b = 0;
        for (int i = 0; i < 1024; ++i) { int64_t x = s_sum[i], y = s_chk[i]; s_sum[i] = a; s_chk[i] = b; a += x; b += y; }
        h->total = a; h->nchunks = (int32_t)b;
        offsets[ntiles] = a;
    }
    __syncthreads();
    int64_t off = s_sum[threadIdx.x], ck = s_chk[threadIdx.x];
    int multi = 0;
    for (int t = t0; t < t1; ++t) {
        const int cnt = counts[t], n = (cnt + SB_CH - 1) / SB_CH;
        offsets[t] = off; cursor[t] = off; nch[t] = n;
        for (int k = 0; k < n; ++k) { ctile[ck + k] = t; cbeg[ck + k] = off + (int64_t)k * SB_CH; cn[ck + k] = min(SB_CH, cnt - k * SB_CH); }
        if (n > 1) ++multi;
        off += cnt; ck += n;
    }
    if (multi) atomicAdd(&h->nmulti, multi);
}

__global__ __launch_bounds__(256) void k_sb_fill(const float *__restrict__ uv, const int64_t *__restrict__ mask_idx, int64_t N, int T, int ntx,
                                                 int ntiles, int64_t *__restrict__ cursor, uint32_t *__restrict__ entries)
{
    extern __shared__ int32_t s_mem[];                              // [ntiles] counts -> running positions, then [ntiles] bases (2 x i32)
    int32_t *s_hist = s_mem;
    int64_t *s_base = (int64_t *)(s_mem + ((ntiles + 1) & ~1));
    for (int i = threadIdx.x; i < ntiles; i += 256) s_hist[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * SB_PIX_PER_WG, p1 = min(N, p0 + SB_PIX_PER_WG);
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        if (mask_idx && mask_idx[p] < 0) continue;
        const float2 q = *(const float2 *)(uv + p * 2);
        int t[4];
        const int n = sb_tiles(q.x, q.y, T, ntx, t);
        for (int k = 0; k < n; ++k) atomicAdd(&s_hist[t[k]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += 256) {
        const int c = s_hist[i];
        s_base[i] = c ? (int64_t)atomicAdd((unsigned long long *)&cursor[i], (unsigned long long)c) : 0;
        s_hist[i] = 0;
    }
    __syncthreads();
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        if (mask_idx && mask_idx[p] < 0) continue;
        const float2 q = *(const float2 *)(uv + p * 2);
        int t[4];
        const int n = sb_tiles(q.x, q.y, T, ntx, t);
        for (int k = 0; k < n; ++k) entries[s_base[t[k]] + atomicAdd(&s_hist[t[k]], 1)] = (uint32_t)p;
    }
}

__global__ __launch_bounds__(256) void k_sb_zero(const int32_t *__restrict__ nch, int ntx, int T, int C, long long *__restrict__ acc)
{
    const int t = blockIdx.x;
    if (nch[t] <= 1) return;
    const int ty = t / ntx, tx = t - ty * ntx;
    for (int i = threadIdx.x; i < C * SB_TS * SB_TS; i += 256) {
        const int c = i / (SB_TS * SB_TS), r = i - c * SB_TS * SB_TS, y = ty * SB_TS + r / SB_TS, x = tx * SB_TS + r % SB_TS;
        if (y < T && x < T) acc[((size_t)c * T + y) * T + x] = 0;
    }
}

__global__ __launch_bounds__(256) void k_sb_accum(const SbHeader *__restrict__ h, const float *__restrict__ go, const float *__restrict__ uv, int C, int T,
                                                  int ntx, const int32_t *__restrict__ nch, const int32_t *__restrict__ ctile,
                                                  const int64_t *__restrict__ cbeg, const int32_t *__restrict__ cn,
                                                  const uint32_t *__restrict__ entries, long long *__restrict__ acc, float *__restrict__ gt)
{
    if ((int)blockIdx.x >= h->nchunks) return;
    __shared__ unsigned long long s_t[SB_MAXC * SB_TS * SB_TS];     // 32 KiB: [c][y][x] fixed-point sums
    const int t = ctile[blockIdx.x], n = cn[blockIdx.x];
    const int64_t beg = cbeg[blockIdx.x];
    const int ty = t / ntx, tx = t - ty * ntx, X0 = tx * SB_TS, Y0 = ty * SB_TS;
    for (int i = threadIdx.x; i < C * SB_TS * SB_TS; i += 256) s_t[i] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const size_t pix = entries[beg + i];
        const float2 q = *(const float2 *)(uv + pix * 2);
        const float ix = sb_src_index(q.x * 2.0f - 1.0f, T), iy = sb_src_index((1.0f - q.y) * 2.0f - 1.0f, T);
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), x1 = x0 + 1, y1 = y0 + 1;
        const float w[4] = {((float)x1 - ix) * ((float)y1 - iy), (ix - (float)x0) * ((float)y1 - iy),
                            ((float)x1 - ix) * (iy - (float)y0), (ix - (float)x0) * (iy - (float)y0)};
        const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
        float g[SB_MAXC];
        for (int c = 0; c < C; ++c) g[c] = go[pix * C + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lx = xs[k] - X0, ly = ys[k] - Y0;
            if (xs[k] >= T || ys[k] >= T || lx < 0 || lx >= SB_TS || ly < 0 || ly >= SB_TS) continue;
            for (int c = 0; c < C; ++c) {
                const long long v = __float2ll_rn(g[c] * w[k] * SB_FIX);
                if (v) atomicAdd(&s_t[(c * SB_TS + ly) * SB_TS + lx], (unsigned long long)v);
            }
        }
    }
    __syncthreads();
    const bool single = nch[t] <= 1;
    for (int i = threadIdx.x; i < C * SB_TS * SB_TS; i += 256) {
        const long long v = (long long)s_t[i];
        if (!v) continue;
        const int c = i / (SB_TS * SB_TS), r = i - c * SB_TS * SB_TS, y = Y0 + r / SB_TS, x = X0 + r % SB_TS;
        if (y >= T || x >= T) continue;
        const size_t o = ((size_t)c * T + y) * T + x;
        if (single) gt[o] += (float)((double)v * (1.0 / 4294967296.0));           // this workgroup is the texel's only writer
        else atomicAdd((unsigned long long *)&acc[o], (unsigned long long)v);
    }
}

__global__ __launch_bounds__(256) void k_sb_finish(const int32_t *__restrict__ nch, int ntx, int T, int C, const long long *__restrict__ acc,
                                                   float *__restrict__ gt)
{
    const int t = blockIdx.x;
    if (nch[t] <= 1) return;
    const int ty = t / ntx, tx = t - ty * ntx;
    for (int i = threadIdx.x; i < C * SB_TS * SB_TS; i += 256) {
        const int c = i / (SB_TS * SB_TS), r = i - c * SB_TS * SB_TS, y = ty * SB_TS + r / SB_TS, x = tx * SB_TS + r % SB_TS;
        if (y >= T || x >= T) continue;
        const size_t o = ((size_t)c * T + y) * T + x;
        const long long v = acc[o];
        if (v) gt[o] += (float)((double)v * (1.0 / 4294967296.0));
    }
}

static int64_t sb_cap(int64_t N) { return 4 * N; }                 // worst case: every pixel straddles a tile corner

extern "C" int64_t ctx_texmap_bwd_plan_bytes(int32_t B, int32_t HW, int32_t T)
{
    if (B < 1 || HW < 1 || T < 1) return -1;
    const int64_t N = (int64_t)B * HW;
    if (N >= (1ll << 32)) return -1;
    const int ntx = cdiv(T, SB_TS), ntiles = ntx * ntx;
    SbPlan p = sb_map(nullptr, ntiles, sb_cap(N));
    return (int64_t)((char *)p.entries - (char *)nullptr) + sb_cap(N) * 4 + 256;
}

extern "C" int32_t ctx_texmap_bwd_plan(const float *uv, const int64_t *mask_idx, int32_t B, int32_t HW, int32_t T, void *plan, ctx_stream_t stream)
{
    CTX_REQUIRE(uv && plan && B > 0 && HW > 0 && T > 0, "texmap_bwd_plan: bad args");
    const int64_t N = (int64_t)B * HW;
    CTX_REQUIRE(N < (1ll << 32), "texmap_bwd_plan: %lld pixels do not fit 32-bit pixel ids", (long long)N);
    const int ntx = cdiv(T, SB_TS), ntiles = ntx * ntx;
    CTX_REQUIRE((size_t)ntiles * 12 + 16 <= 60 * 1024, "texmap_bwd_plan: T=%d gives %d tiles, beyond the LDS histogram", T, ntiles);
    hipStream_t s = (hipStream_t)stream;
    SbPlan p = sb_map(plan, ntiles, sb_cap(N));
    (void)hipMemsetAsync(plan, 0, (size_t)((char *)p.offsets - (char *)plan), s);        // header + counts
    SbHeader hh = {}; hh.ntx = ntx; hh.ntiles = ntiles; hh.B = B; hh.T = T; hh.HW = HW; hh.cap = sb_cap(N);
    (void)hipMemcpyAsync(p.h, &hh, sizeof(hh), hipMemcpyHostToDevice, s);
    const unsigned nb = (unsigned)cdiv64(N, SB_PIX_PER_WG);
    hipLaunchKernelGGL(k_sb_count, dim3(nb), dim3(256), (size_t)ntiles * 4, s, uv, mask_idx, N, T, ntx, ntiles, p.counts);
    hipLaunchKernelGGL(k_sb_scan, dim3(1), dim3(1024), 0, s, p.h, p.counts, p.offsets, p.cursor, p.nch, p.ctile, p.cbeg, p.cn, ntiles);
    hipLaunchKernelGGL(k_sb_fill, dim3(nb), dim3(256), (size_t)((ntiles + 1) & ~1) * 4 + (size_t)ntiles * 8, s, uv, mask_idx, N, T, ntx, ntiles, p.cursor,
                       p.entries);
    CTX_CHECK_LAUNCH("texmap_bwd_plan");
    return CTX_OK;
}

extern "C" int64_t ctx_texture_mapping_bwd_binned_ws_bytes(int32_t C, int32_t T) { return (C < 1 || T < 1) ? -1 : (int64_t)C * T * T * 8 + 256; }

extern "C" int32_t ctx_texture_mapping_bwd_binned(const float *grad_out, const float *uv, int32_t B, int32_t HW, int32_t C, int32_t T, const void *plan,
                                                  void *ws, float *grad_tex, ctx_stream_t stream)
{
    CTX_REQUIRE(grad_out && uv && plan && ws && grad_tex && B > 0 && HW > 0 && T > 0, "texture_mapping_bwd_binned: bad args");
    CTX_REQUIRE(C >= 1 && C <= SB_MAXC, "texture_mapping_bwd_binned: C=%d outside [1, %d] (use ctx_texture_mapping_bwd)", C, SB_MAXC);
    const int64_t N = (int64_t)B * HW;
    const int ntx = cdiv(T, SB_TS), ntiles = ntx * ntx;
    SbPlan p = sb_map(const_cast<void *>(plan), ntiles, sb_cap(N));
    hipStream_t s = (hipStream_t)stream;
    long long *acc = (long long *)ws;
    hipLaunchKernelGGL(k_sb_zero, dim3(ntiles), dim3(256), 0, s, p.nch, ntx, T, C, acc);
    hipLaunchKernelGGL(k_sb_accum, dim3((unsigned)p.maxchunks), dim3(256), 0, s, p.h, grad_out, uv, C, T, ntx, p.nch, p.ctile, p.cbeg, p.cn, p.entries, acc,
                       grad_tex);
    hipLaunchKernelGGL(k_sb_finish, dim3(ntiles), dim3(256), 0, s, p.nch, ntx, T, C, acc, grad_tex);
    CTX_CHECK_LAUNCH("texture_mapping_bwd_binned");
    return CTX_OK;
}

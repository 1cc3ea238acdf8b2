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
//     k_sb_absmax  (float mode) max |grad_out| as a bit pattern -> the call's power-of-two scale
//     k_sb_zero    zero the int64 accumulators of multi-chunk tiles only; block 0 fixes the scale exponent
//     k_sb_accum   one workgroup per chunk: LDS tile of int64 FIXED-POINT sums, ds_add_u64
//     k_sb_finish  multi-chunk tiles: int64 accumulator -> float, added to grad_tex
// Sums are integers, so the result does not depend on the order in which pixels, lanes or chunks arrive: bit-reproducible run
// to run.  A pixel whose 2 x 2 footprint straddles tiles is listed in each tile it touches (1 + ~2/32 entries per pixel on
// average) and each tile adds only the taps that fall inside it.
//
// Fixed-point unit.  Float mode (ctx_texture_mapping_bwd_binned, the autograd backward): the unit is chosen PER CALL from
// max|grad_out| = m * 2^x (m in [0.5,1)): a tap g*w (rounded to float once, as the atomics kernel does) is scaled by 2^(E-x),
// E = 62 - ceil(log2(B*HW)), so the largest tap sits just under 2^E, N of them cannot overflow an int64, and the unit is
// 2^-E of the largest tap (2^-38 at 7 x 1200^2): gradients of any magnitude (1e-8 under a mean-reduced loss, 1e+4 under a
// summed one) keep the same RELATIVE resolution, which the scale-invariant Adam of the reference's SDS loop (eps 1e-15,
// src/training/trainer.py:603) relies on.  Non-finite gradients are not hidden: the whole output becomes NaN.
// Fixed mode (ctx_uv_scatter_fixed, the UV back-projection of painted views): the caller names the unit (2^-frac_bits) and gets
// the raw int64 sums, so that view shards on different ranks can be all-reduced as INTEGERS and divided once: the N-rank atlas
// is bit-identical to the 1-rank atlas.  Without a plan (atlases beyond the LDS histogram, T > 2272) the
// fixed mode falls back to one global int64 atomic per tap (k_sb_direct): slower, same sums.
#include <algorithm>
#include "common.h"
#include "kernels.h"

#define SB_TS 32                  // tile side (texels)
#define SB_CH 8192                // entries per chunk (measured at 7 x 1200^2 onto 1024^2: 4096 -> 74.9 us, 8192 -> 70, 16384 -> 78.1)
#define SB_MAXC 4

struct SbCtrl {                   // 256 bytes behind the accumulators of the float mode's workspace
    uint32_t absmax_bits;         // max over grad_out of (bits & 0x7fffffff): monotone in |x| for non-NaN, NaN patterns exceed +inf's
    int32_t e;                    // taps are scaled by 2^e
    int32_t nonfinite;
    uint32_t chk_now;             // sampled checksum of (uv, mask) as they are NOW (compared with the plan's)
};

struct SbHeader {                 // first 64 bytes of the plan
    int32_t ntx, ntiles, B, T;
    int64_t HW, cap;              // entries capacity
    int32_t nchunks, nmulti;
    int64_t total;
    uint32_t chk;                 // sampled checksum of (uv, mask) at plan time
    int32_t stale;                // sticky: a scatter found the raster changed under the plan
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

// ---- per-call control: |grad_out| maximum, raster checksum ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sb_absmax(const float *__restrict__ go, int64_t n, SbCtrl *__restrict__ ctrl)
{
    uint32_t m = 0;
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const uint4 v = *(const uint4 *)(go + i * 4);
        m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(go[n4 * 4 + threadIdx.x]) & 0x7fffffffu);
    for (int o = 32; o; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    // ONE atomic per workgroup: thousands of atomics on one address serialise at the memory side (~11 ns each)
    __shared__ uint32_t s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m) atomicMax(&ctrl->absmax_bits, m);
    }
}

#define SB_CHK_SAMPLES 65536
__device__ __forceinline__ uint32_t sb_mix(uint32_t h, uint32_t v) { h ^= v; h *= 0x9E3779B1u; return h ^ (h >> 15); }
// order-free (xor of per-sample hashes) checksum of <= 64k evenly spaced (uv, mask sign) samples; out must be zero before
__global__ __launch_bounds__(256) void k_sb_checksum(const float *__restrict__ uv, const int64_t *__restrict__ mask_idx, int64_t N, uint32_t *__restrict__ out)
{
    const int64_t step = max((int64_t)1, N / SB_CHK_SAMPLES);
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x, p = k * step;
    uint32_t h = 0;
    if (p < N) {
        const uint2 q = *(const uint2 *)(uv + p * 2);
        h = sb_mix(sb_mix(sb_mix(0x811C9DC5u, (uint32_t)k), q.x), q.y);
        h = sb_mix(h, (mask_idx && mask_idx[p] < 0) ? 1u : 2u);
    }
    for (int o = 32; o; o >>= 1) h ^= (uint32_t)__shfl_xor((int)h, o);
    __shared__ uint32_t s_h[4];
    if ((threadIdx.x & 63) == 0) s_h[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        h = s_h[0] ^ s_h[1] ^ s_h[2] ^ s_h[3];
        if (h) atomicXor(out, h);
    }
}

__global__ __launch_bounds__(256) void k_sb_zero(SbHeader *__restrict__ h, const int32_t *__restrict__ nch, int ntx, int T, int C,
                                                 long long *__restrict__ acc, SbCtrl *__restrict__ ctrl, int E)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (ctrl->chk_now != h->chk) h->stale = 1;
        const uint32_t b = ctrl->absmax_bits;
        int e = 0, bad = 0;
        if (b >= 0x7f800000u) bad = 1;                                  // inf or NaN somewhere in grad_out
        else if (b) { int x; (void)frexpf(__uint_as_float(b), &x); e = E - x; }
        ctrl->e = e; ctrl->nonfinite = bad;
    }
    const int t = blockIdx.x;
    if (nch[t] <= 1) return;
    const int ty = t / ntx, tx = t - ty * ntx;
    for (int i = threadIdx.x; i < C * SB_TS * SB_TS; i += 256) {
        const int c = i / (SB_TS * SB_TS), r = i - c * SB_TS * SB_TS, y = ty * SB_TS + r / SB_TS, x = tx * SB_TS + r % SB_TS;
        if (y < T && x < T) acc[((size_t)c * T + y) * T + x] = 0;
    }
}

// FIXED: the unit is 2^-frac (caller's) and every tile adds into the int64 accumulator `acc` (which the caller owns and may
// carry across calls / ranks); otherwise the unit comes from ctrl->e and single-chunk tiles write their floats directly.
template <bool FIXED>
__global__ __launch_bounds__(256) void k_sb_accum(const SbHeader *__restrict__ h, const float *__restrict__ go, const float *__restrict__ uv, int C, int T,
                                                  int ntx, const int32_t *__restrict__ nch, const int32_t *__restrict__ ctile,
                                                  const int64_t *__restrict__ cbeg, const int32_t *__restrict__ cn,
                                                  const uint32_t *__restrict__ entries, long long *__restrict__ acc, float *__restrict__ gt,
                                                  const SbCtrl *__restrict__ ctrl, int frac)
{
    if ((int)blockIdx.x >= h->nchunks) return;
    __shared__ unsigned long long s_t[SB_MAXC * SB_TS * SB_TS];     // 32 KiB: [c][y][x] fixed-point sums
    const int e = FIXED ? frac : ctrl->e;
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
                const long long v = __float2ll_rn(ldexpf(g[c] * w[k], e));      // one float rounding of the tap, exact scaling
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
        if (FIXED) {
            if (single) acc[o] += v;                                               // this workgroup is the texel's only writer
            else atomicAdd((unsigned long long *)&acc[o], (unsigned long long)v);
        } else {
            if (single) gt[o] += (float)ldexp((double)v, -e);
            else atomicAdd((unsigned long long *)&acc[o], (unsigned long long)v);
        }
    }
}

__global__ __launch_bounds__(256) void k_sb_finish(const SbHeader *__restrict__ h, const int32_t *__restrict__ nch, int ntx, int T, int C,
                                                   const long long *__restrict__ acc, float *__restrict__ gt, const SbCtrl *__restrict__ ctrl)
{
    const int t = blockIdx.x;
    const bool poison = ctrl->nonfinite || h->stale;                 // never hand back a plausible-looking wrong gradient
    if (nch[t] <= 1 && !poison) return;
    const int ty = t / ntx, tx = t - ty * ntx, e = ctrl->e;
    for (int i = threadIdx.x; i < C * SB_TS * SB_TS; i += 256) {
        const int c = i / (SB_TS * SB_TS), r = i - c * SB_TS * SB_TS, y = ty * SB_TS + r / SB_TS, x = tx * SB_TS + r % SB_TS;
        if (y >= T || x >= T) continue;
        const size_t o = ((size_t)c * T + y) * T + x;
        if (poison) { gt[o] = __uint_as_float(0x7fc00000u); continue; }
        const long long v = acc[o];
        if (v) gt[o] += (float)ldexp((double)v, -e);
    }
}

// fixed mode without a plan: one global int64 atomic per tap
__global__ __launch_bounds__(256) void k_sb_direct(const float *__restrict__ go, const float *__restrict__ uv, const int64_t *__restrict__ mask_idx, int64_t N,
                                                   int C, int T, int frac, long long *__restrict__ acc)
{
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= N || (mask_idx && mask_idx[pix] < 0)) return;
    const float2 q = *(const float2 *)(uv + pix * 2);
    const float ix = sb_src_index(q.x * 2.0f - 1.0f, T), iy = sb_src_index((1.0f - q.y) * 2.0f - 1.0f, T);
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), x1 = x0 + 1, y1 = y0 + 1;
    const float w[4] = {((float)x1 - ix) * ((float)y1 - iy), (ix - (float)x0) * ((float)y1 - iy),
                        ((float)x1 - ix) * (iy - (float)y0), (ix - (float)x0) * (iy - (float)y0)};
    const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    for (int k = 0; k < 4; ++k) {
        if (xs[k] >= T || ys[k] >= T) continue;
        for (int c = 0; c < C; ++c) {
            const long long v = __float2ll_rn(ldexpf(go[pix * C + c] * w[k], frac));
            if (v) atomicAdd((unsigned long long *)&acc[((size_t)c * T + ys[k]) * T + xs[k]], (unsigned long long)v);
        }
    }
}

// int64 sums in units of 2^-frac -> float (out = or +=)
__global__ __launch_bounds__(256) void k_fixed_to_float(const long long *__restrict__ acc, int64_t n, int frac, int accumulate, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = (float)ldexp((double)acc[i], -frac);
    out[i] = accumulate ? out[i] + v : v;
}

static int64_t sb_cap(int64_t N) { return 4 * N; }
static unsigned sb_chk_blocks(int64_t N) { const int64_t step = N / SB_CHK_SAMPLES > 1 ? N / SB_CHK_SAMPLES : 1; return (unsigned)cdiv64(cdiv64(N, step), 256); }
static int sb_ceil_log2(int64_t n) { int k = 0; while ((1ll << k) < n) ++k; return k; }                 // worst case: every pixel straddles a tile corner

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
    hipLaunchKernelGGL(k_sb_checksum, dim3(sb_chk_blocks(N)), dim3(256), 0, s, uv, mask_idx, N, &p.h->chk);
    CTX_CHECK_LAUNCH("texmap_bwd_plan");
    return CTX_OK;
}

extern "C" int64_t ctx_texture_mapping_bwd_binned_ws_bytes(int32_t C, int32_t T) { return (C < 1 || T < 1) ? -1 : (int64_t)C * T * T * 8 + 512; }

extern "C" int32_t ctx_texmap_plan_max_res(void)
{
    int ntx = 1;
    while ((size_t)(ntx + 1) * (ntx + 1) * 12 + 16 <= 60 * 1024) ++ntx;
    return ntx * SB_TS;
}

static SbCtrl *sb_ctrl(void *ws, int C, int T) { return (SbCtrl *)((char *)ws + (((size_t)C * T * T * 8 + 255) / 256) * 256); }

extern "C" int32_t ctx_texture_mapping_bwd_binned(const float *grad_out, const float *uv, const int64_t *mask_idx, int32_t B, int32_t HW, int32_t C, int32_t T,
                                                  const void *plan, void *ws, float *grad_tex, ctx_stream_t stream)
{
    CTX_REQUIRE(grad_out && uv && plan && ws && grad_tex && B > 0 && HW > 0 && T > 0, "texture_mapping_bwd_binned: bad args");
    CTX_REQUIRE(C >= 1 && C <= SB_MAXC, "texture_mapping_bwd_binned: C=%d outside [1, %d] (use ctx_texture_mapping_bwd)", C, SB_MAXC);
    const int64_t N = (int64_t)B * HW;
    const int ntx = cdiv(T, SB_TS), ntiles = ntx * ntx;
    SbPlan p = sb_map(const_cast<void *>(plan), ntiles, sb_cap(N));
    hipStream_t s = (hipStream_t)stream;
    long long *acc = (long long *)ws;
    SbCtrl *ctrl = sb_ctrl(ws, C, T);
    (void)hipMemsetAsync(ctrl, 0, 256, s);
    hipLaunchKernelGGL(k_sb_absmax, dim3((unsigned)std::min<int64_t>(1024, cdiv64(N * C, 1024))), dim3(256), 0, s, grad_out, N * C, ctrl);
    hipLaunchKernelGGL(k_sb_checksum, dim3(sb_chk_blocks(N)), dim3(256), 0, s, uv, mask_idx, N, &ctrl->chk_now);
    hipLaunchKernelGGL(k_sb_zero, dim3(ntiles), dim3(256), 0, s, p.h, p.nch, ntx, T, C, acc, ctrl, 62 - sb_ceil_log2(N));
    hipLaunchKernelGGL(k_sb_accum<false>, dim3((unsigned)p.maxchunks), dim3(256), 0, s, p.h, grad_out, uv, C, T, ntx, p.nch, p.ctile, p.cbeg, p.cn, p.entries,
                       acc, grad_tex, ctrl, 0);
    hipLaunchKernelGGL(k_sb_finish, dim3(ntiles), dim3(256), 0, s, p.h, p.nch, ntx, T, C, acc, grad_tex, ctrl);
    CTX_CHECK_LAUNCH("texture_mapping_bwd_binned");
    return CTX_OK;
}

extern "C" int32_t ctx_uv_scatter_fixed(const float *values, const float *uv, const int64_t *mask_idx, int32_t B, int32_t HW, int32_t C, int32_t T,
                                        const void *plan, int32_t frac_bits, int64_t *acc, ctx_stream_t stream)
{
    CTX_REQUIRE(values && uv && acc && B > 0 && HW > 0 && T > 0 && C >= 1, "uv_scatter_fixed: bad args");
    CTX_REQUIRE(frac_bits >= -64 && frac_bits <= 62, "uv_scatter_fixed: frac_bits=%d outside [-64, 62]", frac_bits);
    const int64_t N = (int64_t)B * HW;
    hipStream_t s = (hipStream_t)stream;
    if (!plan) {
        hipLaunchKernelGGL(k_sb_direct, dim3((unsigned)cdiv64(N, 256)), dim3(256), 0, s, values, uv, mask_idx, N, C, T, frac_bits, (long long *)acc);
        CTX_CHECK_LAUNCH("uv_scatter_fixed(direct)");
        return CTX_OK;
    }
    CTX_REQUIRE(C <= SB_MAXC, "uv_scatter_fixed: C=%d beyond %d with a plan (pass plan = NULL)", C, SB_MAXC);
    const int ntx = cdiv(T, SB_TS), ntiles = ntx * ntx;
    SbPlan p = sb_map(const_cast<void *>(plan), ntiles, sb_cap(N));
    hipLaunchKernelGGL(k_sb_accum<true>, dim3((unsigned)p.maxchunks), dim3(256), 0, s, p.h, values, uv, C, T, ntx, p.nch, p.ctile, p.cbeg, p.cn, p.entries,
                       (long long *)acc, (float *)nullptr, (const SbCtrl *)nullptr, frac_bits);
    CTX_CHECK_LAUNCH("uv_scatter_fixed");
    return CTX_OK;
}

extern "C" int32_t ctx_fixed_to_float(const int64_t *acc, int64_t n, int32_t frac_bits, int32_t accumulate, float *out, ctx_stream_t stream)
{
    CTX_REQUIRE(acc && out && n > 0, "fixed_to_float: bad args");
    hipLaunchKernelGGL(k_fixed_to_float, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, (const long long *)acc, n, frac_bits, accumulate, out);
    CTX_CHECK_LAUNCH("fixed_to_float");
    return CTX_OK;
}

/* 1 when a scatter found (uv, mask) changed since the plan was built (its output was poisoned with NaN).  Synchronises the stream. */
extern "C" int32_t ctx_texmap_plan_stale(const void *plan, ctx_stream_t stream)
{
    CTX_REQUIRE(plan, "texmap_plan_stale: bad args");
    SbHeader hh;
    if (hipMemcpyAsync(&hh, plan, sizeof(hh), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return CTX_E_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return CTX_E_LAUNCH;
    return hh.stale ? 1 : 0;
}

// Raster / UV / view-weight kernels for gfx950.  HBM-bound integer+float32 work: no MFMA here.
// Compiled with -ffp-contract=off: every float op below is one IEEE binary32 op in source order,
// mirroring oracle/geometry_ref.c so face indices AND interpolated floats compare bit-exactly.
//
// Reference call sites replaced (see include/ctx_nerf.h for the per-entry citations):
//   src/models/render.py:112-157, src/models/textured_mesh.py:167-190, src/training/trainer.py:155-249
#include "common.h"
#include <math.h>

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// prepare_vertices: two tiny kernels (vertex transform, then per-face gather + normal).
__global__ void k_vertex_transform(const float *__restrict__ verts, const float *__restrict__ cam,
                                   const float *__restrict__ proj3, int V, float *__restrict__ vc,
                                   float *__restrict__ vi)
{
    int b = blockIdx.y;
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *M = cam + (size_t)b * 12;
    const float *p = verts + ((size_t)b * V + v) * 3;
    float x = p[0], y = p[1], z = p[2];
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc = x * M[0 * 3 + k];
        acc = acc + y * M[1 * 3 + k];
        acc = acc + z * M[2 * 3 + k];
        acc = acc + M[3 * 3 + k];
        o[k] = acc;
    }
    float *q = vc + ((size_t)b * V + v) * 3;
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
    float px = o[0] * proj3[0], py = o[1] * proj3[1], pz = o[2] * proj3[2];
    vi[((size_t)b * V + v) * 2 + 0] = px / pz;
    vi[((size_t)b * V + v) * 2 + 1] = py / pz;
}

__global__ void k_face_gather(const float *__restrict__ vc, const float *__restrict__ vi,
                              const int64_t *__restrict__ faces, int V, int F,
                              float *__restrict__ fv_cam, float *__restrict__ fv_img,
                              float *__restrict__ fnorm)
{
    int b = blockIdx.y;
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float p[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int64_t vid = faces[(size_t)f * 3 + k];
        const float *s = vc + ((size_t)b * V + vid) * 3;
        p[k][0] = s[0]; p[k][1] = s[1]; p[k][2] = s[2];
        float *d = fv_cam + (((size_t)b * F + f) * 3 + k) * 3;
        d[0] = p[k][0]; d[1] = p[k][1]; d[2] = p[k][2];
        const float *si = vi + ((size_t)b * V + vid) * 2;
        float *di = fv_img + (((size_t)b * F + f) * 3 + k) * 2;
        di[0] = si[0]; di[1] = si[1];
    }
    float e0x = p[1][0] - p[0][0], e0y = p[1][1] - p[0][1], e0z = p[1][2] - p[0][2];
    float e1x = p[2][0] - p[0][0], e1y = p[2][1] - p[0][1], e1z = p[2][2] - p[0][2];
    float nx = e0y * e1z - e0z * e1y;
    float ny = e0z * e1x - e0x * e1z;
    float nz = e0x * e1y - e0y * e1x;
    float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    float d = len + 1e-10f;
    float *o = fnorm + ((size_t)b * F + f) * 3;
    o[0] = nx / d; o[1] = ny / d; o[2] = nz / d;
}

extern "C" int32_t ctx_prepare_vertices(const float *verts, const int64_t *faces, const float *cam,
                                        const float *proj3, int32_t B, int32_t V, int32_t F,
                                        float *fv_cam, float *fv_img, float *fnorm, void *ws,
                                        ctx_stream_t stream)
{
    CTX_REQUIRE(verts && faces && cam && proj3 && fv_cam && fv_img && fnorm && ws, "prepare_vertices: null pointer");
    CTX_REQUIRE(B > 0 && V > 0 && F > 0, "prepare_vertices: bad sizes B=%d V=%d F=%d", B, V, F);
    hipStream_t s = (hipStream_t)stream;
    float *vc = (float *)ws;
    float *vi = vc + (size_t)B * V * 3;
    hipLaunchKernelGGL(k_vertex_transform, dim3(cdiv(V, 256), B), dim3(256), 0, s, verts, cam, proj3, V, vc, vi);
    hipLaunchKernelGGL(k_face_gather, dim3(cdiv(F, 256), B), dim3(256), 0, s, vc, vi, faces, V, F, fv_cam, fv_img, fnorm);
    CTX_CHECK_LAUNCH("prepare_vertices");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Rasteriser.  Two kernels:
//   k_raster_bin : one workgroup per 64x64-pixel coarse tile scans all F faces (bbox vs tile) and
//                  appends surviving face ids to a per-tile list in the workspace.
//   k_raster     : one workgroup per 32x8 fine tile (one pixel per lane, 256 B of int64 indices per
//                  row) culls its coarse list to an LDS record list, then every pixel walks the
//                  LDS list.  Winner = max z, ties -> lowest face id (== "first face wins under a
//                  strict >" of the brute-force order), so list order is irrelevant.
#define COARSE 64
#define FT_W 32
#define FT_H 8

struct __attribute__((aligned(16))) FaceRec {   // 64 B, bbox first so it is one aligned 16-byte load
    float xmin, xmax, ymin, ymax;
    float ax, ay, bx, by, cx, cy;
    float z0, z1, z2;
    int id;
    int pad0, pad1;
};
static_assert(sizeof(FaceRec) == 64, "FaceRec must be 64 bytes");

__device__ __forceinline__ void load_face(const float *__restrict__ fxy, const float *__restrict__ fz,
                                          int zstride, float mult, FaceRec &r)
{
    r.ax = fxy[0] * mult; r.ay = fxy[1] * mult; r.bx = fxy[2] * mult;
    r.by = fxy[3] * mult; r.cx = fxy[4] * mult; r.cy = fxy[5] * mult;
    r.z0 = fz[0]; r.z1 = fz[zstride]; r.z2 = fz[2 * zstride];
    r.xmin = fminf(fminf(r.ax, r.bx), r.cx); r.xmax = fmaxf(fmaxf(r.ax, r.bx), r.cx);
    r.ymin = fminf(fminf(r.ay, r.by), r.cy); r.ymax = fmaxf(fmaxf(r.ay, r.by), r.cy);
}

__device__ __forceinline__ float pix_x(int i, int W, float mult) { return (mult / (float)W) * (float)(2 * i + 1 - W); }
__device__ __forceinline__ float pix_y(int j, int H, float mult) { return (mult / (float)H) * (float)(H - 2 * j - 1); }

// rect of pixels [i0,i1] x [j0,j1] (inclusive) can contain a covered pixel only if this is true
__device__ __forceinline__ bool rect_overlap(const FaceRec &r, float xlo, float xhi, float ylo, float yhi)
{
    return !(xhi < r.xmin || xlo >= r.xmax || yhi < r.ymin || ylo >= r.ymax);
}

// per-(view, face) record: scaled vertices, z, bbox — computed once, read as four 16-byte vectors afterwards
__global__ __launch_bounds__(256) void k_raster_setup(const float *__restrict__ fz, int zstride, const float *__restrict__ fxy,
                                                      int F, float mult, FaceRec *__restrict__ recs, float4 *__restrict__ bbox)
{
    int b = blockIdx.y;
    int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    FaceRec r;
    load_face(fxy + ((size_t)b * F + f) * 6, fz + ((size_t)b * F + f) * 3 * zstride, zstride, mult, r);
    r.id = f; r.pad0 = 0; r.pad1 = 0;
    recs[(size_t)b * F + f] = r;
    bbox[(size_t)b * F + f] = make_float4(r.xmin, r.xmax, r.ymin, r.ymax);   // compact copy: 16 B per face for the culls
}

// grid (coarse tiles, face-range groups, views): with few views a tile's face scan is cut into `gridDim.y` ranges so that the launch
// still has >= ~1000 workgroups; a group appends its survivors to the tile's list behind one global reservation per wave (the list
// order is then arbitrary: the fine kernel's winner rule does not depend on it).  `counts` is zeroed by the caller.
__global__ __launch_bounds__(256) void k_raster_bin(int H, int W, const float4 *__restrict__ bbox, int F, float mult,
                                                    int ntx, int nty, int *__restrict__ lists,
                                                    int *__restrict__ counts)
{
    int b = blockIdx.z;
    int tile = blockIdx.x;
    int tx = tile % ntx, ty = tile / ntx;
    int i0 = tx * COARSE, i1 = min(W, i0 + COARSE) - 1;
    int j0 = ty * COARSE, j1 = min(H, j0 + COARSE) - 1;
    float xlo = pix_x(i0, W, mult), xhi = pix_x(i1, W, mult);
    float yhi = pix_y(j0, H, mult), ylo = pix_y(j1, H, mult);
    int *list = lists + ((size_t)b * ntx * nty + tile) * F;
    int *cnt = counts + (size_t)b * ntx * nty + tile;
    const float4 *bbv = bbox + (size_t)b * F;
    const int per = ((F + (int)gridDim.y - 1) / (int)gridDim.y + 255) & ~255;
    const int fbeg = blockIdx.y * per, fend = min(F, fbeg + per);
    for (int f0 = fbeg; f0 < fend; f0 += 256) {
        int f = f0 + threadIdx.x;
        bool keep = false;
        if (f < fend) {
            const float4 bb = bbv[f];                             // xmin, xmax, ymin, ymax
            keep = !(xhi < bb.x || xlo >= bb.y || yhi < bb.z || ylo >= bb.w);
        }
        unsigned long long m = __ballot(keep);
        int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(cnt, __popcll(m));
        base = __shfl(base, 0, 64);
        if (keep) list[base + __popcll(m & ((1ull << lane) - 1))] = f;
    }
}

template <bool FUSED>
__global__ __launch_bounds__(256) void k_raster(int H, int W, const float *__restrict__ fz, int zstride,
                                                const FaceRec *__restrict__ recs, const float4 *__restrict__ bbox,
                                                const float *__restrict__ feat,
                                                int featB, int C, const float *__restrict__ fnorm, int F,
                                                float mult, float eps, int ntx, int nty,
                                                const int *__restrict__ lists, const int *__restrict__ counts,
                                                float *__restrict__ out, float *__restrict__ out_uv,
                                                int64_t *__restrict__ face_idx, float *__restrict__ normals)
{
    __shared__ __attribute__((aligned(16))) FaceRec s_rec[256];
    __shared__ int s_n;
    int b = blockIdx.z;
    int fi0 = blockIdx.x * FT_W, fj0 = blockIdx.y * FT_H;
    int lx = threadIdx.x % FT_W, ly = threadIdx.x / FT_W;
    int i = fi0 + lx, j = fj0 + ly;
    bool active = (i < W) && (j < H);
    int fi1 = min(W, fi0 + FT_W) - 1, fj1 = min(H, fj0 + FT_H) - 1;
    float txlo = pix_x(fi0, W, mult), txhi = pix_x(fi1, W, mult);
    float tyhi = pix_y(fj0, H, mult), tylo = pix_y(fj1, H, mult);
    float x0 = pix_x(i, W, mult), y0 = pix_y(j, H, mult);
    int ctile = (fj0 / COARSE) * ntx + (fi0 / COARSE);
    const int *list = lists + ((size_t)b * ntx * nty + ctile) * F;
    int n = counts[(size_t)b * ntx * nty + ctile];
    const FaceRec *rb = recs + (size_t)b * F;
    const float *zb = fz + (size_t)b * F * 3 * zstride;

    float best = -INFINITY, bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
    int bi = -1;
    for (int c0 = 0; c0 < n; c0 += 256) {
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        int k = c0 + threadIdx.x;
        if (k < n) {
            int f = list[k];
            const float4 bb = bbox[(size_t)b * F + f];
            if (!(txhi < bb.x || txlo >= bb.y || tyhi < bb.z || tylo >= bb.w)) {
                int slot = atomicAdd(&s_n, 1);
                const float4 *src = (const float4 *)&rb[f];
                float4 *dst = (float4 *)&s_rec[slot];
                dst[0] = bb; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
            }
        }
        __syncthreads();
        int m = s_n;
        if (active) {
            for (int q = 0; q < m; ++q) {
                const FaceRec &r = s_rec[q];
                if (x0 < r.xmin || x0 >= r.xmax || y0 < r.ymin || y0 >= r.ymax) continue;
                float mm = r.bx - r.ax, p = r.by - r.ay, nn = r.cx - r.ax, qq = r.cy - r.ay;
                float s = x0 - r.ax, t = y0 - r.ay;
                float k1 = s * qq - nn * t;
                float k2 = mm * t - s * p;
                float k3 = mm * qq - nn * p;
                float den = k3 + eps;
                float w1 = k1 / den;
                float w2 = k2 / den;
                float w0 = (1.0f - w1) - w2;
                if (w0 < 0.0f || w1 < 0.0f || w2 < 0.0f) continue;
                float z = (w0 * r.z0 + w1 * r.z1) + w2 * r.z2;
                if (z > best || (z == best && r.id < bi)) {
                    best = z; bi = r.id; bw0 = w0; bw1 = w1; bw2 = w2;
                }
            }
        }
    }
    if (!active) return;
    size_t pix = ((size_t)b * H + j) * W + i;
    face_idx[pix] = (int64_t)bi;
    if (FUSED) {
        float d = 0.f, u = 0.f, v = 0.f;
        if (bi >= 0) {
            const float *zf = zb + (size_t)bi * 3 * zstride;
            d = (bw0 * zf[0] + bw1 * zf[zstride]) + bw2 * zf[2 * zstride];
            const float *ff = feat + ((size_t)(featB == 1 ? 0 : b) * F + bi) * 6;
            u = (bw0 * ff[0] + bw1 * ff[2]) + bw2 * ff[4];
            v = (bw0 * ff[1] + bw1 * ff[3]) + bw2 * ff[5];
        }
        out[pix] = d;
        out_uv[pix * 2 + 0] = u;
        out_uv[pix * 2 + 1] = v;
        if (normals) {
            int fnb = bi >= 0 ? bi : F - 1;   // python negative index: -1 -> last face
            const float *nf = fnorm + ((size_t)b * F + fnb) * 3;
            normals[pix * 3 + 0] = nf[0]; normals[pix * 3 + 1] = nf[1]; normals[pix * 3 + 2] = nf[2];
        }
    } else {
        for (int c = 0; c < C; ++c) {
            float val = 0.f;
            if (bi >= 0) {
                const float *ff = feat + ((size_t)(featB == 1 ? 0 : b) * F + bi) * 3 * C;
                val = (bw0 * ff[0 * C + c] + bw1 * ff[1 * C + c]) + bw2 * ff[2 * C + c];
            }
            out[pix * C + c] = val;
        }
    }
}

extern "C" int64_t ctx_rasterize_ws_bytes(int32_t H, int32_t W, int32_t B, int32_t F)
{
    int64_t nt = (int64_t)cdiv(W, COARSE) * cdiv(H, COARSE);
    return ((int64_t)B * nt * F + (int64_t)B * nt) * 4 + (int64_t)B * F * (64 + 16) + 512;
}

static int32_t raster_common(bool fused, int H, int W, const float *fz, int zstride, const float *fxy,
                             const float *feat, int featB, int C, const float *fnorm, int B, int F, float mult,
                             float eps, float *out, float *out_uv, int64_t *face_idx, float *normals, void *ws,
                             int64_t ws_bytes, hipStream_t s)
{
    CTX_REQUIRE(fz && fxy && feat && out && face_idx && ws, "rasterize: null pointer");
    CTX_REQUIRE(H > 0 && W > 0 && B > 0 && F > 0 && H <= 16384 && W <= 16384, "rasterize: bad sizes H=%d W=%d B=%d F=%d", H, W, B, F);
    CTX_REQUIRE(ws_bytes >= ctx_rasterize_ws_bytes(H, W, B, F), "rasterize: workspace too small (%lld < %lld)",
                (long long)ws_bytes, (long long)ctx_rasterize_ws_bytes(H, W, B, F));
    int ntx = cdiv(W, COARSE), nty = cdiv(H, COARSE);
    FaceRec *recs = (FaceRec *)(((uintptr_t)ws + 63) & ~(uintptr_t)63);
    float4 *bbox = (float4 *)(recs + (size_t)B * F);
    int *lists = (int *)(bbox + (size_t)B * F);
    int *counts = lists + (size_t)B * ntx * nty * F;
    hipLaunchKernelGGL(k_raster_setup, dim3(cdiv(F, 256), B), dim3(256), 0, s, fz, zstride, fxy, F, mult, recs, bbox);
    (void)hipMemsetAsync(counts, 0, (size_t)B * ntx * nty * sizeof(int), s);
    int G = cdiv(1024, ntx * nty * B);                          // face-range groups per tile: keep the bin launch >= ~1000 workgroups
    G = max(1, min(min(G, 8), cdiv(F, 256)));
    hipLaunchKernelGGL(k_raster_bin, dim3(ntx * nty, G, B), dim3(256), 0, s, H, W, bbox, F, mult, ntx, nty, lists, counts);
    dim3 grid(cdiv(W, FT_W), cdiv(H, FT_H), B);
    if (fused)
        hipLaunchKernelGGL(k_raster<true>, grid, dim3(256), 0, s, H, W, fz, zstride, recs, bbox, feat, featB, C, fnorm, F, mult, eps,
                           ntx, nty, lists, counts, out, out_uv, face_idx, normals);
    else
        hipLaunchKernelGGL(k_raster<false>, grid, dim3(256), 0, s, H, W, fz, zstride, recs, bbox, feat, featB, C, fnorm, F, mult, eps,
                           ntx, nty, lists, counts, out, out_uv, face_idx, normals);
    CTX_CHECK_LAUNCH("rasterize");
    return CTX_OK;
}

extern "C" int32_t ctx_rasterize_fwd(int32_t H, int32_t W, const float *face_z, const float *face_xy,
                                     const float *feat, int32_t B, int32_t F, int32_t C, float multiplier,
                                     float eps, float *out, int64_t *face_idx, void *ws, int64_t ws_bytes,
                                     ctx_stream_t stream)
{
    CTX_REQUIRE(C >= 1 && C <= 64, "rasterize: C=%d unsupported", C);
    return raster_common(false, H, W, face_z, 1, face_xy, feat, B, C, nullptr, B, F, multiplier, eps, out, nullptr,
                         face_idx, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int32_t ctx_rasterize_fused(int32_t H, int32_t W, const float *fv_cam, const float *face_xy,
                                       const float *uv_attr, int32_t Bu, const float *fnorm, int32_t B, int32_t F,
                                       float multiplier, float eps, float *depth, float *uv, int64_t *face_idx,
                                       float *normals, void *ws, int64_t ws_bytes, ctx_stream_t stream)
{
    CTX_REQUIRE(uv && (Bu == 1 || Bu == B), "rasterize_fused: uv null or Bu=%d not in {1,%d}", Bu, B);
    CTX_REQUIRE((normals == nullptr) || (fnorm != nullptr), "rasterize_fused: normals requested without fnorm");
    // z of vertex k of face f is fv_cam[b,f,k,2]: base +2, stride 3
    return raster_common(true, H, W, fv_cam + 2, 3, face_xy, uv_attr, Bu, 2, fnorm, B, F, multiplier, eps, depth, uv,
                         face_idx, normals, ws, ws_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// normalize_multiple_depth: partial min/max per (view, block) -> normalise.
#define ND_BLOCKS 256
__global__ __launch_bounds__(256) void k_depth_minmax(const float *__restrict__ d, int HW, float *__restrict__ part,
                                                      int *__restrict__ flags)
{
    int b = blockIdx.y;
    const float *p = d + (size_t)b * HW;
    float mn = INFINITY, mx = -INFINITY;
    int pos = 0, any = 0;
    const int n4 = HW >> 2;
    const float4 *p4 = (const float4 *)p;            // views start 16-B aligned when HW % 4 == 0 (else scalar path)
    if ((HW & 3) == 0) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += ND_BLOCKS * 256) {
            float4 v = p4[i];
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (e[j] > 0.f) pos = 1;
                if (e[j] != 0.f) { any = 1; mn = fminf(mn, e[j]); mx = fmaxf(mx, e[j]); }
            }
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += ND_BLOCKS * 256) {
            float v = p[i];
            if (v > 0.f) pos = 1;
            if (v != 0.f) { any = 1; mn = fminf(mn, v); mx = fmaxf(mx, v); }
        }
    }
    mn = wave_min(mn); mx = wave_max(mx);
    __shared__ float s_mn[4], s_mx[4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_mn[w] = mn; s_mx[w] = mx; }
    int bpos = __syncthreads_or(pos), bany = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
        mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        part[((size_t)b * ND_BLOCKS + blockIdx.x) * 2 + 0] = mn;
        part[((size_t)b * ND_BLOCKS + blockIdx.x) * 2 + 1] = mx;
        // one flag word per block (no same-address atomics): bit0 = positive value seen, bit1 = non-zero seen
        flags[4 + b * ND_BLOCKS + blockIdx.x] = (bpos ? 1 : 0) | (bany ? 2 : 0);
    }
}

__global__ __launch_bounds__(256) void k_depth_apply(const float *__restrict__ d, int HW, const float *__restrict__ part,
                                                     const int *__restrict__ flags, float *__restrict__ out,
                                                     int *__restrict__ status)
{
    int b = blockIdx.y;
    float mn = part[((size_t)b * ND_BLOCKS + threadIdx.x) * 2 + 0];
    float mx = part[((size_t)b * ND_BLOCKS + threadIdx.x) * 2 + 1];
    mn = wave_min(mn); mx = wave_max(mx);
    __shared__ float s_mn[4], s_mx[4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_mn[w] = mn; s_mx[w] = mx; }
    __syncthreads();
    mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    float range = mx - mn;
    if (status && b == 0 && blockIdx.x == 0) {
        int acc = 0;
        for (int i = threadIdx.x; i < (int)gridDim.y * ND_BLOCKS; i += 256) acc |= flags[4 + i];
        int any_pos = __syncthreads_or(acc & 1), any_nz = __syncthreads_or(acc & 2);   // returns "any non-zero", not the OR
        if (threadIdx.x == 0) status[0] = any_pos ? 1 : (any_nz ? 0 : 2);
    }
    const float *p = d + (size_t)b * HW;
    float *o = out + (size_t)b * HW;
    if ((HW & 3) == 0) {
        const int n4 = HW >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
            float4 v = ((const float4 *)p)[i];
            float4 r;
            r.x = (v.x != 0.f) ? (v.x - mn) / range : v.x;
            r.y = (v.y != 0.f) ? (v.y - mn) / range : v.y;
            r.z = (v.z != 0.f) ? (v.z - mn) / range : v.z;
            r.w = (v.w != 0.f) ? (v.w - mn) / range : v.w;
            ((float4 *)o)[i] = r;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            float v = p[i];
            o[i] = (v != 0.f) ? (v - mn) / range : v;
        }
    }
}

extern "C" int64_t ctx_normalize_depth_ws_bytes(int32_t B) { return (int64_t)B * ND_BLOCKS * 3 * 4 + 64; }

extern "C" int32_t ctx_normalize_depth(const float *depth, int32_t B, int32_t HW, float *out, void *ws,
                                       int32_t *status, ctx_stream_t stream)
{
    CTX_REQUIRE(depth && out && ws && B > 0 && HW > 0, "normalize_depth: bad args");
    hipStream_t s = (hipStream_t)stream;
    int *flags = (int *)ws;                                   // [4 + B*ND_BLOCKS] ints, every word written by pass 1
    float *part = (float *)ws + 4 + (size_t)B * ND_BLOCKS;
    hipLaunchKernelGGL(k_depth_minmax, dim3(ND_BLOCKS, B), dim3(256), 0, s, depth, HW, part, flags);
    int nb = min(cdiv(HW, 1024), 1024);
    hipLaunchKernelGGL(k_depth_apply, dim3(nb, B), dim3(256), 0, s, depth, HW, part, flags, out, status);
    CTX_CHECK_LAUNCH("normalize_depth");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// texture_mapping == grid_sample(bilinear|nearest, align_corners=False, padding 'border').
__device__ __forceinline__ float src_index(float g, int size)
{
    float c = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
    return fminf((float)(size - 1), fmaxf(c, 0.0f));
}

__global__ __launch_bounds__(256) void k_texmap_fwd(const float *__restrict__ uv, const float *__restrict__ tex,
                                                    int64_t HW, int C, int T, int Bt, int mode,
                                                    const int64_t *__restrict__ mask_idx, float *__restrict__ out)
{
    int b = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        size_t pix = (size_t)b * HW + i;
        float2 q = *(const float2 *)(uv + pix * 2);
        const float *tb = tex + (Bt == 1 ? 0 : (size_t)b * C * T * T);
        float ix = src_index(q.x * 2.0f - 1.0f, T), iy = src_index((1.0f - q.y) * 2.0f - 1.0f, T);
        float msk = 1.0f;
        if (mask_idx) msk = mask_idx[pix] > -1 ? 1.0f : 0.0f;
        float *o = out + pix * C;
        if (mode == 1) {
            int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            bool in = xn >= 0 && xn < T && yn >= 0 && yn < T;
            for (int c = 0; c < C; ++c) {
                float v = in ? tb[((size_t)c * T + yn) * T + xn] : 0.0f;
                o[c] = mask_idx ? v * msk : v;
            }
            continue;
        }
        float fx = floorf(ix), fy = floorf(iy);
        int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        float wnw = ((float)x1 - ix) * ((float)y1 - iy);
        float wne = (ix - (float)x0) * ((float)y1 - iy);
        float wsw = ((float)x1 - ix) * (iy - (float)y0);
        float wse = (ix - (float)x0) * (iy - (float)y0);
        bool bx0 = x0 >= 0 && x0 < T, bx1 = x1 >= 0 && x1 < T, by0 = y0 >= 0 && y0 < T, by1 = y1 >= 0 && y1 < T;
        for (int c = 0; c < C; ++c) {
            const float *tc = tb + (size_t)c * T * T;
            float acc = 0.0f;
            if (bx0 && by0) acc = acc + tc[(size_t)y0 * T + x0] * wnw;
            if (bx1 && by0) acc = acc + tc[(size_t)y0 * T + x1] * wne;
            if (bx0 && by1) acc = acc + tc[(size_t)y1 * T + x0] * wsw;
            if (bx1 && by1) acc = acc + tc[(size_t)y1 * T + x1] * wse;
            o[c] = mask_idx ? acc * msk : acc;
        }
    }
}

__global__ __launch_bounds__(256) void k_texmap_bwd(const float *__restrict__ go, const float *__restrict__ uv,
                                                    int64_t HW, int C, int T, const int64_t *__restrict__ mask_idx,
                                                    float *__restrict__ gt)
{
    int b = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        size_t pix = (size_t)b * HW + i;
        if (mask_idx && mask_idx[pix] < 0) continue;
        float2 q = *(const float2 *)(uv + pix * 2);
        float ix = src_index(q.x * 2.0f - 1.0f, T), iy = src_index((1.0f - q.y) * 2.0f - 1.0f, T);
        float fx = floorf(ix), fy = floorf(iy);
        int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        float wnw = ((float)x1 - ix) * ((float)y1 - iy);
        float wne = (ix - (float)x0) * ((float)y1 - iy);
        float wsw = ((float)x1 - ix) * (iy - (float)y0);
        float wse = (ix - (float)x0) * (iy - (float)y0);
        bool bx0 = x0 >= 0 && x0 < T, bx1 = x1 >= 0 && x1 < T, by0 = y0 >= 0 && y0 < T, by1 = y1 >= 0 && y1 < T;
        for (int c = 0; c < C; ++c) {
            float g = go[pix * C + c];
            float *tc = gt + (size_t)c * T * T;
            if (bx0 && by0) atomicAdd(tc + (size_t)y0 * T + x0, g * wnw);
            if (bx1 && by0) atomicAdd(tc + (size_t)y0 * T + x1, g * wne);
            if (bx0 && by1) atomicAdd(tc + (size_t)y1 * T + x0, g * wsw);
            if (bx1 && by1) atomicAdd(tc + (size_t)y1 * T + x1, g * wse);
        }
    }
}

// Texel-interleaved variant for C <= 4 and one shared texture: [C,T,T] is repacked once into [T,T,4] (16 bytes per
// texel) so a bilinear tap is ONE 16-byte gather instead of C 4-byte gathers; the per-channel arithmetic and its order are
// those of k_texmap_fwd (results are bit-identical).
__global__ __launch_bounds__(256) void k_tex_pack4(const float *__restrict__ tex, int C, int T, float4 *__restrict__ packed)
{
    const int64_t n = (int64_t)T * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = tex[i];
        if (C > 1) v.y = tex[n + i];
        if (C > 2) v.z = tex[2 * n + i];
        if (C > 3) v.w = tex[3 * n + i];
        packed[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_texmap_fwd4(const float *__restrict__ uv, const float4 *__restrict__ tex, int64_t HW, int C,
                                                     int T, int mode, const int64_t *__restrict__ mask_idx, float *__restrict__ out)
{
    const int b = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const size_t pix = (size_t)b * HW + i;
        float2 q = *(const float2 *)(uv + pix * 2);
        float ix = src_index(q.x * 2.0f - 1.0f, T), iy = src_index((1.0f - q.y) * 2.0f - 1.0f, T);
        float msk = 1.0f;
        if (mask_idx) msk = mask_idx[pix] > -1 ? 1.0f : 0.0f;
        float r[4];
        if (mode == 1) {
            int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            bool in = xn >= 0 && xn < T && yn >= 0 && yn < T;
            float4 t = in ? tex[(size_t)yn * T + xn] : make_float4(0.f, 0.f, 0.f, 0.f);
            r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
        } else {
            float fx = floorf(ix), fy = floorf(iy);
            int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            float wnw = ((float)x1 - ix) * ((float)y1 - iy);
            float wne = (ix - (float)x0) * ((float)y1 - iy);
            float wsw = ((float)x1 - ix) * (iy - (float)y0);
            float wse = (ix - (float)x0) * (iy - (float)y0);
            bool bx0 = x0 >= 0 && x0 < T, bx1 = x1 >= 0 && x1 < T, by0 = y0 >= 0 && y0 < T, by1 = y1 >= 0 && y1 < T;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            // unconditional (clamped) gathers, then the same guarded accumulation order as the planar kernel
            float4 tnw = tex[(size_t)min(max(y0, 0), T - 1) * T + min(max(x0, 0), T - 1)];
            float4 tne = tex[(size_t)min(max(y0, 0), T - 1) * T + min(max(x1, 0), T - 1)];
            float4 tsw = tex[(size_t)min(max(y1, 0), T - 1) * T + min(max(x0, 0), T - 1)];
            float4 tse = tex[(size_t)min(max(y1, 0), T - 1) * T + min(max(x1, 0), T - 1)];
            (void)z4;
            const float a_nw[4] = {tnw.x, tnw.y, tnw.z, tnw.w}, a_ne[4] = {tne.x, tne.y, tne.z, tne.w};
            const float a_sw[4] = {tsw.x, tsw.y, tsw.z, tsw.w}, a_se[4] = {tse.x, tse.y, tse.z, tse.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
                if (bx0 && by0) acc = acc + a_nw[c] * wnw;
                if (bx1 && by0) acc = acc + a_ne[c] * wne;
                if (bx0 && by1) acc = acc + a_sw[c] * wsw;
                if (bx1 && by1) acc = acc + a_se[c] * wse;
                r[c] = acc;
            }
        }
        float *o = out + pix * C;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) o[c] = mask_idx ? r[c] * msk : r[c];
    }
}

extern "C" int32_t ctx_texture_pack4(const float *tex, int32_t C, int32_t T, float *packed, ctx_stream_t stream)
{
    CTX_REQUIRE(tex && packed && C >= 1 && C <= 4 && T > 0, "texture_pack4: bad args C=%d T=%d", C, T);
    int nb = min(cdiv(T * T, 256), 4096);
    hipLaunchKernelGGL(k_tex_pack4, dim3(nb), dim3(256), 0, (hipStream_t)stream, tex, C, T, (float4 *)packed);
    CTX_CHECK_LAUNCH("texture_pack4");
    return CTX_OK;
}

extern "C" int32_t ctx_texture_mapping_packed_fwd(const float *uv, const float *packed, int32_t B, int32_t HW, int32_t C, int32_t T,
                                                  int32_t mode, const int64_t *mask_idx, float *out, ctx_stream_t stream)
{
    CTX_REQUIRE(uv && packed && out, "texture_mapping_packed: null pointer");
    CTX_REQUIRE(B > 0 && HW > 0 && C >= 1 && C <= 4 && T > 0 && (mode == 0 || mode == 1), "texture_mapping_packed: bad args");
    int nb = min(cdiv(HW, 256), 4096);
    hipLaunchKernelGGL(k_texmap_fwd4, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, uv, (const float4 *)packed, (int64_t)HW, C, T, mode,
                       mask_idx, out);
    CTX_CHECK_LAUNCH("texture_mapping_packed_fwd");
    return CTX_OK;
}

extern "C" int32_t ctx_texture_mapping_fwd(const float *uv, const float *tex, int32_t B, int32_t HW, int32_t C,
                                           int32_t T, int32_t Bt, int32_t mode, const int64_t *mask_idx, float *out,
                                           ctx_stream_t stream)
{
    CTX_REQUIRE(uv && tex && out, "texture_mapping: null pointer");
    CTX_REQUIRE(B > 0 && HW > 0 && C > 0 && T > 0 && (Bt == 1 || Bt == B) && (mode == 0 || mode == 1),
                "texture_mapping: bad args B=%d HW=%d C=%d T=%d Bt=%d mode=%d", B, HW, C, T, Bt, mode);
    int nb = min(cdiv(HW, 256), 4096);
    hipLaunchKernelGGL(k_texmap_fwd, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, uv, tex, (int64_t)HW, C, T, Bt, mode, mask_idx, out);
    CTX_CHECK_LAUNCH("texture_mapping_fwd");
    return CTX_OK;
}

extern "C" int32_t ctx_texture_mapping_bwd(const float *grad_out, const float *uv, int32_t B, int32_t HW, int32_t C,
                                           int32_t T, const int64_t *mask_idx, float *grad_tex, ctx_stream_t stream)
{
    CTX_REQUIRE(grad_out && uv && grad_tex && B > 0 && HW > 0 && C > 0 && T > 0, "texture_mapping_bwd: bad args");
    int nb = min(cdiv(HW, 256), 4096);
    hipLaunchKernelGGL(k_texmap_bwd, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, grad_out, uv, (int64_t)HW, C, T, mask_idx, grad_tex);
    CTX_CHECK_LAUNCH("texture_mapping_bwd");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// View weights.  max over pixels of fnz[b, face_idx[b,p]] per face == max over the views in which
// the face is visible, so phase 0 only marks (view, face) visibility (idempotent byte stores, no
// atomics) and then reduces B values per face.
__global__ __launch_bounds__(256) void k_vw_mark(const int64_t *__restrict__ face_idx, int64_t HW, int F,
                                                 uint8_t *__restrict__ vis)
{
    int b = blockIdx.y;
    const int64_t *p = face_idx + (size_t)b * HW;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < HW; i += (int64_t)gridDim.x * 512) {
        int64_t f0 = p[i];
        int64_t f1 = (i + 1 < HW) ? p[i + 1] : -1;
        if (f0 >= 0 && f0 < F) vis[(size_t)b * F + f0] = 1;      // a face id outside [0, F) must not become a stray store
        if (f1 >= 0 && f1 < F) vis[(size_t)b * F + f1] = 1;
    }
}

__global__ void k_vw_reduce(const uint8_t *__restrict__ vis, const float *__restrict__ fnz, int B, int F,
                            float *__restrict__ max_z)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float m = max_z[f];
    for (int b = 0; b < B; ++b)
        if (vis[(size_t)b * F + f]) {
            float z = fnz[(size_t)b * F + f];
            if (z > m) m = z;
        }
    max_z[f] = m;
}

__global__ __launch_bounds__(256) void k_vw_mask(const int64_t *__restrict__ face_idx, const float *__restrict__ fnz,
                                                 const float *__restrict__ max_z, int64_t HW, int F,
                                                 uint8_t *__restrict__ mask)
{
    int b = blockIdx.y;
    const int64_t *p = face_idx + (size_t)b * HW;
    uint8_t *o = mask + (size_t)b * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        int64_t f = p[i];
        uint8_t m = 1;
        if (f >= 0 && f < F) m = !(fnz[(size_t)b * F + f] < max_z[f]);
        o[i] = m;
    }
}

extern "C" int32_t ctx_view_weights_max(const int64_t *face_idx, const float *fnz, int32_t B, int32_t HW, int32_t F,
                                        float *max_z, void *vis_ws, ctx_stream_t stream)
{
    CTX_REQUIRE(face_idx && fnz && max_z && vis_ws && B > 0 && HW > 0 && F > 0, "view_weights_max: bad args");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(vis_ws, 0, (size_t)B * F, s);
    int nb = min(cdiv(HW, 512), 2048);
    hipLaunchKernelGGL(k_vw_mark, dim3(nb, B), dim3(256), 0, s, face_idx, (int64_t)HW, F, (uint8_t *)vis_ws);
    hipLaunchKernelGGL(k_vw_reduce, dim3(cdiv(F, 256)), dim3(256), 0, s, (const uint8_t *)vis_ws, fnz, B, F, max_z);
    CTX_CHECK_LAUNCH("view_weights_max");
    return CTX_OK;
}

extern "C" int32_t ctx_view_weights_mask(const int64_t *face_idx, const float *fnz, const float *max_z, int32_t B,
                                         int32_t HW, int32_t F, uint8_t *mask, ctx_stream_t stream)
{
    CTX_REQUIRE(face_idx && fnz && max_z && mask && B > 0 && HW > 0 && F > 0, "view_weights_mask: bad args");
    int nb = min(cdiv(HW, 256), 4096);
    hipLaunchKernelGGL(k_vw_mask, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, face_idx, fnz, max_z, (int64_t)HW, F, mask);
    CTX_CHECK_LAUNCH("view_weights_mask");
    return CTX_OK;
}

// create_face_view_map: ordered stream compaction (count -> scan -> write).
#define FVM_BLK 1024
__global__ __launch_bounds__(FVM_BLK) void k_fvm_count(const int64_t *__restrict__ fi, int64_t N, int *__restrict__ counts)
{
    int64_t i = (int64_t)blockIdx.x * FVM_BLK + threadIdx.x;
    bool v = i < N && fi[i] >= 0;
    unsigned long long m = __ballot(v);
    __shared__ int s[FVM_BLK / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < FVM_BLK / 64; ++k) t += s[k];
        counts[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void k_fvm_scan(int *__restrict__ counts, int64_t nblk, int64_t *__restrict__ base,
                                                   int64_t *__restrict__ n_rows)
{
    // single workgroup: every thread sums one contiguous segment, one Hillis-Steele pass over the 1024 segment sums,
    // then each thread writes its segment's exclusive prefix
    __shared__ int64_t s[1024];
    const int64_t per = (nblk + 1023) / 1024;
    const int64_t i0 = (int64_t)threadIdx.x * per, i1 = i0 + per < nblk ? i0 + per : nblk;
    int64_t sum = 0;
    for (int64_t i = i0; i < i1; ++i) sum += counts[i];
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int64_t t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = s[threadIdx.x] - sum;
    for (int64_t i = i0; i < i1; ++i) { base[i] = run; run += counts[i]; }
    if (threadIdx.x == 1023) n_rows[0] = s[1023];
}

__global__ __launch_bounds__(FVM_BLK) void k_fvm_write(const int64_t *__restrict__ fi, int64_t N, int64_t HW, int W,
                                                   const int64_t *__restrict__ base, int64_t *__restrict__ rows)
{
    int64_t i = (int64_t)blockIdx.x * FVM_BLK + threadIdx.x;
    int64_t f = i < N ? fi[i] : -1;
    bool v = f >= 0;
    unsigned long long m = __ballot(v);
    __shared__ int s[FVM_BLK / 64];
    int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s[w] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int k = 0; k < w; ++k) off += s[k];
    if (v) {
        int64_t r = base[blockIdx.x] + off + __popcll(m & ((1ull << lane) - 1));
        int64_t view = i / HW, pix = i % HW;
        int64_t *o = rows + r * 4;
        o[0] = f; o[1] = view; o[2] = pix / W; o[3] = pix % W;
    }
}

extern "C" int64_t ctx_face_view_map_ws_bytes(int32_t B, int32_t H, int32_t W)
{
    int64_t nblk = cdiv64((int64_t)B * H * W, FVM_BLK);
    return nblk * 4 + nblk * 8 + 64;
}

extern "C" int32_t ctx_face_view_map(const int64_t *face_idx, int32_t B, int32_t H, int32_t W, int64_t *rows,
                                     int64_t *n_rows, void *ws, ctx_stream_t stream)
{
    CTX_REQUIRE(face_idx && rows && n_rows && ws && B > 0 && H > 0 && W > 0, "face_view_map: bad args");
    hipStream_t s = (hipStream_t)stream;
    int64_t N = (int64_t)B * H * W, nblk = cdiv64(N, FVM_BLK);
    int64_t *base = (int64_t *)ws;
    int *counts = (int *)(base + nblk);
    hipLaunchKernelGGL(k_fvm_count, dim3((unsigned)nblk), dim3(FVM_BLK), 0, s, face_idx, N, counts);
    hipLaunchKernelGGL(k_fvm_scan, dim3(1), dim3(1024), 0, s, counts, nblk, base, n_rows);
    hipLaunchKernelGGL(k_fvm_write, dim3((unsigned)nblk), dim3(FVM_BLK), 0, s, face_idx, N, (int64_t)H * W, W, base, rows);
    CTX_CHECK_LAUNCH("face_view_map");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Positional encoding (unfused form, kept for API parity with get_embedder; the MLP kernel fuses it).
__global__ __launch_bounds__(256) void k_embed(const float *__restrict__ x, int64_t N, int d, int L, float *__restrict__ out)
{
    int od = d * (1 + 2 * L);
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        float *o = out + n * od;
        for (int c = 0; c < d; ++c) {
            float v = x[n * d + c];
            o[c] = v;
            float fr = 1.0f;
            for (int l = 0; l < L; ++l) {
                float a = v * fr;
                o[d + (2 * l) * d + c] = sinf(a);
                o[d + (2 * l + 1) * d + c] = cosf(a);
                fr = fr * 2.0f;
            }
        }
    }
}

extern "C" int32_t ctx_embed_fwd(const float *x, int64_t N, int32_t d, int32_t L, float *out, ctx_stream_t stream)
{
    CTX_REQUIRE(x && out && N > 0 && d > 0 && L >= 0, "embed: bad args");
    int nb = (int)((N + 255) / 256 < 4096 ? (N + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_embed, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, N, d, L, out);
    CTX_CHECK_LAUNCH("embed");
    return CTX_OK;
}

// ------------------------------------------------------------------------------------------------
// Rays + volume-render compositing.
__global__ __launch_bounds__(256) void k_get_rays(int H, int W, float fx, float fy, float cx, float cy,
                                                  const float *__restrict__ c2w, float *__restrict__ ro,
                                                  float *__restrict__ rd)
{
    int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= (int64_t)H * W) return;
    int j = (int)(n / W), i = (int)(n % W);
    float dx = ((float)i - cx) / fx, dy = -((float)j - cy) / fy, dz = -1.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        // torch.sum(dirs[..., None, :] * c2w[:3,:3], -1): sequential left-to-right sum
        rd[n * 3 + r] = (dx * c2w[r * 4 + 0] + dy * c2w[r * 4 + 1]) + dz * c2w[r * 4 + 2];
        ro[n * 3 + r] = c2w[r * 4 + 3];
    }
}

extern "C" int32_t ctx_get_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float *c2w,
                                float *rays_o, float *rays_d, ctx_stream_t stream)
{
    CTX_REQUIRE(c2w && rays_o && rays_d && H > 0 && W > 0, "get_rays: bad args");
    hipLaunchKernelGGL(k_get_rays, dim3((unsigned)cdiv64((int64_t)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       H, W, fx, fy, cx, cy, c2w, rays_o, rays_d);
    CTX_CHECK_LAUNCH("get_rays");
    return CTX_OK;
}

// One wavefront per ray: lane s holds sample s of the current 64-sample chunk.  Transmittance is an
// exclusive prefix product across lanes (6 DPP steps), colour/depth/acc are wave sums on the DPP path.  The next chunk's (or next
// ray's first) loads are issued before the current chunk is reduced, so HBM latency overlaps the shuffle chain.
__global__ __launch_bounds__(256) void k_composite(const float4 *__restrict__ raw, const float *__restrict__ z,
                                                   const float *__restrict__ rays_d, int64_t R, int S, int white,
                                                   float *__restrict__ rgb, float *__restrict__ disp,
                                                   float *__restrict__ acc, float *__restrict__ weights,
                                                   float *__restrict__ depth)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const int nch = (S + 63) >> 6;
    // software pipeline over the flattened (ray, chunk) sequence of this wave
    int64_t r = wave;
    int ch = 0;
    float4 q_n = make_float4(0.f, 0.f, 0.f, 0.f);
    float z_n = 0.f, zx_n = 0.f;
    auto fetch = [&](int64_t rr, int cc) {
        int s = cc * 64 + lane;
        bool ok = rr < R && s < S;
        q_n = ok ? raw[rr * S + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        z_n = ok ? z[rr * S + s] : 0.f;
        zx_n = (rr < R && lane == 63 && s + 1 < S) ? z[rr * S + s + 1] : 0.f;      // first depth of the following chunk
    };
    if (r < R) fetch(r, 0);
    float Tc = 1.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f, dep = 0.f, a = 0.f, nrm = 0.f;
    while (r < R) {
        float4 q = q_n;
        float zv = z_n, zx = zx_n;
        const int s = ch * 64 + lane;
        const bool ok = s < S;
        if (ch == 0) {
            float d0 = rays_d[r * 3 + 0], d1 = rays_d[r * 3 + 1], d2 = rays_d[r * 3 + 2];
            nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
            Tc = 1.0f; c0 = c1 = c2 = dep = a = 0.f;
        }
        // issue the next (ray, chunk) loads now
        int64_t rn = r; int cn = ch + 1;
        if (cn == nch) { cn = 0; rn = r + nwaves; }
        fetch(rn, cn);
        // next sample's depth: wave_shl:1 on the DPP path (lane 63 keeps `old` = the first depth of the following chunk)
        float zn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, zx), __builtin_bit_cast(int, zv), 0x130, 0xf, 0xf, false));
        float dist = (s + 1 < S) ? (zn - zv) : 1e10f;
        dist = dist * nrm;
        float sigma = q.w > 0.f ? q.w : 0.f;
        float alpha = ok ? 1.0f - __builtin_amdgcn_exp2f(-1.4426950408889634f * sigma * dist) : 0.f;
        float t = ok ? (1.0f - alpha) + 1e-10f : 1.0f;
        // inclusive prefix product over the 64 lanes on the DPP path (no LDS crossbar): Hillis-Steele inside the 16-lane
        // rows (row_shr 1, 2, 4, 8; lanes without a source multiply by `old` = 1), then row_bcast 15 / 31 across rows
        float inc = t;
        const int one = 0x3f800000;
#define CTX_SCAN_STEP(ctrl, rmask) inc = inc * __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(one, __builtin_bit_cast(int, inc), ctrl, rmask, 0xf, false))
        CTX_SCAN_STEP(0x111, 0xf);
        CTX_SCAN_STEP(0x112, 0xf);
        CTX_SCAN_STEP(0x114, 0xf);
        CTX_SCAN_STEP(0x118, 0xf);
        CTX_SCAN_STEP(0x142, 0xa);                    // row_bcast:15 into rows 1 and 3
        CTX_SCAN_STEP(0x143, 0xc);                    // row_bcast:31 into rows 2 and 3
#undef CTX_SCAN_STEP
        // exclusive = inclusive shifted right by one lane (wave_shr:1; lane 0 keeps `old` = 1)
        float exc = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(one, __builtin_bit_cast(int, inc), 0x138, 0xf, 0xf, false));
        float w = alpha * (Tc * exc);
        if (weights && ok) weights[r * S + s] = w;
        c0 += w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * q.x));
        c1 += w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * q.y));
        c2 += w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * q.z));
        dep += w * zv;
        a += w;
        Tc = Tc * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inc), 63));
        if (ch + 1 == nch) {
            float s0 = wave_sum_dpp(c0), s1 = wave_sum_dpp(c1), s2 = wave_sum_dpp(c2), sd = wave_sum_dpp(dep), sa = wave_sum_dpp(a);
            if (lane == 0) {
                if (white) { s0 += 1.0f - sa; s1 += 1.0f - sa; s2 += 1.0f - sa; }
                rgb[r * 3 + 0] = s0; rgb[r * 3 + 1] = s1; rgb[r * 3 + 2] = s2;
                depth[r] = sd; acc[r] = sa;
                float qd = sd / sa;
                float dv = 1.0f / (qd > 1e-10f ? qd : 1e-10f);
                disp[r] = (qd != qd) ? qd : dv;
            }
        }
        r = rn; ch = cn;
    }
}

extern "C" int32_t ctx_raymarch_composite_fwd(const float *raw, const float *z_vals, const float *rays_d, int64_t R,
                                              int32_t S, int32_t white_bkgd, float *rgb, float *disp, float *acc,
                                              float *weights, float *depth, ctx_stream_t stream)
{
    CTX_REQUIRE(raw && z_vals && rays_d && rgb && disp && acc && depth && R > 0 && S > 0, "raymarch: bad args");
    int64_t nb = cdiv64(R, 4);
    static int cap = -1;
    if (cap < 0) { const char *e = getenv("CTX_COMPOSITE_BLOCKS"); cap = e ? atoi(e) : 262144; }   // one ray per wave up to 1 M rays: 136.8 us vs 142.6 at 32768 blocks (512^2 x 128)
    if (nb > cap) nb = cap;
    hipLaunchKernelGGL(k_composite, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float4 *)raw, z_vals,
                       rays_d, R, S, white_bkgd, rgb, disp, acc, weights, depth);
    CTX_CHECK_LAUNCH("raymarch_composite");
    return CTX_OK;
}

"""Drop-in for the slice of `kaolin` (0.15.0) the reference calls — same attribute paths, argument
order and return layouts, backed by libctxnerf.so on gfx950:

    kal.render.camera.generate_perspective_projection   src/models/render.py:11
    kal.render.camera.generate_transformation_matrix    src/models/render.py:31,45
    kal.render.mesh.prepare_vertices                    src/models/render.py:112; textured_mesh.py:167
    kal.render.mesh.rasterize                           src/models/render.py:115,119; textured_mesh.py:170,174
    kal.render.mesh.texture_mapping                     src/models/render.py:135
    kal.ops.mesh.index_vertices_by_faces                src/models/textured_mesh.py:149
    kal.io.obj.import_mesh                              src/models/mesh.py:12-14

plus `kal.render.mesh.rasterize_fused`, the single-pass depth+uv+index(+normals) raster the renderer
uses instead of the reference's two passes.  Camera helpers are host-side torch (tiny).
"""
import math
import types
import torch
from . import _lib as L


# ---- camera ------------------------------------------------------------------------------------
def generate_perspective_projection(fovyangle, ratio=1.0, dtype=torch.float):
    tanfov = math.tan(fovyangle / 2.0)
    return torch.tensor([[1.0 / (ratio * tanfov)], [1.0 / tanfov], [-1]], dtype=dtype)


def generate_transformation_matrix(camera_position, look_at, camera_up_direction):
    z_axis = camera_position - look_at
    z_axis = z_axis / z_axis.norm(dim=1, keepdim=True)
    x_axis = torch.cross(camera_up_direction, z_axis, dim=1)
    x_axis = x_axis / x_axis.norm(dim=1, keepdim=True)
    y_axis = torch.cross(z_axis, x_axis, dim=1)
    rot_part = torch.stack([x_axis, y_axis, z_axis], dim=2)
    trans_part = -camera_position.unsqueeze(1) @ rot_part
    return torch.cat([rot_part, trans_part], dim=1)


# ---- mesh ops ----------------------------------------------------------------------------------
def index_vertices_by_faces(vertices_features, faces):
    return vertices_features[:, faces.long()]


def prepare_vertices(vertices, faces, camera_proj, camera_rot=None, camera_trans=None, camera_transform=None):
    if camera_transform is None:
        raise L.CtxError("prepare_vertices: only the camera_transform form used by the reference is implemented")
    lib = L.load()
    dev = vertices.device
    verts = L.f32c(vertices)
    faces = faces.to(device=dev, dtype=torch.int64).contiguous()
    cam = L.f32c(camera_transform, dev)
    proj = L.f32c(camera_proj.reshape(3), dev)
    B, V, _ = verts.shape
    F = faces.shape[0]
    fv_cam = torch.empty(B, F, 3, 3, device=dev)
    fv_img = torch.empty(B, F, 3, 2, device=dev)
    fnorm = torch.empty(B, F, 3, device=dev)
    ws = torch.empty(B * V * 5, device=dev)
    L.check(lib.ctx_prepare_vertices(L.ptr(verts, torch.float32, "vertices"), L.ptr(faces, torch.int64, "faces"),
                                     L.ptr(cam), L.ptr(proj), B, V, F, L.ptr(fv_cam), L.ptr(fv_img), L.ptr(fnorm),
                                     L.ptr(ws), L.stream()))
    return fv_cam, fv_img, fnorm


def _raster_ws(H, W, B, F, dev):
    n = L.load().ctx_rasterize_ws_bytes(H, W, B, F)
    return torch.empty(n, dtype=torch.uint8, device=dev), n


def rasterize(height, width, face_vertices_z, face_vertices_image, face_features, valid_faces=None,
              multiplier=None, eps=None, backend='cuda'):
    """-> (interpolated_features [B,H,W,C] f32, face_idx [B,H,W] i64, -1 = background)."""
    if valid_faces is not None:
        raise L.CtxError("rasterize: valid_faces is not used by the reference and is not implemented")
    multiplier = 1000.0 if multiplier is None else float(multiplier)
    eps = 1e-8 if eps is None else float(eps)
    lib = L.load()
    dev = face_vertices_z.device
    fz, fxy, feat = L.f32c(face_vertices_z), L.f32c(face_vertices_image), L.f32c(face_features)
    B, F, _ = fz.shape
    C = feat.shape[-1]
    out = torch.empty(B, height, width, C, device=dev)
    idx = torch.empty(B, height, width, dtype=torch.int64, device=dev)
    ws, n = _raster_ws(height, width, B, F, dev)
    L.check(lib.ctx_rasterize_fwd(height, width, L.ptr(fz, torch.float32, "face_vertices_z"), L.ptr(fxy), L.ptr(feat),
                                  B, F, C, multiplier, eps, L.ptr(out), L.ptr(idx), L.ptr(ws), n, L.stream()))
    return out, idx


def rasterize_fused(height, width, face_vertices_camera, face_vertices_image, uv_face_attr, face_normals=None,
                    multiplier=1000.0, eps=1e-8):
    """One pass for what the reference does in two (render.py:115-120):
    -> depth [B,H,W,1], uv [B,H,W,2], face_idx [B,H,W] i64, normals [B,H,W,3] | None."""
    lib = L.load()
    dev = face_vertices_camera.device
    fvc, fxy, uva = L.f32c(face_vertices_camera), L.f32c(face_vertices_image), L.f32c(uv_face_attr)
    B, F = fvc.shape[0], fvc.shape[1]
    Bu = uva.shape[0]
    depth = torch.empty(B, height, width, 1, device=dev)
    uv = torch.empty(B, height, width, 2, device=dev)
    idx = torch.empty(B, height, width, dtype=torch.int64, device=dev)
    fn = L.f32c(face_normals) if face_normals is not None else None
    normals = torch.empty(B, height, width, 3, device=dev) if fn is not None else None
    ws, n = _raster_ws(height, width, B, F, dev)
    L.check(lib.ctx_rasterize_fused(height, width, L.ptr(fvc), L.ptr(fxy), L.ptr(uva), Bu, L.ptr(fn), B, F,
                                    float(multiplier), float(eps), L.ptr(depth), L.ptr(uv), L.ptr(idx), L.ptr(normals),
                                    L.ptr(ws), n, L.stream()))
    return depth, uv, idx, normals


# ---- UV scatter: plans ------------------------------------------------------------------------------
# A plan (binning of a raster by atlas tile, uvscatter.hip) depends on (uv, mask) only.  Plans are cached ONLY when the caller
# says the raster will come back (`reuse=True`: the SDS loop's render_cache); one-shot scatters build theirs and drop it.  The
# cache is keyed by the tensors' addresses and versions, holds the tensors alive, is capped in bytes, and every scatter re-checks
# a sampled checksum of (uv, mask) on the device: a raster rewritten behind the key poisons the result with NaN instead of
# silently scattering along the old lists.
_PLANS = {}
PLAN_CACHE_BYTES = 768 << 20


def clear_scatter_plans():
    """Drop every cached scatter plan (and the rasters they keep alive)."""
    _PLANS.clear()


def _plan_bytes(ent):
    return sum(t.numel() * t.element_size() for t in ent if t is not None)


def binned_fits(C, T):
    """True when the tile-binned scatter can take this atlas: C <= 4 and T within the plan's LDS histogram (2272)."""
    return C <= 4 and T <= L.load().ctx_texmap_plan_max_res()


def scatter_plan(uv, mask_idx, T, reuse=False):
    """-> plan buffer for (uv [B,..,2] f32 contiguous, mask_idx | None, T)."""
    lib = L.load()
    B = uv.shape[0]
    HW = uv[0].numel() // 2
    key = (uv.data_ptr(), uv._version, None if mask_idx is None else (mask_idx.data_ptr(), mask_idx._version), B, HW, T)
    ent = _PLANS.get(key) if reuse else None
    if ent is not None:
        return ent[0]
    nbytes = lib.ctx_texmap_bwd_plan_bytes(B, HW, T)
    if nbytes < 0:
        raise L.CtxError(f"scatter_plan: B*HW = {B * HW} pixels do not fit the binned path")
    plan = torch.empty(nbytes, dtype=torch.uint8, device=uv.device)
    L.check(lib.ctx_texmap_bwd_plan(L.ptr(uv, torch.float32, "uv"), L.ptr(mask_idx), B, HW, T, L.ptr(plan), L.stream()))
    if reuse:
        ent = (plan, uv, mask_idx)                    # keep uv / mask alive: the key is their address
        if _plan_bytes(ent) <= PLAN_CACHE_BYTES:
            while _PLANS and sum(_plan_bytes(e) for e in _PLANS.values()) + _plan_bytes(ent) > PLAN_CACHE_BYTES:
                _PLANS.pop(next(iter(_PLANS)))
            _PLANS[key] = ent
    return plan


def scatter_add_texture(go, uv, mask_idx, grad_tex, binned=None, reuse=False):
    """grad_tex [C,T,T] += bilinear scatter of go [B,H,W,C] (or [B,HW,C]) at uv — the backward of texture_mapping.  Large rasters
    (>= 64k pixels, C <= 4, T within the plan's limit) go through the binned, atomics-free path of uvscatter.hip, whose fixed-point
    unit follows max|go| of the call; otherwise the float-atomics kernel.  reuse: keep the plan for the next call on this raster."""
    lib = L.load()
    B = uv.shape[0]
    HW = uv[0].numel() // 2
    C, T = grad_tex.shape[0], grad_tex.shape[-1]
    use = (B * HW >= 65536 and binned_fits(C, T)) if binned is None else binned
    if not use:
        L.check(lib.ctx_texture_mapping_bwd(L.ptr(go, torch.float32, "grad_out"), L.ptr(uv, torch.float32, "uv"), B, HW, C, T,
                                            L.ptr(mask_idx), L.ptr(grad_tex, torch.float32, "grad_tex"), L.stream()))
        return grad_tex
    plan = scatter_plan(uv, mask_idx, T, reuse=reuse)
    ws = torch.empty(lib.ctx_texture_mapping_bwd_binned_ws_bytes(C, T), dtype=torch.uint8, device=uv.device)
    L.check(lib.ctx_texture_mapping_bwd_binned(L.ptr(go, torch.float32, "grad_out"), L.ptr(uv, torch.float32, "uv"), L.ptr(mask_idx), B, HW, C, T,
                                               L.ptr(plan), L.ptr(ws), L.ptr(grad_tex, torch.float32, "grad_tex"), L.stream()))
    return grad_tex


SCATTER_FRAC_BITS = 32          # unit of the painted-view accumulators: 2^-32 (values are colours x weights in [0, 1])


def scatter_fixed(values, uv, mask_idx, acc, frac_bits=SCATTER_FRAC_BITS, reuse=False):
    """acc [C,T,T] int64 += round(values * bilinear weight * 2^frac_bits): the UV back-projection of painted views as integer
    sums (order-free, so view shards on different ranks add up to the same bits).  Any raster size; atlases beyond the plan's
    limit (or C > 4) take the plan-less kernel (one int64 atomic per tap)."""
    lib = L.load()
    B = uv.shape[0]
    HW = uv[0].numel() // 2
    C, T = acc.shape[0], acc.shape[-1]
    plan = scatter_plan(uv, mask_idx, T, reuse=reuse) if binned_fits(C, T) else None
    L.check(lib.ctx_uv_scatter_fixed(L.ptr(values, torch.float32, "values"), L.ptr(uv, torch.float32, "uv"), L.ptr(mask_idx), B, HW, C, T,
                                     L.ptr(plan), int(frac_bits), L.ptr(acc, torch.int64, "acc"), L.stream()))
    return acc


def fixed_to_float(acc, frac_bits=SCATTER_FRAC_BITS, out=None):
    """int64 sums in units of 2^-frac_bits -> float32 (one rounding per texel)."""
    lib = L.load()
    if out is None:
        out = torch.empty(acc.shape, dtype=torch.float32, device=acc.device)
    L.check(lib.ctx_fixed_to_float(L.ptr(acc, torch.int64, "acc"), acc.numel(), int(frac_bits), 0, L.ptr(out, torch.float32, "out"), L.stream()))
    return out


class _TextureMapping(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, tex, mode, mask_idx):
        lib = L.load()
        B, H, W, _ = uv.shape
        Bt, Cc, T, T2 = tex.shape
        if T != T2:
            raise L.CtxError("texture_mapping: square texture expected")
        # an expanded (stride-0) batch is the reference's texture_img.expand(B,...) — keep one copy
        if Bt > 1 and tex.stride(0) == 0:
            tex_c, Bt_eff = tex[:1].contiguous(), 1
        else:
            tex_c, Bt_eff = tex.contiguous(), Bt
        uvc = L.f32c(uv)
        out = torch.empty(B, H, W, Cc, device=uv.device)
        m = {'bilinear': 0, 'nearest': 1}[mode]
        if Bt_eff == 1 and Cc <= 4 and B * H * W >= 4 * T * T:
            # one shared texture sampled by many more pixels than it has texels: interleave the channels once (16 B per
            # texel), then every bilinear tap is one gather (bit-identical results)
            packed = torch.empty(T, T, 4, device=uv.device)
            L.check(lib.ctx_texture_pack4(L.ptr(tex_c.float() if tex_c.dtype != torch.float32 else tex_c, torch.float32, "texture_maps"),
                                          Cc, T, L.ptr(packed), L.stream()))
            L.check(lib.ctx_texture_mapping_packed_fwd(L.ptr(uvc, torch.float32, "texture_coordinates"), L.ptr(packed), B, H * W, Cc, T, m,
                                                       L.ptr(mask_idx), L.ptr(out), L.stream()))
        else:
            L.check(lib.ctx_texture_mapping_fwd(L.ptr(uvc, torch.float32, "texture_coordinates"),
                                                L.ptr(tex_c, torch.float32, "texture_maps"), B, H * W, Cc, T, Bt_eff,
                                                m, L.ptr(mask_idx), L.ptr(out), L.stream()))
        ctx.save_for_backward(uvc, mask_idx if mask_idx is not None else torch.empty(0))
        ctx.meta = (B, H * W, Cc, T, Bt, Bt_eff, mode, tex.shape, mask_idx is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        uvc, mask_idx = ctx.saved_tensors
        B, HW, Cc, T, Bt, Bt_eff, mode, tshape, has_mask = ctx.meta
        if mode != 'bilinear':
            return None, torch.zeros(tshape, device=grad_out.device), None, None
        lib = L.load()
        go = L.f32c(grad_out)
        if Bt_eff == 1:
            g = torch.zeros(Cc, T, T, device=go.device)
            scatter_add_texture(go, uvc, mask_idx if has_mask else None, g, reuse=True)
            if Bt > 1:      # input was an expand(): autograd sums the batch slices, so put the whole sum in slice 0
                full = torch.zeros(tshape, device=go.device)
                full[0] = g
                g = full
            else:
                g = g[None]
        else:
            g = torch.zeros(Bt, Cc, T, T, device=go.device)
            for b in range(Bt):
                mi = mask_idx[b:b + 1].contiguous() if has_mask else None
                L.check(lib.ctx_texture_mapping_bwd(L.ptr(go[b:b + 1].contiguous()), L.ptr(uvc[b:b + 1].contiguous()), 1, HW,
                                                    Cc, T, L.ptr(mi), L.ptr(g[b]), L.stream()))
        return None, g, None, None


def texture_mapping(texture_coordinates, texture_maps, mode='bilinear', mask_idx=None):
    """grid_sample(tex, (u, 1-v)*2-1, align_corners=False, padding_mode='border') -> [B,H,W,C].
    Differentiable w.r.t. texture_maps (bilinear).  mask_idx: optional face_idx to fuse `* mask`."""
    if mode not in ('bilinear', 'nearest'):
        raise L.CtxError(f"texture_mapping: mode {mode!r} not implemented on the HIP path (bilinear/nearest)")
    if texture_coordinates.dim() == 3:
        return _TextureMapping.apply(texture_coordinates[:, :, None, :], texture_maps, mode, mask_idx)[:, :, 0]
    return _TextureMapping.apply(texture_coordinates, texture_maps, mode, mask_idx)


# ---- OBJ import (host, Python like kaolin's) ------------------------------------------------------
def import_mesh(path, with_normals=False, with_materials=False, heterogeneous_mesh_handler=None):
    vs, vts, fs, fts = [], [], [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith('v '):
                vs.append([float(x) for x in line.split()[1:4]])
            elif line.startswith('vt '):
                vts.append([float(x) for x in line.split()[1:3]])
            elif line.startswith('f '):
                vi, ti = [], []
                for tok in line.split()[1:]:
                    p = tok.split('/')
                    vi.append(int(p[0]))
                    ti.append(int(p[1]) if len(p) > 1 and p[1] else 0)
                nv, nt = len(vs), len(vts)
                vi = [i - 1 if i > 0 else nv + i for i in vi]
                ti = [i - 1 if i > 0 else (nt + i if i < 0 else -1) for i in ti]
                for k in range(1, len(vi) - 1):          # naive homogenize: fan triangulation
                    fs.append([vi[0], vi[k], vi[k + 1]])
                    fts.append([ti[0], ti[k], ti[k + 1]])
    return types.SimpleNamespace(
        vertices=torch.tensor(vs, dtype=torch.float32), faces=torch.tensor(fs, dtype=torch.int64),
        uvs=torch.tensor(vts, dtype=torch.float32).reshape(-1, 2),
        face_uvs_idx=torch.tensor(fts, dtype=torch.int64) if fts else torch.zeros(0, 3, dtype=torch.int64))


render = types.SimpleNamespace(
    camera=types.SimpleNamespace(generate_perspective_projection=generate_perspective_projection,
                                 generate_transformation_matrix=generate_transformation_matrix),
    mesh=types.SimpleNamespace(prepare_vertices=prepare_vertices, rasterize=rasterize, rasterize_fused=rasterize_fused,
                               texture_mapping=texture_mapping))
ops = types.SimpleNamespace(mesh=types.SimpleNamespace(index_vertices_by_faces=index_vertices_by_faces))
io = types.SimpleNamespace(obj=types.SimpleNamespace(import_mesh=import_mesh),
                           utils=types.SimpleNamespace(heterogeneous_mesh_handler_naive_homogenize=None))

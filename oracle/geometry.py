"""ORACLE (test infrastructure) — numpy front-end of oracle/geometry_ref.c plus the host-side
camera maths.  See geometry_ref.c for the reference file:line each routine restates."""
import ctypes
import numpy as np
from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# -- camera (src/models/render.py:8-46 -> kaolin camera helpers) ----------
def generate_perspective_projection(fovy, ratio=1.0):
    t = np.tan(fovy / 2.0)
    return np.array([[1.0 / (ratio * t)], [1.0 / t], [-1.0]], dtype=np.float32)


def generate_transformation_matrix(pos, look_at, up):
    pos, look_at, up = _f32(pos), _f32(look_at), _f32(up)
    z = pos - look_at
    z = z / np.linalg.norm(z, axis=1, keepdims=True)
    x = np.cross(up, z)
    x = x / np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    rot = np.stack([x, y, z], axis=2)                       # [B,3,3]
    trans = -(pos[:, None, :] @ rot)                        # [B,1,3]
    return np.concatenate([rot, trans], axis=1).astype(np.float32)   # [B,4,3]


def get_camera_from_multiple_view(elev, azim, r, look_at_height=0.0):
    elev, azim, r = _f32(elev), _f32(azim), _f32(r)
    x = r * np.sin(elev) * np.sin(azim)
    y = r * np.cos(elev)
    z = r * np.sin(elev) * np.cos(azim)
    pos = np.stack([x, y, z], axis=1).astype(np.float32)
    look = np.zeros_like(pos); look[:, 1] = look_at_height
    up = np.zeros_like(pos); up[:, 1] = 1.0
    return generate_transformation_matrix(pos, look, up)


# -- kaolin-seam restatements ------------------------------------------------------------------
def prepare_vertices(verts, faces, proj, cam):
    verts, faces, cam = _f32(verts), _i64(faces), _f32(cam)
    proj3 = _f32(np.asarray(proj).reshape(3))
    B, V, _ = verts.shape
    F = faces.shape[0]
    fv_cam = np.empty((B, F, 3, 3), np.float32)
    fv_img = np.empty((B, F, 3, 2), np.float32)
    fn = np.empty((B, F, 3), np.float32)
    lib().orc_prepare_vertices(_p(verts), _p(faces), _p(cam), _p(proj3), B, V, F, _p(fv_cam), _p(fv_img), _p(fn))
    return fv_cam, fv_img, fn


def rasterize(H, W, face_z, face_xy, feat, multiplier=1000.0, eps=1e-8, brute=None):
    """brute=True: the literal pixel-parallel loop over every face; False: the face-major walk of
    bounding boxes (identical results, see geometry_ref.c); None: brute below 2^27 pixel-face pairs."""
    face_z, face_xy, feat = _f32(face_z), _f32(face_xy), _f32(feat)
    B, F, _ = face_z.shape
    C = feat.shape[-1]
    out = np.empty((B, H, W, C), np.float32)
    idx = np.empty((B, H, W), np.int64)
    if brute is None:
        brute = B * H * W * F <= (1 << 27)
    fn = lib().orc_rasterize if brute else lib().orc_rasterize_bbox
    fn(H, W, _p(face_z), _p(face_xy), _p(feat), B, F, C,
                        ctypes.c_float(multiplier), ctypes.c_float(eps), _p(out), _p(idx))
    return out, idx


def normalize_multiple_depth(depth):
    depth = _f32(depth)
    B = depth.shape[0]
    out = np.empty_like(depth)
    rc = lib().orc_normalize_depth(_p(depth), B, depth[0].size, _p(out))
    if rc == 1:
        raise AssertionError('depth map should be negative')
    if rc == 2:
        raise AssertionError('depth map should not be empty')
    return out


def texture_mapping(uv, tex, mode='bilinear'):
    uv, tex = _f32(uv), _f32(tex)
    B, H, W, _ = uv.shape
    Bt, C, T, T2 = tex.shape
    assert T == T2 and Bt in (1, B)
    out = np.empty((B, H, W, C), np.float32)
    lib().orc_texture_mapping(_p(uv), _p(tex), B, H * W, C, T, Bt, {'bilinear': 0, 'nearest': 1}[mode], _p(out))
    return out


def texture_mapping_bwd(grad_out, uv, T):
    grad_out, uv = _f32(grad_out), _f32(uv)
    B, H, W, C = grad_out.shape
    g = np.empty((C, T, T), np.float32)
    lib().orc_texture_mapping_bwd(_p(grad_out), _p(uv), B, H * W, C, T, _p(g))
    return g


def uv_scatter_fixed(values, uv, mask_idx, T, frac_bits=32, acc=None):
    """-> acc [C,T,T] int64 (+= when given): integer restatement of the UV back-projection scatter."""
    values, uv = _f32(values), _f32(uv)
    B = uv.shape[0]
    HW = uv[0].size // 2
    C = values.shape[-1]
    if acc is None:
        acc = np.zeros((C, T, T), np.int64)
    mi = _i64(mask_idx) if mask_idx is not None else None
    lib().orc_uv_scatter_fixed(_p(values), _p(uv), _p(mi) if mi is not None else None, B, HW, C, T, int(frac_bits), _p(acc))
    return acc


def gather_normals(face_idx, fnorm):
    face_idx, fnorm = _i64(face_idx), _f32(fnorm)
    B, F, _ = fnorm.shape
    out = np.empty(face_idx.shape + (3,), np.float32)
    lib().orc_gather_normals(_p(face_idx), _p(fnorm), B, face_idx[0].size, F, _p(out))
    return out


def view_weights(face_idx, fnz):
    """face_idx [B,H,W] (or [B,1,H,W]) i64, fnz [B,F] f32 -> (max_z [F], mask bool same shape)."""
    shp = face_idx.shape
    face_idx, fnz = _i64(face_idx), _f32(fnz)
    B, F = fnz.shape
    mz = np.empty(F, np.float32)
    mask = np.empty(face_idx.size, np.uint8)
    lib().orc_view_weights(_p(face_idx), _p(fnz), B, face_idx.size // B, F, _p(mz), _p(mask))
    return mz, mask.reshape(shp).astype(bool)


def raw2outputs(raw, z_vals, rays_d, white_bkgd=False):
    raw, z_vals, rays_d = _f32(raw), _f32(z_vals), _f32(rays_d)
    R, S, _ = raw.shape
    rgb = np.empty((R, 3), np.float32); disp = np.empty(R, np.float32); acc = np.empty(R, np.float32)
    w = np.empty((R, S), np.float32); depth = np.empty(R, np.float32)
    lib().orc_raw2outputs(_p(raw), _p(z_vals), _p(rays_d), R, S, int(white_bkgd), _p(rgb), _p(disp), _p(acc), _p(w), _p(depth))
    return rgb, disp, acc, w, depth


# -- mesh helpers (src/models/mesh.py:27-65) ---------------------------------------------------
def calculate_face_normals(v, f):
    v = np.asarray(v, np.float32)
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(v1 - v0, v2 - v0)
    ta = np.linalg.norm(n, axis=-1)
    return (n / ta[:, None]).astype(np.float32), (ta / 2).astype(np.float32)


def normalize_mesh(v, target_scale=1.0, dy=0.0):
    v = np.asarray(v, np.float32)
    v = v - v.mean(axis=0, dtype=np.float32)
    scale = np.max(np.linalg.norm(v, axis=1))
    v = v / scale
    v = v * np.float32(target_scale)
    v[:, 1] += np.float32(dy)
    return v.astype(np.float32)


def load_obj(path):
    """Minimal OBJ reader (v / vt / f with v, v/vt, v//vn, v/vt/vn; polygons fan-triangulated).
    Stands in for kal.io.obj.import_mesh (src/models/mesh.py:12-17)."""
    vs, vts, fs, fts = [], [], [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith('v '):
                vs.append([float(x) for x in line.split()[1:4]])
            elif line.startswith('vt '):
                vts.append([float(x) for x in line.split()[1:3]])
            elif line.startswith('f '):
                toks = line.split()[1:]
                vi, ti = [], []
                for t in toks:
                    p = t.split('/')
                    vi.append(int(p[0]))
                    ti.append(int(p[1]) if len(p) > 1 and p[1] else 0)
                nv, nt = len(vs), len(vts)
                vi = [i - 1 if i > 0 else nv + i for i in vi]
                ti = [i - 1 if i > 0 else (nt + i if i < 0 else -1) for i in ti]
                for k in range(1, len(vi) - 1):
                    fs.append([vi[0], vi[k], vi[k + 1]])
                    fts.append([ti[0], ti[k], ti[k + 1]])
    return (np.array(vs, np.float32), np.array(fs, np.int64),
            np.array(vts, np.float32).reshape(-1, 2), np.array(fts, np.int64))

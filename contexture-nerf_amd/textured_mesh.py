"""TexturedMeshModel: mirror of the hot-path members of src/models/textured_mesh.py
(`__init__` :85-156, `render_face_normals_face_idx` :158-194, `get_texture_map` :266-301,
`init_texture_map` :371-409, `render` :476-580) on the HIP raster / texture-field kernels.

Out of scope here: spectral/axis augmentations, Laplacian helpers, mesh export.
UV atlases: meshes that carry UVs use them; for meshes without UVs the reference calls xatlas (C++,
absent offline) — `atlas.chart_atlas` (connected charts grown over face adjacency, planar projection, skyline
packing with a gutter) is the deterministic stand-in; its layout cannot match xatlas and is documented as such
in DESIGN.md.  `grid_atlas` below (two triangles per grid cell, a seam on every edge) remains as the fallback
selected by CTX_ATLAS=grid.
"""
import json
import math
import os
import numpy as np
import torch
from . import kal
from .mesh import Mesh
from .render import Renderer


def grid_atlas(n_faces, margin=0.08):
    """Deterministic per-face UV chart: cell (i,j) of an n x n grid holds faces 2k (lower-left triangle) and
    2k+1 (upper-right triangle), shrunk by `margin` of a cell.  -> vt [3F,2] f32, ft [F,3] i64."""
    n = max(1, math.ceil(math.sqrt((n_faces + 1) // 2)))
    k = np.arange(n_faces)
    cell = k // 2
    cx = (cell % n).astype(np.float32); cy = (cell // n).astype(np.float32)
    m = margin
    lower = np.array([[m, m], [1 - 2 * m, m], [m, 1 - 2 * m]], np.float32)
    upper = np.array([[1 - m, 1 - m], [2 * m, 1 - m], [1 - m, 2 * m]], np.float32)
    tri = np.where((k % 2 == 0)[:, None, None], lower[None], upper[None])          # [F,3,2]
    vt = (tri + np.stack([cx, cy], -1)[:, None, :]) / n
    ft = np.arange(3 * n_faces, dtype=np.int64).reshape(n_faces, 3)
    return torch.from_numpy(vt.reshape(-1, 2).astype(np.float32)), torch.from_numpy(ft)


_ATLAS_MEMO = {}


def _chart_atlas_memo(v_np, f_np, resolution):
    """atlas.chart_atlas with an in-process memo keyed by the mesh bytes (a batch that repeats a mesh, or several trainers over
    one mesh, unwrap it once; the generator is deterministic)."""
    import hashlib
    from .atlas import chart_atlas
    key = (hashlib.sha1(np.ascontiguousarray(v_np).tobytes() + np.ascontiguousarray(f_np).tobytes()).hexdigest(), int(resolution))
    if key not in _ATLAS_MEMO:
        vt_np, ft_np = chart_atlas(v_np, f_np, resolution=int(resolution))
        _ATLAS_MEMO[key] = (torch.from_numpy(vt_np), torch.from_numpy(ft_np))
    vt, ft = _ATLAS_MEMO[key]
    return vt.clone(), ft.clone()


class TexturedMeshModel(torch.nn.Module):
    def __init__(self, opt, render_grid_size=1024, texture_resolution=1024, initial_texture_path=None, cache_path=None,
                 device=torch.device('cuda'), augmentations=False, augment_prob=0.5, fovyangle=np.pi / 3,
                 texture_mlp=None, uv_embedder=None, mesh_arrays=None):
        super().__init__()
        self.device = device
        self.opt = opt
        self.augmentations = False          # disabled in the reference too (trainer.py:265)
        self.dy = self.opt.dy
        self.mesh_scale = self.opt.shape_scale
        self.texture_resolution = texture_resolution
        self.cache_path = cache_path
        self.num_features = 3
        self.dim = (render_grid_size, render_grid_size)
        self.renderer = Renderer(device=self.device, dim=self.dim, interpolation_mode=self.opt.texture_interpolation_mode,
                                 fovyangle=fovyangle)
        self.mesh = self.init_meshes(mesh_arrays)
        self.texture_mlp = texture_mlp
        self.uv_embedder = uv_embedder
        self.vt, self.ft = self.init_texture_map()
        self.face_attributes = kal.ops.mesh.index_vertices_by_faces(self.vt.unsqueeze(0), self.ft.long()).detach()

    def init_meshes(self, mesh_arrays=None):
        mesh = Mesh(self.opt.shape_path, self.device, arrays=mesh_arrays)
        return mesh.normalize_mesh(inplace=True, target_scale=self.mesh_scale, dy=self.dy)

    def init_texture_map(self):
        cache_path = self.cache_path
        if cache_path is not None:
            vt_cache, ft_cache = os.path.join(str(cache_path), 'vt.pth'), os.path.join(str(cache_path), 'ft.pth')
        # the reference takes the file's UVs whenever ft.min() > -1 (textured_mesh.py:380-381); shapes/sphere.obj carries 960 vt
        # lines that are ALL (0, 0) — every face would land on one texel — so a chart of zero extent also falls through to the atlas
        if self.mesh.vt is not None and self.mesh.ft is not None and self.mesh.vt.shape[0] > 0 and self.mesh.ft.numel() > 0 \
                and self.mesh.ft.min() > -1 and float((self.mesh.vt.max(0).values - self.mesh.vt.min(0).values).min()) > 0:
            vt, ft = self.mesh.vt.to(self.device), self.mesh.ft.to(self.device)
        elif cache_path is not None and os.path.exists(vt_cache) and os.path.exists(ft_cache) and self._cache_meta_ok(cache_path):
            vt = torch.load(vt_cache, weights_only=True).to(self.device)
            ft = torch.load(ft_cache, weights_only=True).to(self.device)
        else:
            if os.environ.get("CTX_ATLAS", "chart") == "grid":
                vt, ft = grid_atlas(self.mesh.faces.shape[0])
            else:                                                   # the xatlas stand-in (textured_mesh.py:392-404)
                vt, ft = _chart_atlas_memo(self.mesh.vertices.detach().cpu().numpy(), self.mesh.faces.cpu().numpy(), self.texture_resolution)
            vt, ft = vt.to(self.device), ft.to(self.device)
            if cache_path is not None:
                os.makedirs(str(cache_path), exist_ok=True)
                # every rank of a job unwraps the same (deterministic) atlas: write through temporaries and rename, so that a
                # peer never reads a half-written file
                suffix = f'.tmp{os.getpid()}'
                torch.save(vt.cpu(), vt_cache + suffix); os.replace(vt_cache + suffix, vt_cache)
                torch.save(ft.cpu(), ft_cache + suffix); os.replace(ft_cache + suffix, ft_cache)
                meta = os.path.join(str(cache_path), 'atlas_meta.json')
                with open(meta + suffix, 'w') as fh:
                    json.dump(self._cache_meta(), fh)
                os.replace(meta + suffix, meta)
        return vt, ft

    def _cache_meta(self):
        return {"generator": os.environ.get("CTX_ATLAS", "chart"), "resolution": int(self.texture_resolution),
                "faces": int(self.mesh.faces.shape[0])}

    def _cache_meta_ok(self, cache_path):
        """A cached atlas is reused when it was generated for this face count, generator and atlas resolution (the gutter is in
        texels).  A cache without the side file (written by another tool, as the reference's xatlas cache) is taken as is."""
        f = os.path.join(str(cache_path), 'atlas_meta.json')
        if not os.path.exists(f):
            return True
        try:
            return json.load(open(f)) == self._cache_meta()
        except Exception:
            return False

    def get_texture_map(self):
        """-> (texture [1,3,res,res] in [0,1], mlp_output [res*res,3]); uv grid, embedding, MLP and (tanh+1)/2 fused."""
        return self.texture_mlp.texture_map(self.texture_resolution)

    def export_mesh(self, path, texture=None):
        """src/models/textured_mesh.py:418-474: albedo.png + mesh.obj + mesh.mtl.  texture (optional [1,3,T,T] in [0,1]): the
        painted atlas (ConTEXTure.paint's merged UV scatter) instead of the texture field's current output."""
        from .mesh import write_textured_obj
        with torch.no_grad():
            tex = self.get_texture_map()[0] if texture is None else texture
            colors = (tex.permute(0, 2, 3, 1).contiguous().clamp(0, 1)[0] * 255).to(torch.uint8).cpu().numpy()
        write_textured_obj(path, self.mesh.vertices.detach().cpu().numpy(), self.mesh.faces.cpu().numpy(),
                           self.vt.detach().cpu().numpy(), self.ft.detach().cpu().numpy(), colors)

    def _angles(self, v):
        if v is None:
            return None
        if isinstance(v, (float, int)):
            return torch.tensor([v], dtype=torch.float32).to(self.device)
        if isinstance(v, list):
            return torch.tensor(v, dtype=torch.float32).to(self.device)
        return v.to(self.device)

    def render_face_normals_face_idx(self, verts, faces, uv_face_attr, elev, azim, radius, look_at_height=0.0, dims=None,
                                     background_type='none'):
        dims = self.dim if dims is None else dims
        cam = self.renderer.get_camera_from_multiple_view(elev, azim, r=radius, look_at_height=look_at_height)
        fvc, fvi, fn = kal.render.mesh.prepare_vertices(verts, faces, self.renderer.camera_projection, camera_transform=cam)
        depth, uv, face_idx, normals_image = kal.render.mesh.rasterize_fused(dims[1], dims[0], fvc, fvi, uv_face_attr, fn)
        depth = self.renderer.normalize_multiple_depth(depth)
        mask = (face_idx > -1).float()[..., None]
        return mask.permute(0, 3, 1, 2), depth.permute(0, 3, 1, 2), normals_image.permute(0, 3, 1, 2), \
            fn.permute(0, 2, 1), face_idx[:, None, :, :]

    def render(self, theta=None, phi=None, radius=None, background=None, use_meta_texture=False, render_cache=None,
               use_median=False, dims=None):
        theta, phi, radius = self._angles(theta), self._angles(phi), self._angles(radius)
        if render_cache is None:
            assert theta is not None and phi is not None and radius is not None
            batch_size = theta.shape[0]
        else:
            batch_size = render_cache["uv_features"].shape[0]
        texture_img, mlp_output = self.get_texture_map()
        background_type, use_render_back = 'none', False
        if background is not None and type(background) == str:
            background_type, use_render_back = background, True
        pred_features, mask, depth, normals, render_cache = self.renderer.render_multiple_view_texture(
            self.mesh.vertices[None].repeat(batch_size, 1, 1), self.mesh.faces, self.face_attributes,
            texture_img.expand(batch_size, -1, -1, -1), elev=theta, azim=phi, radius=radius, look_at_height=self.dy,
            render_cache=render_cache, dims=dims, background_type=background_type)
        mask = mask.detach()
        if use_render_back:
            pred_map, pred_back = pred_features, pred_features
        else:
            pred_back = torch.ones_like(pred_features) * background.reshape(1, 3, 1, 1) if len(background.shape) == 1 else background
            pred_map = pred_back * (1 - mask) + pred_features * mask
        if not use_meta_texture:
            pred_map = pred_map.clamp(0, 1)
            pred_features = pred_features.clamp(0, 1)
        return {'image': pred_map, 'mask': mask, 'background': pred_back, 'foreground': pred_features, 'depth': depth,
                'normals': normals, 'render_cache': render_cache, 'texture_map': texture_img, 'mlp_output': mlp_output}

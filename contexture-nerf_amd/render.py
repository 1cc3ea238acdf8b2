"""Renderer: mirror of src/models/render.py (class Renderer) on the HIP raster path.

Same constructor, `get_camera_from_multiple_view`, `normalize_multiple_depth`,
`render_multiple_view_texture` signatures and the same 5-tuple / render_cache contract
(render.py:159-170).  Differences, all behaviour-preserving: the two kaolin raster passes + the
normals gather run as ONE fused kernel; the mask multiply is folded into the texture sampler.
"""
import numpy as np
import torch
from . import _lib as L
from . import kal


class Renderer:
    def __init__(self, device, dim=(224, 224), interpolation_mode='nearest', fovyangle=np.pi / 3):
        assert interpolation_mode in ['nearest', 'bilinear', 'bicubic'], f'no interpolation mode {interpolation_mode}'
        self.device = device
        self.interpolation_mode = interpolation_mode
        self.camera_projection = kal.render.camera.generate_perspective_projection(fovyangle).to(device)
        self.dim = dim
        self.background = torch.ones(dim).to(device).float()

    @staticmethod
    def get_camera_from_view(elev, azim, r=3.0, look_at_height=0.0):
        x = r * torch.sin(elev) * torch.sin(azim)
        y = r * torch.cos(elev)
        z = r * torch.sin(elev) * torch.cos(azim)
        pos = torch.tensor([x, y, z]).unsqueeze(0)
        look_at = torch.zeros_like(pos)
        look_at[:, 1] = look_at_height
        up = torch.tensor([0.0, 1.0, 0.0]).unsqueeze(0)
        return kal.render.camera.generate_transformation_matrix(pos, look_at, up)

    @staticmethod
    def get_camera_from_multiple_view(elev, azim, r, look_at_height=0.0):
        x = r * torch.sin(elev) * torch.sin(azim)
        y = r * torch.cos(elev)
        z = r * torch.sin(elev) * torch.cos(azim)
        pos = torch.stack([x, y, z], dim=1)
        look_at = torch.zeros_like(pos)
        look_at[:, 1] = look_at_height
        up = torch.ones_like(pos) * torch.tensor([0.0, 1.0, 0.0]).to(pos.device)
        return kal.render.camera.generate_transformation_matrix(pos, look_at, up)

    def normalize_multiple_depth(self, depth_maps):
        """render.py:48-74: per-view masked min/max rescale; background stays 0; same two asserts."""
        lib = L.load()
        d = L.f32c(depth_maps)
        B = d.shape[0]
        out = torch.empty_like(d)
        ws = torch.empty(lib.ctx_normalize_depth_ws_bytes(B), dtype=torch.uint8, device=d.device)
        status = torch.zeros(1, dtype=torch.int32, device=d.device)
        L.check(lib.ctx_normalize_depth(L.ptr(d, torch.float32, "depth_maps"), B, d[0].numel(), L.ptr(out), L.ptr(ws),
                                        L.ptr(status), L.stream()))
        st = int(status.item())        # the reference's asserts sync too (render.py:49-50)
        assert st != 1, 'depth map should be negative'
        assert st != 2, 'depth map should not be empty'
        return out

    def render_multiple_view_texture(self, verts, faces, uv_face_attr, texture_map, elev, azim, radius,
                                     look_at_height=0.0, dims=None, background_type='none', render_cache=None):
        dims = self.dim if dims is None else dims
        if render_cache is None:
            camera_transform = self.get_camera_from_multiple_view(elev, azim, r=radius, look_at_height=look_at_height)
            face_vertices_camera, face_vertices_image, face_normals = kal.render.mesh.prepare_vertices(
                verts, faces, self.camera_projection, camera_transform=camera_transform)
            raw_depth_map, uv_features, face_idx, normals_image = kal.render.mesh.rasterize_fused(
                dims[1], dims[0], face_vertices_camera, face_vertices_image, uv_face_attr, face_normals)
            depth_map = self.normalize_multiple_depth(raw_depth_map)
        else:
            camera_transform = render_cache['camera_transform']
            face_normals = render_cache['face_normals']
            uv_features = render_cache['uv_features']
            face_idx = render_cache['face_idx']
            depth_map = render_cache['depth_map']
            raw_depth_map = render_cache['raw_depth_map']
            face_vertices_image = render_cache['face_vertices_image']
            normals_image = render_cache.get('normals_image')
            if normals_image is None:
                b = torch.arange(face_normals.shape[0], device=face_idx.device).view(-1, 1, 1).expand(-1, *face_idx.shape[1:])
                normals_image = face_normals[b, face_idx]

        mask = (face_idx > -1).float()[..., None]
        image_features = kal.render.mesh.texture_mapping(uv_features, texture_map, mode=self.interpolation_mode,
                                                         mask_idx=face_idx)          # == texture_mapping(...) * mask
        if background_type == 'white':
            image_features = image_features + 1 * (1 - mask)
        elif background_type == 'random':
            image_features = image_features + torch.rand((1, 1, 1, 3)).to(self.device) * (1 - mask)

        render_cache = {'camera_transform': camera_transform, 'uv_features': uv_features, 'face_normals': face_normals,
                        'face_idx': face_idx, 'depth_map': depth_map, 'raw_depth_map': raw_depth_map,
                        'face_vertices_image': face_vertices_image, 'normals_image': normals_image}
        return image_features.permute(0, 3, 1, 2), mask.permute(0, 3, 1, 2), depth_map.permute(0, 3, 1, 2), \
            normals_image.permute(0, 3, 1, 2), render_cache

"""Mesh: mirror of src/models/mesh.py (OBJ import, face normals, normalisation into the unit cube)."""
import copy
import os
import numpy as np
import torch
from . import kal

_ARCHIVE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shapes", "meshes.npz")


def _bundled(obj_path):
    stem = os.path.splitext(os.path.basename(obj_path))[0]
    if not os.path.exists(_ARCHIVE):
        return None
    z = np.load(_ARCHIVE)
    if stem + "_v" not in z.files:
        return None
    return z[stem + "_v"], z[stem + "_f"].astype(np.int64), z[stem + "_vt"], z[stem + "_ft"].astype(np.int64)


class Mesh:
    def __init__(self, obj_path=None, device='cpu', arrays=None):
        if arrays is not None:                      # (vertices, faces, uvs, face_uvs_idx) — test/bench fixtures
            v, f, vt, ft = arrays
            self.vertices = torch.as_tensor(v, dtype=torch.float32).to(device)
            self.faces = torch.as_tensor(f).long().to(device)
            self.vt = torch.as_tensor(vt, dtype=torch.float32)
            self.ft = torch.as_tensor(ft).long()
        elif ".obj" in obj_path and not os.path.exists(obj_path) and _bundled(obj_path) is not None:
            v, f, vt, ft = _bundled(obj_path)          # bundled benchmark meshes live in shapes/meshes.npz
            self.vertices = torch.as_tensor(v, dtype=torch.float32).to(device)
            self.faces = torch.as_tensor(f).long().to(device)
            self.vt = torch.as_tensor(vt, dtype=torch.float32)
            self.ft = torch.as_tensor(ft).long()
        elif ".obj" in obj_path:
            mesh = kal.io.obj.import_mesh(obj_path, with_normals=True, with_materials=False)
            self.vertices = mesh.vertices.to(device)
            self.faces = mesh.faces.to(device)
            self.ft = mesh.face_uvs_idx
            self.vt = mesh.uvs
        else:
            raise ValueError(f"{obj_path} extension not implemented in mesh reader.")
        self.normals, self.face_area = self.calculate_face_normals(self.vertices, self.faces)

    @staticmethod
    def calculate_face_normals(vertices, faces):
        v0, v1, v2 = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
        n = torch.cross(v1 - v0, v2 - v0, dim=-1)
        twice_area = torch.norm(n, dim=-1)
        return n / twice_area[:, None], twice_area / 2

    def normalize_mesh(self, inplace=False, target_scale=1, dy=0):
        mesh = self if inplace else copy.deepcopy(self)
        verts = mesh.vertices
        verts = verts - verts.mean(dim=0)
        scale = torch.max(torch.norm(verts, p=2, dim=1))
        verts = verts / scale
        verts *= target_scale
        verts[:, 1] += dy
        mesh.vertices = verts
        return mesh


def _write_png_rgb(path, rgb):
    """Minimal 8-bit RGB PNG writer (zlib only): the albedo map of export_mesh."""
    import struct, zlib
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = rgb.shape
    assert c == 3
    raw = b''.join(b'\x00' + rgb[i].tobytes() for i in range(h))

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    with open(path, 'wb') as fp:
        fp.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
                 chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def write_textured_obj(path, vertices, faces, uvs, face_uvs_idx, albedo_rgb_u8, name=''):
    """The on-disk result of the paint loop, in the layout the reference's TexturedMeshModel.export_mesh produces
    (src/models/textured_mesh.py:418-474): `{name}albedo.png` (the [T,T,3] uint8 atlas as is, v not flipped),
    `{name}mesh.obj` (mtllib line, `v x y z`, `vt u v`, `usemtl mat0`, `f v/vt v/vt v/vt` 1-based) and `{name}mesh.mtl`
    (material mat0 with map_Kd -> the albedo map)."""
    os.makedirs(path, exist_ok=True)
    _write_png_rgb(os.path.join(path, f'{name}albedo.png'), albedo_rgb_u8)
    v_np, f_np = np.asarray(vertices), np.asarray(faces).astype(np.int64)
    vt_np, ft_np = np.asarray(uvs), np.asarray(face_uvs_idx).astype(np.int64)
    with open(os.path.join(path, f'{name}mesh.obj'), 'w') as fp:
        fp.write(f'mtllib {name}mesh.mtl \n')
        fp.writelines(f'v {v[0]} {v[1]} {v[2]} \n' for v in v_np)
        fp.writelines(f'vt {t[0]} {t[1]} \n' for t in vt_np)
        fp.write('usemtl mat0 \n')
        fp.writelines(f'f {a[0] + 1}/{b[0] + 1} {a[1] + 1}/{b[1] + 1} {a[2] + 1}/{b[2] + 1} \n' for a, b in zip(f_np, ft_np))
    with open(os.path.join(path, f'{name}mesh.mtl'), 'w') as fp:
        fp.write('newmtl mat0 \nKa 1.000000 1.000000 1.000000 \nKd 1.000000 1.000000 1.000000 \nKs 0.000000 0.000000 0.000000 \n'
                 f'Tr 1.000000 \nillum 1 \nNs 0.000000 \nmap_Kd {name}albedo.png \n')

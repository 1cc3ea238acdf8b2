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

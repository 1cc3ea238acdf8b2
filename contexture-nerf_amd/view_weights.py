"""View-weight masks: the torch_scatter.scatter_max seam of the reference
(src/training/trainer.py:155-249), backed by libctxnerf.so.

`create_face_view_map` / `compare_face_normals_between_views` keep the reference's method
signatures; `view_weight_masks` is the fused form that never materialises the [N,4] map, and its
`group=` argument runs the cross-rank all-reduce(MAX) between the two phases when views are
sharded over GPUs.
"""
import torch
from . import _lib as L


def create_face_view_map(face_idx):
    """face_idx [B,1,H,W] i64 -> [N,4] i64 rows (face, view, i, j), (view,row,col) order (trainer.py:155-211)."""
    lib = L.load()
    B, _, H, W = face_idx.shape
    fi = face_idx.to(torch.int64).contiguous()
    rows = torch.empty(B * H * W, 4, dtype=torch.int64, device=fi.device)
    n = torch.zeros(1, dtype=torch.int64, device=fi.device)
    ws = torch.empty(lib.ctx_face_view_map_ws_bytes(B, H, W), dtype=torch.uint8, device=fi.device)
    L.check(lib.ctx_face_view_map(L.ptr(fi, torch.int64, "face_idx"), B, H, W, L.ptr(rows), L.ptr(n), L.ptr(ws), L.stream()))
    return rows[: int(n.item())]


def local_max_z(face_idx, face_normals, max_z=None):
    """Phase 0: per-face max over the local views of face_normals[view,2,face] where the face is visible."""
    lib = L.load()
    B = face_idx.shape[0]
    HW = face_idx[0].numel()
    fi = face_idx.to(torch.int64).contiguous()
    fnz = L.f32c(face_normals[:, 2, :])
    F = fnz.shape[1]
    if max_z is None:
        max_z = torch.full((F,), float('-inf'), device=fi.device)
    vis = torch.empty(B * F, dtype=torch.uint8, device=fi.device)
    L.check(lib.ctx_view_weights_max(L.ptr(fi, torch.int64, "face_idx"), L.ptr(fnz), B, HW, F, L.ptr(max_z, torch.float32),
                                     L.ptr(vis), L.stream()))
    return max_z, fnz


def masks_from_max_z(face_idx, fnz, max_z):
    """Phase 1: mask = ~(z < max_z[face]) ; background True (trainer.py:236-247)."""
    lib = L.load()
    B = face_idx.shape[0]
    HW = face_idx[0].numel()
    fi = face_idx.to(torch.int64).contiguous()
    mask = torch.empty(face_idx.shape, dtype=torch.uint8, device=fi.device)
    L.check(lib.ctx_view_weights_mask(L.ptr(fi), L.ptr(fnz), L.ptr(max_z), B, HW, fnz.shape[1], L.ptr(mask), L.stream()))
    return mask.view(torch.bool) if hasattr(mask, 'view') else mask.bool()


def view_weight_masks(face_idx, face_normals, group=None):
    """face_idx [B,1,H,W] i64, face_normals [B,3,F] f32 -> weight_masks [B,1,H,W] bool.
    With `group` (torch.distributed process group) B is the LOCAL view shard and the per-face maxima are
    all-reduced (MAX) over RCCL; max is exact, so masks equal the single-process result bit for bit."""
    max_z, fnz = local_max_z(face_idx, face_normals)
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(max_z, op=dist.ReduceOp.MAX, group=group)
    return masks_from_max_z(face_idx, fnz, max_z)


def compare_face_normals_between_views(face_view_map, face_normals, face_idx):
    """Reference signature (trainer.py:213). The map argument is redundant with face_idx and ignored."""
    return view_weight_masks(face_idx, face_normals)


def scatter_max(src, index, dim=0):
    """torch_scatter.scatter_max subset used by the reference (1-D, dim=0) — for API parity only."""
    if src.dim() != 1 or dim != 0:
        raise L.CtxError("scatter_max: only the 1-D dim=0 form the reference calls is implemented")
    n = int(index.max().item()) + 1
    out = torch.full((n,), float('-inf'), dtype=src.dtype, device=src.device).scatter_reduce(0, index, src, 'amax')
    return out, None

"""CPU, world_size 2, gloo: the PRODUCT's own multi-rank control flow — `view_weights.view_weight_masks(group=...)`,
`ConTEXTure.define_view_weights` / `paint` (incl. the idle-rank branch) and `MeshBatchPainter.paint_all` (BASELINE configs[3]) —
executed with the HIP kernel calls stubbed at the `_lib` seam: `_lib.load()` returns a numpy stand-in for the three entry points
this path calls, `_lib.ptr()` hands the tensor through.  Nothing from oracle/ does the per-rank work here; the expected values
are computed in plain numpy over ALL views in one process.  What is checked: sharding, the order and content of the collectives
(all-reduce(MAX) of per-face maxima between the two view-weight phases, all-reduce(SUM) of the int64 fixed-point atlas
contribution), that a rank
without views still joins them, and that the result equals the unsharded one BIT FOR BIT (integer sums)."""
import os
import sys
import types
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = W = 12
F = 17
T = 8


class FakeLib:
    """numpy stand-ins with the C-ABI argument order of include/ctx_nerf.h (tensors arrive where pointers would)."""

    def ctx_view_weights_max(self, fi, fnz, B, HW, F_, max_z, vis, stream):
        f = fi.numpy().reshape(B, HW); z = fnz.numpy(); m = max_z.numpy()
        for b in range(B):
            for face in np.unique(f[b][f[b] >= 0]):
                m[face] = max(m[face], z[b, face])
        return 0

    def ctx_view_weights_mask(self, fi, fnz, max_z, B, HW, F_, mask, stream):
        f = fi.numpy().reshape(B, HW); z = fnz.numpy(); m = max_z.numpy(); o = mask.numpy().reshape(B, HW)
        for b in range(B):
            fc = np.clip(f[b], 0, None)
            o[b] = np.where(f[b] >= 0, ~(z[b][fc] < m[fc]), True)
        return 0

    def ctx_texmap_plan_max_res(self):
        return 0                                           # no tile plans here: the scatter takes its plan-less form

    def ctx_uv_scatter_fixed(self, go, uv, face_idx, B, HW, C, T_, plan, frac, acc, stream):
        assert plan is None and acc.dtype == torch.int64
        g = go.numpy().reshape(B * HW, C); u = uv.numpy().reshape(B * HW, 2); f = face_idx.numpy().reshape(B * HW)
        out = acc.numpy()
        x = np.clip((u[:, 0] * T_).astype(np.int64), 0, T_ - 1); y = np.clip(((1 - u[:, 1]) * T_).astype(np.int64), 0, T_ - 1)
        for i in np.nonzero(f >= 0)[0]:                    # nearest-texel stand-in for the bilinear scatter, integer sums
            out[:, y[i], x[i]] += np.rint(g[i].astype(np.float64) * 2.0 ** frac).astype(np.int64)
        return 0

    def ctx_last_error(self):
        return b""


def fake_view(vid):
    """Deterministic synthetic raster of view `vid`: face_idx [H,W], face z-normals [F], uv [H,W,2], painted rgb [3,H,W]."""
    rng = np.random.default_rng(1000 + vid)
    fi = rng.integers(-1, F, (H, W)).astype(np.int64)
    fnz = rng.standard_normal(F).astype(np.float32)
    if vid in (1, 2):
        fnz[5] = 0.25                                     # a tie between views that live on different ranks
    uv = rng.random((H, W, 2)).astype(np.float32)
    rgb = rng.random((3, H, W)).astype(np.float32)
    return fi, fnz, uv, rgb


class FakeMeshModel:
    dy = 0.25
    face_attributes = None

    def __init__(self, key=0):
        self.key = key
        self.mesh = types.SimpleNamespace(faces=torch.zeros(F, 3, dtype=torch.int64), vertices=torch.zeros(5, 3))

    def render_face_normals_face_idx(self, verts, faces, uv_attr, elev, azim, radius, look_at_height=0.0):
        vids = [int(round(float(a) * 100)) for a in azim]  # the test encodes the view id in phi
        fi = np.stack([fake_view(self.key + v)[0] for v in vids]); fn = np.zeros((len(vids), 3, F), np.float32)
        fn[:, 2] = np.stack([fake_view(self.key + v)[1] for v in vids])
        mask = torch.from_numpy((fi >= 0).astype(np.float32))[:, None]
        return mask, mask.clone(), torch.zeros(len(vids), 3, H, W), torch.from_numpy(fn), torch.from_numpy(fi)[:, None]


def make_trainer(rank, world, n_views, key=0):
    from contexture_nerf_amd import config as CFG
    from contexture_nerf_amd.trainer import ConTEXTure
    tr = ConTEXTure.__new__(ConTEXTure)
    tr.cfg = CFG.TrainConfig(); tr.cfg.guide.texture_resolution = T; tr.cfg.optim.views_in_flight = 2
    tr.paint_step, tr.group, tr.rank, tr.world, tr.device = 0, None, rank, world, torch.device('cpu')
    tr.mesh_model = FakeMeshModel(key)
    tr.train_views = [dict(theta=1.0, phi=v / 100.0, radius=1.5, vid=v) for v in range(n_views)]
    tr.view_weights, tr.text_z, tr.diffusion = None, None, types.SimpleNamespace(img2img_step_multi=None)

    def prep(data, image_size=None, num_inference_steps=None):
        fi, fnz, uv, rgb = fake_view(key + data['vid'])
        rc = dict(uv_features=torch.from_numpy(uv)[None], face_idx=torch.from_numpy(fi)[None])
        return dict(text_embeddings=None, inputs=None, original_depth_mask=None, vid=data['vid']), dict(render_cache=rc, rgb=torch.from_numpy(rgb)[None], object_mask=torch.from_numpy((fi >= 0).astype(np.float32))[None, None])
    tr._paint_prepare = prep
    tr._paint_finish = lambda ctx, rgb: (ctx['rgb'], ctx['object_mask'])

    def multi(datas, image_size=None, num_inference_steps=None):
        out = []
        for d in datas:
            kw, ctx = prep(d)
            out.append((ctx['rgb'], ctx['object_mask'], dict(render_cache=ctx['render_cache'])))
        return out
    tr.paint_viewpoints_multi = multi

    def single(data, should_project_back=True, image_size=None, num_inference_steps=None):
        assert should_project_back is False                # paint() scatters itself
        kw, ctx = prep(data)
        tr._last = dict(render_cache=ctx['render_cache'])
        return ctx['rgb'], ctx['object_mask']
    tr.paint_viewpoint = single
    return tr


def expected(n_views, key=0):
    """Unsharded numpy result: masks per view, atlas contribution summed over views."""
    vs = [fake_view(key + v) for v in range(n_views)]
    mx = np.full(F, -np.inf, np.float32)
    for fi, fnz, _, _ in vs:
        for f in np.unique(fi[fi >= 0]):
            mx[f] = max(mx[f], fnz[f])
    masks, contrib = [], np.zeros((4, T, T), np.int64)
    for fi, fnz, uv, rgb in vs:
        fc = np.clip(fi, 0, None)
        m = np.where(fi >= 0, ~(fnz[fc] < mx[fc]), True)
        masks.append(m)
        w = (m & (fi >= 0)).astype(np.float32)
        x = np.clip((uv[..., 0] * T).astype(np.int64), 0, T - 1); y = np.clip(((1 - uv[..., 1]) * T).astype(np.int64), 0, T - 1)
        for i, j in zip(*np.nonzero(fi >= 0)):
            contrib[:3, y[i, j], x[i, j]] += _fix(rgb[:, i, j] * w[i, j])
            contrib[3, y[i, j], x[i, j]] += _fix(w[i, j])
    return np.stack(masks), _unfix(contrib)


def _fix(v):
    return np.rint(np.asarray(v, np.float32).astype(np.float64) * 2.0 ** 32).astype(np.int64)


def _unfix(acc):
    return (acc.astype(np.float64) * 2.0 ** -32).astype(np.float32)


def _atlas(c):
    return c[:3] / np.maximum(c[3:], np.float32(1e-8))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from contexture_nerf_amd import _lib as L, dist as D, view_weights as VW
    from contexture_nerf_amd.batch import MeshBatchPainter, schedule
    L.load = lambda: FakeLib()                              # the seam: no HIP library, no GPU
    L.ptr = lambda t, dtype=None, name="tensor": t
    L.stream = lambda: None
    L.f32c = lambda t, device=None: t.to(torch.float32).contiguous()
    r, w, dev = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    calls = []
    real_ar = dist.all_reduce

    def spy(t, op=dist.ReduceOp.SUM, group=None, **k):
        calls.append((str(op).split('.')[-1], tuple(t.shape)))
        return real_ar(t, op=op, group=group, **k)
    dist.all_reduce = spy

    # 1. view_weights.view_weight_masks(group=...) on the local shard == unsharded masks
    n = 5
    want_masks, want_contrib = expected(n)
    mine = D.shard_views(n, rank, world)
    fi = torch.from_numpy(np.stack([fake_view(v)[0] for v in mine]))[:, None]
    fn = torch.zeros(len(mine), 3, F); fn[:, 2] = torch.from_numpy(np.stack([fake_view(v)[1] for v in mine]))
    got = VW.view_weight_masks(fi, fn, group=dist.group.WORLD)
    assert calls == [('MAX', (F,))]
    assert np.array_equal(got[:, 0].numpy(), want_masks[mine])

    # 2. ConTEXTure.paint, 5 views over 2 ranks (rank 0: 0, 2, 4 -> a pair in flight + a single; rank 1: 1, 3)
    calls.clear()
    tr = make_trainer(rank, world, n)
    atlas, cov = tr.paint()
    assert calls == [('MAX', (F,)), ('SUM', (4, T, T))]
    # integer sums all-reduced as int64: the 2-rank atlas equals the single-process one BIT FOR BIT
    assert np.array_equal(cov.numpy(), want_contrib[3])
    assert np.array_equal(atlas.numpy(), _atlas(want_contrib))
    assert np.array_equal(tr.view_weights[:, 0].numpy(), want_masks[mine])

    # 3. one view, two ranks: rank 1 is idle and must still join both collectives
    calls.clear()
    tr1 = make_trainer(rank, world, 1)
    atlas1, cov1 = tr1.paint()
    assert calls == [('MAX', (F,)), ('SUM', (4, T, T))]
    m1, c1 = expected(1)
    assert (tr1.view_weights is None) == (rank == 1)
    assert np.array_equal(cov1.numpy(), c1[3]) and np.array_equal(atlas1.numpy(), _atlas(c1))

    # 4. BASELINE configs[3] driver: 3 meshes x 3 views over 2 ranks; items of different meshes share a group in flight
    calls.clear()
    trs = [make_trainer(rank, world, 4, key=100 * m) for m in range(3)]
    for t_ in trs:
        t_.diffusion = types.SimpleNamespace(
            img2img_step_multi=lambda kws: [(None, []) for _ in kws],
            img2img_step=lambda te, inp, dm, **k: (None, []))
    bp = MeshBatchPainter(trs, view_ids=[1, 2, 3])
    assert bp.plan == schedule(3, 3, 2) and sorted(sum(bp.plan, [])) == [(m, v) for m in range(3) for v in range(3)]
    assert [len(p) for p in schedule(8, 6, 8)] == [6] * 8 and schedule(8, 6, 8)[6][0] == (1, 0)     # mesh 1 starts on the rank mesh 0 leaves idle
    res = bp.paint_all()
    assert calls == [('MAX', (F,))] * 3 + [('SUM', (4, T, T))] * 3
    for m in range(3):
        vs = [fake_view(100 * m + v) for v in (1, 2, 3)]
        mx = np.full(F, -np.inf, np.float32)
        for fi_, fnz_, _, _ in vs:
            for f in np.unique(fi_[fi_ >= 0]):
                mx[f] = max(mx[f], fnz_[f])
        c = np.zeros((4, T, T), np.int64)
        for fi_, fnz_, uv_, rgb_ in vs:
            fc = np.clip(fi_, 0, None)
            wgt = (np.where(fi_ >= 0, ~(fnz_[fc] < mx[fc]), True) & (fi_ >= 0)).astype(np.float32)
            x = np.clip((uv_[..., 0] * T).astype(np.int64), 0, T - 1); y = np.clip(((1 - uv_[..., 1]) * T).astype(np.int64), 0, T - 1)
            for i, j in zip(*np.nonzero(fi_ >= 0)):
                c[:3, y[i, j], x[i, j]] += _fix(rgb_[:, i, j] * wgt[i, j]); c[3, y[i, j], x[i, j]] += _fix(wgt[i, j])
        c = _unfix(c)
        assert np.array_equal(res[m][1].numpy(), c[3]) and np.array_equal(res[m][0].numpy(), _atlas(c))
    torch.save(torch.tensor(1), os.path.join(tmp, f"ok{rank}"))
    dist.destroy_process_group()


def test_product_multi_rank_control_flow_world2(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")

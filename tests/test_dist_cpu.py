"""CPU, world_size 2, gloo: the N>1 path of the view-sharded paint loop.  The sharding / collective logic is the
product's (contexture_nerf_amd.dist); the per-rank local work is done by the oracle here (tests may call it), so
the test checks exactly what the multi-GPU run relies on: shard -> all-reduce(MAX) of per-face maxima gives masks
bit-identical to the single-process result; all-reduce(SUM) of atlas contributions equals the unsharded sum."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from contexture_nerf_amd import dist as D
    from oracle import geometry as og
    r, w, dev = D.init(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    rng = np.random.default_rng(0)                       # same data on every rank
    B, H, W, F = 7, 24, 20, 31
    face_idx = rng.integers(-1, F, (B, H, W)).astype(np.int64)
    fnz = rng.standard_normal((B, F)).astype(np.float32)
    fnz[3, 5] = fnz[0, 5]                                # tie across ranks
    mine = D.shard_views(B, rank, world)
    assert mine == [k for k in range(B) if k % world == rank]
    # phase 0 on the local shard (oracle), exchange, phase 1 locally
    local_max, _ = og.view_weights(face_idx[mine], fnz[mine])
    t = torch.from_numpy(local_max.copy())
    D.all_reduce_max_(t)
    gmax = t.numpy()
    masks = np.stack([np.where(face_idx[k] >= 0, ~(fnz[k][np.clip(face_idx[k], 0, None)] < gmax[np.clip(face_idx[k], 0, None)]), True)
                      for k in mine])
    full_max, full_masks = og.view_weights(face_idx, fnz)
    seen = np.isfinite(full_max)
    assert np.array_equal(gmax[seen], full_max[seen])
    assert np.array_equal(masks, full_masks[mine])
    # atlas: per-rank int64 fixed-point contributions (weights in channel 3) summed over ranks == the single-process sum, bit for bit
    T = 16
    contrib_all = rng.random((B, 4, T, T)).astype(np.float32)
    contrib_all[:, 3] = (contrib_all[:, 3] > 0.5)
    fixed_all = np.rint(contrib_all.astype(np.float64) * 2.0 ** 32).astype(np.int64)
    local = torch.from_numpy(fixed_all[mine].sum(0))
    atlas, cov = D.merge_atlas(local.clone())
    tot = (fixed_all.sum(0).astype(np.float64) * 2.0 ** -32).astype(np.float32)          # any order: integer sums
    assert np.array_equal(cov.numpy(), tot[3])
    assert np.array_equal(atlas.numpy(), tot[:3] / np.maximum(tot[3:], np.float32(1e-8)))
    # a float32 contrib still goes through (summed as floats, per-rank sums then ranks)
    atlas_f, cov_f = D.merge_atlas(torch.from_numpy(contrib_all[mine].sum(0)))
    np.testing.assert_allclose(cov_f.numpy(), tot[3], rtol=0, atol=1e-5)
    # ray path (configs[4]): contiguous row tiles, ragged H, gathered image == the unsharded one
    from contexture_nerf_amd import volume_render as vr
    Hh = 13
    img = torch.arange(Hh * 5 * 3, dtype=torch.float32).reshape(Hh, 5, 3)
    r0, r1 = vr.shard_rows(Hh, rank, world)
    assert (r0, r1) == ((0, 7) if rank == 0 else (7, 13))
    assert torch.equal(vr.gather_rows(img[r0:r1].clone(), Hh), img)
    assert torch.equal(vr.gather_rows(img[r0:r1, :, 0].clone(), Hh), img[:, :, 0])
    torch.save(torch.tensor(1), os.path.join(tmp, f"ok{rank}"))
    dist.destroy_process_group()


def test_view_shard_allreduce_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_shard_views_covers_all():
    from contexture_nerf_amd.dist import shard_views
    for n in (6, 7, 8, 10):
        for world in (1, 2, 4, 8):
            got = sorted(k for r in range(world) for k in shard_views(n, r, world))
            assert got == list(range(n))

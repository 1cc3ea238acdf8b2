"""BASELINE configs[3]: a batch of meshes, 6 views each, painted on a view-sharded node (SURVEY section 8d cfg 4, 8e; the intended
multi-mesh usage is visible in the reference's batch driver generate_survey_textures.py:117-162, which loops
`ConTEXTure(cfg).paint()` over meshes one after another on one GPU).

Work items are (mesh, view) pairs in mesh-major order; item k goes to rank k mod world, so with 6 views and 8 ranks the views of
mesh m+1 start on the ranks mesh m leaves idle and every rank ends up with the same number of denoise loops (48 items / 8 ranks).
Exchange steps, both over RCCL, one of each per mesh:
  phase A  all-reduce(MAX) of the per-face view-weight maxima [F]    (before painting: the masks gate the scatter)
  phase B  each rank paints its items, `views_in_flight` denoise loops at a time (items of different meshes may share a group)
  phase C  all-reduce(SUM) of the atlas contribution [3+1, T, T] as int64 fixed-point sums (bit-identical for any world size)
Every rank calls the collectives of every mesh in mesh order, whether or not it holds views of that mesh."""
import torch
from . import dist as D


def schedule(n_meshes, n_views, world):
    """-> plan[rank] = ordered [(mesh, view)]; item k = mesh * n_views + view -> rank k mod world."""
    plan = [[] for _ in range(world)]
    for k in range(n_meshes * n_views):
        plan[k % world].append((k // n_views, k % n_views))
    return plan


class MeshBatchPainter:
    """trainers: one ConTEXTure per mesh (same diffusion engine, same process group); view_ids: the dataset indices painted per
    mesh (Zero123PlusDataset views 1..6 by default)."""

    def __init__(self, trainers, view_ids=None, group=None):
        self.trainers = list(trainers)
        self.view_ids = list(view_ids) if view_ids is not None else list(range(1, 7))
        self.group = group
        t0 = self.trainers[0]
        self.rank, self.world, self.device = t0.rank, t0.world, t0.device
        self.plan = schedule(len(self.trainers), len(self.view_ids), self.world)

    def paint_all(self, image_size=None, num_inference_steps=None):
        """-> [(atlas [3,T,T], coverage [T,T])] per mesh (identical on every rank)."""
        mine = self.plan[self.rank]
        # phase A: view-weight masks of the local views of every mesh; one all-reduce(MAX) per mesh
        slot = {}
        for m, tr in enumerate(self.trainers):
            ids = [self.view_ids[v] for (mm, v) in mine if mm == m]
            tr.define_view_weights(ids)
            for j, vid in enumerate(ids):
                slot[(m, vid)] = j
        # phase B: denoise loops, `views_in_flight` at a time across mesh boundaries
        T = self.trainers[0].cfg.guide.texture_resolution
        contrib = [torch.zeros(4, T, T, dtype=torch.int64, device=self.device) for _ in self.trainers]   # 2^-32 fixed point
        diffusion = self.trainers[0].diffusion
        infl = max(1, int(getattr(self.trainers[0].cfg.optim, 'views_in_flight', 3)))
        vpe = int(getattr(self.trainers[0].cfg.optim, 'views_per_eval', 0))
        if vpe > 1 and hasattr(diffusion, 'img2img_step_batched'):
            infl = 2 * vpe if len(mine) >= 2 * vpe else vpe          # lockstep groups of vpe views (batch 2 x vpe), two groups in flight
        if not hasattr(diffusion, 'img2img_step_multi'):
            infl = 1
        for j in range(0, len(mine), infl):
            grp = mine[j:j + infl]
            preps = []
            for (m, v) in grp:
                tr = self.trainers[m]
                preps.append(tr._paint_prepare(tr.train_views[self.view_ids[v]], image_size, num_inference_steps))
            if len(grp) > 1 and vpe > 1 and hasattr(diffusion, 'img2img_step_batched'):
                outs = diffusion.img2img_step_batched([p[0] for p in preps], views_per_eval=vpe)
            elif len(grp) > 1:
                outs = diffusion.img2img_step_multi([p[0] for p in preps])
            else:
                kw = dict(preps[0][0])
                outs = [diffusion.img2img_step(kw.pop('text_embeddings'), kw.pop('inputs'), kw.pop('original_depth_mask'), **kw)]
            for (m, v), (kw, ctx), (rgb, _) in zip(grp, preps, outs):
                tr = self.trainers[m]
                rgb_output, obj_mask = tr._paint_finish(ctx, rgb)
                k = slot[(m, self.view_ids[v])]
                tr.project_back_scatter(ctx['render_cache'], rgb_output, tr.view_weights[k:k + 1] & (obj_mask > 0), acc=contrib[m])
        # phase C: one all-reduce(SUM) per mesh
        res = []
        for m, tr in enumerate(self.trainers):
            atlas, cov = D.merge_atlas(contrib[m], self.group)
            tr.atlas, tr.atlas_coverage = atlas, cov
            res.append((atlas, cov))
        return res

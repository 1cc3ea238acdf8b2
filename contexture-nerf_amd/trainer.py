"""ConTEXTure (paint path only): mirror of the hot-path members of src/training/trainer.py —
`define_view_weights` :370-415 (-> create_face_view_map / compare_face_normals_between_views :155-249),
`paint_viewpoint` :971-1117, plus the per-view loop north_star describes (views sharded one per GPU,
UV back-projection as a scatter into the atlas, RCCL all-reduce of the atlas).

Not built: the 5000-iteration Zero123++ SDS loop (`paint_zero123plus`), eval/video/mesh export,
wandb/loguru logging.  `project_back` has no body in the reference (trainer.py:1079-1089 calls an undefined
method); `project_back_scatter` below is the direct UV-scatter form (parity unpinned, documented in DESIGN.md).
"""
import math
import torch
import torch.nn.functional as F
from . import _lib as L
from . import dist as D
from . import utils, view_weights
from .run_nerf_helpers import get_embedder, NeRF2D
from .textured_mesh import TexturedMeshModel
from .views_dataset import Zero123PlusDataset, MultiviewDataset


class ConTEXTure:
    def __init__(self, cfg, device=None, diffusion=None, mesh_arrays=None, group=None):
        self.cfg = cfg
        self.paint_step = 0
        self.group = group
        self.rank, self.world, dev = D.init()
        self.device = device if device is not None else dev
        utils.seed_everything(self.cfg.optim.seed)
        self.uv_embedder, input_ch = get_embedder(10)
        self.texture_mlp = NeRF2D(D=8, W=256, input_ch=input_ch, output_ch=3, skips=[4]).to(self.device)
        self.mesh_model = TexturedMeshModel(self.cfg.guide, render_grid_size=self.cfg.render.train_grid_size,
                                            texture_resolution=self.cfg.guide.texture_resolution, device=self.device,
                                            texture_mlp=self.texture_mlp, uv_embedder=self.uv_embedder, mesh_arrays=mesh_arrays)
        self.diffusion = diffusion
        self.text_z = None
        ds = Zero123PlusDataset if self.cfg.guide.use_zero123plus else MultiviewDataset
        self.train_views = list(ds(self.cfg.render, self.device))
        self.view_weights = None
        self.back_im = torch.full((3, 64, 64), 0.5, device=self.device)

    def _offset_phi(self, phi):
        """phi - front_offset, wrapped into [0, 2 pi) (trainer.py:378-380, 974-976): ONE helper for define_view_weights and
        paint_viewpoint, so that view-weight masks and painted views always share their cameras."""
        phi = float(phi) - math.radians(self.cfg.render.front_offset)
        return float(phi + 2 * math.pi if phi < 0 else phi)

    # ---- trainer.py:370-415 ----------------------------------------------------------------------------------
    def define_view_weights(self, view_ids=None):
        """Weight masks for the (local shard of) views; with >1 ranks the per-face maxima are all-reduced (MAX)."""
        ids = list(range(len(self.train_views))) if view_ids is None else list(view_ids)
        thetas = torch.tensor([self.train_views[i]['theta'] for i in ids], device=self.device)
        # same front_offset shift and wrap as paint_viewpoint (trainer.py:378-380): the masks must come from the painted cameras
        phis = torch.tensor([self._offset_phi(self.train_views[i]['phi']) for i in ids], device=self.device)
        radii = torch.tensor([float(self.train_views[i]['radius']) for i in ids], device=self.device)
        B = len(ids)
        mm = self.mesh_model
        mask, depth, normals_image, face_normals, face_idx = mm.render_face_normals_face_idx(
            mm.mesh.vertices[None].repeat(B, 1, 1), mm.mesh.faces, mm.face_attributes, elev=thetas, azim=phis, radius=radii,
            look_at_height=mm.dy)
        group = self.group if self.world > 1 else None
        if self.world > 1 and group is None:
            import torch.distributed as dist
            group = dist.group.WORLD
        self.view_weights = view_weights.view_weight_masks(face_idx, face_normals, group=group)
        self._vw_cache = dict(mask=mask, depth=depth, face_idx=face_idx, face_normals=face_normals)
        return self.view_weights

    # ---- trainer.py:971-1117 ---------------------------------------------------------------------------------
    @torch.no_grad()   # the paint pass is inference; the texture field keeps activations only when gradients are on
    def _paint_prepare(self, data, image_size=None, num_inference_steps=None):
        """Everything of paint_viewpoint up to the diffusion call: render, crop box; returns (img2img kwargs, context)."""
        theta, phi, radius = data['theta'], data['phi'], data['radius']
        phi = self._offset_phi(phi)
        G = self.cfg.render.train_grid_size
        background = F.interpolate(self.back_im.unsqueeze(0), (G, G), mode='bilinear', align_corners=False) \
            if not self.cfg.guide.use_background_color else torch.tensor([0.0, 0.8, 0.0], device=self.device)
        outputs = self.mesh_model.render(theta=theta, phi=phi, radius=radius, background=background)
        render_cache = outputs['render_cache']
        depth_render = outputs['depth']
        outputs = self.mesh_model.render(background=background, render_cache=render_cache, use_median=self.paint_step > 1)
        rgb_render = outputs['image']
        z_normals = outputs['normals'][:, -1:, :, :].clamp(0, 1)
        object_mask = outputs['mask']
        min_h, min_w, max_h, max_w = utils.get_nonzero_region_tuple(object_mask[0, 0])
        crop = lambda x: x[:, :, min_h:max_h, min_w:max_w]
        cropped_rgb_render, cropped_depth_render, cropped_update_mask = crop(rgb_render), crop(depth_render), crop(object_mask)
        text_z = self.text_z if self.text_z is not None else self.diffusion.get_text_embeds([self.cfg.guide.text])
        kw = dict(text_embeddings=text_z, inputs=cropped_rgb_render.detach(), original_depth_mask=cropped_depth_render.detach(),
                  guidance_scale=self.cfg.guide.guidance_scale, strength=1.0, update_mask=cropped_update_mask,
                  fixed_seed=self.cfg.optim.seed, intermediate_vis=False,
                  num_inference_steps=num_inference_steps or self.cfg.guide.num_inference_steps,
                  image_size=image_size or self.cfg.guide.sd_image_size)
        ctx = dict(render_cache=render_cache, z_normals=z_normals, rgb_render=rgb_render, object_mask=object_mask,
                   box=(min_h, min_w, max_h, max_w), crop_hw=(cropped_rgb_render.shape[2], cropped_rgb_render.shape[3]))
        return kw, ctx

    @torch.no_grad()
    def _paint_finish(self, ctx, cropped_rgb_output):
        cropped_rgb_output = F.interpolate(cropped_rgb_output, ctx['crop_hw'], mode='bilinear', align_corners=False)
        min_h, min_w, max_h, max_w = ctx['box']
        rgb_output = ctx['rgb_render'].clone()
        rgb_output[:, :, min_h:max_h, min_w:max_w] = cropped_rgb_output
        self._last = dict(render_cache=ctx['render_cache'], z_normals=ctx['z_normals'])
        return rgb_output, ctx['object_mask']

    def paint_viewpoint(self, data, should_project_back=False, image_size=None, num_inference_steps=None):
        kw, ctx = self._paint_prepare(data, image_size, num_inference_steps)
        te = kw.pop('text_embeddings'); inp = kw.pop('inputs'); dm = kw.pop('original_depth_mask')
        cropped_rgb_output, _ = self.diffusion.img2img_step(te, inp, dm, **kw)
        return self._paint_finish(ctx, cropped_rgb_output)

    def paint_viewpoints_multi(self, datas, image_size=None, num_inference_steps=None):
        """Several views painted with their denoise loops in flight together (StableDiffusion.img2img_step_multi): same result
        per view as paint_viewpoint.  Returns [(rgb_output, object_mask, last)] per view."""
        preps = [self._paint_prepare(d, image_size, num_inference_steps) for d in datas]
        outs = self.diffusion.img2img_step_multi([p[0] for p in preps])
        res = []
        for (kw, ctx), (rgb, _) in zip(preps, outs):
            rgb_output, mask = self._paint_finish(ctx, rgb)
            res.append((rgb_output, mask, self._last))
        return res

    def paint_viewpoints_pair(self, data_a, data_b, image_size=None, num_inference_steps=None):
        return self.paint_viewpoints_multi([data_a, data_b], image_size, num_inference_steps)

    # ---- north_star "UV back-projection" (absent in the reference) ----------------------------
    def project_back_scatter(self, render_cache, rgb_output, weight_mask):
        """Scatter a painted view into the atlas: contrib[0:3] += w*rgb, contrib[3] += w at the 4 bilinear texels of
        each visible pixel (w = view weight mask).  Implemented with the texture-sampling backward kernel."""
        lib = L.load()
        uv = render_cache['uv_features']
        face_idx = render_cache['face_idx']
        B, H, W, _ = uv.shape
        T = self.cfg.guide.texture_resolution
        w = weight_mask.to(torch.float32).permute(0, 2, 3, 1)
        go = torch.cat([rgb_output.permute(0, 2, 3, 1) * w, w], dim=-1).contiguous()
        contrib = torch.zeros(4, T, T, device=uv.device)
        L.check(lib.ctx_texture_mapping_bwd(L.ptr(go, torch.float32), L.ptr(L.f32c(uv)), B, H * W, 4, T,
                                            L.ptr(face_idx.contiguous(), torch.int64), L.ptr(contrib), L.stream()))
        return contrib

    def paint(self, image_size=None, num_inference_steps=None):
        """Per-view paint loop with views sharded one per rank and one atlas all-reduce at the end."""
        n = len(self.train_views)
        mine = D.shard_views(n, self.rank, self.world)
        T = self.cfg.guide.texture_resolution
        contrib = torch.zeros(4, T, T, device=self.device)
        if mine:
            masks = self.define_view_weights(mine)
        else:                                           # idle rank still joins the collectives
            F_ = self.mesh_model.mesh.faces.shape[0]
            D.all_reduce_max_(torch.full((F_,), float('-inf'), device=self.device), self.group)
        # a rank that owns several views keeps `views_in_flight` of them (default 3) in the denoise loop at once
        infl = max(1, int(getattr(self.cfg.optim, 'views_in_flight', 3)))
        if not hasattr(self.diffusion, 'img2img_step_multi'):
            infl = 1
        j = 0
        while j < len(mine):
            grp = mine[j:j + infl]
            if len(grp) > 1:
                res = self.paint_viewpoints_multi([self.train_views[k] for k in grp], image_size=image_size,
                                                  num_inference_steps=num_inference_steps)
                for o, (rgb, obj_mask, last) in enumerate(res):
                    contrib += self.project_back_scatter(last['render_cache'], rgb, masks[j + o:j + o + 1] & (obj_mask > 0))
            else:
                rgb, obj_mask = self.paint_viewpoint(self.train_views[grp[0]], image_size=image_size, num_inference_steps=num_inference_steps)
                contrib += self.project_back_scatter(self._last['render_cache'], rgb, masks[j:j + 1] & (obj_mask > 0))
            j += len(grp)
        atlas, coverage = D.merge_atlas(contrib, self.group)
        self.atlas, self.atlas_coverage = atlas, coverage
        return atlas, coverage

    def export(self, path=None):
        """The outputs the reference writes after painting (src/training/trainer.py:954-968 -> export_mesh): mesh.obj / mesh.mtl /
        albedo.png with the painted atlas (rank 0 only; uncovered texels keep the texture field's colour)."""
        if D.dist.is_initialized() and D.dist.get_rank(self.group) != 0:
            return None
        path = str(self.cfg.log.exp_dir / 'mesh') if path is None else str(path)
        with torch.no_grad():
            base = self.mesh_model.get_texture_map()[0]
            cov = (self.atlas_coverage > 0)[None, None].to(base.dtype)
            tex = base * (1 - cov) + self.atlas[None, :3] * cov
        self.mesh_model.export_mesh(path, texture=tex)
        return path

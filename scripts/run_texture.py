"""`python -m scripts.run_texture --config_path=configs/text_guided/<x>.yaml [--a.b=value]` — the reference's
documented entry (README.md:67; its file is scripts/run_contexture.py:1-17).  Runs the per-view paint path on the
HIP kernels; under torchrun the views are sharded one per GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from contexture_nerf_amd import config as cfgmod  # noqa: E402
from contexture_nerf_amd.trainer import ConTEXTure  # noqa: E402
from contexture_nerf_amd.stable_diffusion_depth import StableDiffusion  # noqa: E402


@cfgmod.wrap()
def main(cfg: cfgmod.TrainConfig):
    trainer = ConTEXTure(cfg)
    trainer.diffusion = StableDiffusion(trainer.device, model_name=cfg.guide.diffusion_name, seed=cfg.optim.seed)
    exp = cfg.log.exp_dir
    os.makedirs(exp, exist_ok=True)
    cfgmod.dump(cfg, os.path.join(exp, 'config.yaml'))
    if cfg.log.eval_only:
        n = trainer.full_eval()
        print(f"full_eval: {n} views -> {exp}/results")
        return
    if cfg.optim.sds_iterations > 0:                    # the reference's live paint(): front view + SDS against Zero123++
        log = trainer.paint_zero123plus(cfg.optim.sds_iterations)
        if trainer.rank == 0:
            print(f"SDS: {len(log)} iterations, last loss {log[-1]['loss']:.4f}")
            trainer.full_eval(size=cfg.log.eval_size)
        return
    atlas, coverage = trainer.paint()
    if trainer.rank == 0:
        torch.save({'atlas': atlas.cpu(), 'coverage': coverage.cpu()}, os.path.join(exp, 'atlas.pt'))
        print(f"painted {len(trainer.train_views)} views -> {exp}/atlas.pt  coverage {float((coverage > 0).float().mean()):.3f}")
    if cfg.log.save_mesh:                               # src/training/trainer.py:962-968: mesh.obj / mesh.mtl / albedo.png
        out = trainer.export()
        if out:
            print(f"mesh written to {out}")


if __name__ == '__main__':
    main()

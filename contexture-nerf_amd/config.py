"""Config surface: the dataclasses of src/configs/train_config.py (same field names and defaults) plus a
minimal pyrallis-compatible loader (pyrallis is not installable offline): YAML via --config_path and dotted
`--a.b=value` overrides, as `python -m scripts.run_texture --config_path=... --optim.seed=3`."""
import dataclasses
import sys
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Tuple, get_type_hints
import yaml


@dataclass
class RenderConfig:
    train_grid_size: int = 1200
    eval_grid_size: int = 1024
    radius: float = 1.5
    overhead_range: float = 40
    front_range: float = 70
    front_offset: float = 0.0
    n_views: int = 8
    base_theta: float = 60
    views_before: List[Tuple[float, float]] = field(default_factory=list)
    views_after: List[Tuple[float, float]] = field(default_factory=lambda: [[180, 30], [180, 150]])
    alternate_views: bool = True


@dataclass
class GuideConfig:
    text: str = ''
    shape_path: str = 'shapes/spot_triangulated.obj'
    append_direction: bool = False
    concept_name: Optional[str] = None
    concept_path: Optional[Path] = None
    diffusion_name: str = 'stabilityai/stable-diffusion-2-depth'
    second_model_type: Optional[str] = None
    individual_control_of_conditions: bool = False
    guidance_scale_i: Optional[int] = None
    guidance_scale_t: Optional[int] = None
    use_zero123plus: Optional[bool] = True
    guess_mode: Optional[bool] = False
    shape_scale: float = 0.6
    dy: float = 0.25
    texture_resolution: int = 1024
    texture_interpolation_mode: str = 'bilinear'
    guidance_scale: float = 7.5
    use_inpainting: bool = True
    reference_texture: Optional[Path] = None
    initial_texture: Optional[Path] = None
    use_background_color: bool = False
    background_img: str = 'textures/brick_wall.png'
    z_update_thr: float = 0.2
    strict_projection: bool = True
    # --- additions of this build (not in the reference): BASELINE.json configs vary the SD image size ---
    sd_image_size: int = 512
    num_inference_steps: int = 50
    zero123plus_model_dir: Optional[str] = None      # LOCAL directory in the zero123plus pipeline layout (vision_encoder/, feature_extractor_clip/,
                                                     # tokenizer/, text_encoder/, model_index.json, optionally unet/ vae/ controlnet/): the condition
                                                     # path of src/zero123plus.py:772-803; nothing is fetched by name


@dataclass
class OptimConfig:
    seed: int = 0
    lr: float = 1e-2
    min_timestep: float = 0.02
    max_timestep: float = 0.98
    no_noise: bool = False
    learn_max_z_normals: bool = True
    alpha: float = -100
    # additions of this build (not in the reference's OptimConfig):
    views_in_flight: int = 3         # denoise loops a rank keeps in flight on one GPU (n HIP streams over one weight blob)
    views_per_eval: int = 0          # > 1: full groups of that many views a rank owns are denoised in lockstep as ONE UNet evaluation of batch
                                     # 2 x views_per_eval (StableDiffusion.img2img_step_batched); left-over views and 0 / 1: views_in_flight streams of batch 2
    sds_iterations: int = 0          # 0: the per-view paint loop (north_star).  > 0: the reference's live paint() = paint_zero123plus
                                     # with that many SDS iterations (the reference hard-codes 5000, src/training/trainer.py:662)


@dataclass
class LogConfig:
    exp_name: str = 'default'
    exp_root: Path = Path('experiments/')
    eval_only: bool = False
    eval_size: int = 10
    full_eval_size: int = 100
    save_mesh: bool = True
    vis_diffusion_steps: bool = False
    log_images: bool = True

    @property
    def exp_dir(self) -> Path:
        return self.exp_root / self.exp_name


@dataclass
class TrainConfig:
    log: LogConfig = field(default_factory=LogConfig)
    render: RenderConfig = field(default_factory=RenderConfig)
    optim: OptimConfig = field(default_factory=OptimConfig)
    guide: GuideConfig = field(default_factory=GuideConfig)


def _coerce(value, typ):
    origin = getattr(typ, '__origin__', None)
    if value is None:
        return None
    if origin is not None and type(None) in getattr(typ, '__args__', ()):      # Optional[T]
        inner = [a for a in typ.__args__ if a is not type(None)][0]
        return _coerce(value, inner)
    if typ is bool:
        return value if isinstance(value, bool) else str(value).lower() in ('1', 'true', 'yes')
    if typ in (int, float, str):
        return typ(value)
    if typ is Path:
        return Path(value)
    return value


def _apply(obj, key, value):
    hints = get_type_hints(type(obj))
    if key not in hints:
        raise KeyError(f"unknown config field {type(obj).__name__}.{key}")     # pyrallis raises on unknown keys too
    cur = getattr(obj, key)
    if dataclasses.is_dataclass(cur):
        if not isinstance(value, dict):
            raise TypeError(f"{key}: expected a mapping")
        for k, v in value.items():
            _apply(cur, k, v)
    else:
        setattr(obj, key, _coerce(value, hints[key]))


def parse(config_class=TrainConfig, argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = config_class()
    overrides = []
    path = None
    for a in argv:
        if not a.startswith('--') or '=' not in a:
            raise SystemExit(f"unrecognised argument {a!r} (use --key=value)")
        k, v = a[2:].split('=', 1)
        if k == 'config_path':
            path = v
        else:
            overrides.append((k, yaml.safe_load(v)))
    if path:
        with open(path) as f:
            for k, v in (yaml.safe_load(f) or {}).items():
                _apply(cfg, k, v)
    for k, v in overrides:
        obj = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            obj = getattr(obj, p)
        _apply(obj, parts[-1], v)
    return cfg


def dump(cfg, path):
    def enc(o):
        if dataclasses.is_dataclass(o):
            return {f.name: enc(getattr(o, f.name)) for f in dataclasses.fields(o)}
        if isinstance(o, Path):
            return str(o)
        if isinstance(o, (list, tuple)):
            return [enc(x) for x in o]
        return o
    with open(path, 'w') as f:
        yaml.safe_dump(enc(cfg), f)


def wrap():
    """pyrallis.wrap() look-alike: `@wrap() def main(cfg: TrainConfig)`."""
    def deco(fn):
        def inner(*a, **k):
            cls = list(get_type_hints(fn).values())[0]
            return fn(parse(cls), *a, **k)
        return inner
    return deco

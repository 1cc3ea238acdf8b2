"""contexture-nerf_amd — MI355X (gfx950) hot path of ConTEXTure's per-view texture painting loop.

Host side = Python mirrors of the reference's call seams (kaolin, torch-scatter, run_nerf_helpers,
diffusers UNet/PNDM); device side = libctxnerf.so (hand-written HIP, C-ABI in include/ctx_nerf.h).
Import as `contexture_nerf_amd` (the directory name carries a hyphen; the sibling shim package
`contexture_nerf_amd/` points its __path__ here).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]

"""ctypes binding of libctxnerf.so (the C-ABI declared in include/ctx_nerf.h).

The product path has NO CPU fallback: if the library is missing, or a tensor is not a contiguous
device tensor of the expected dtype, this module raises.  PyTorch only supplies device memory and
the current HIP stream.
"""
import ctypes as C
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libctxnerf.so")

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/ctx_nerf.h one to one
SIGNATURES = {
    "ctx_version": (_i32, []),
    "ctx_last_error": (C.c_char_p, []),
    "ctx_device_check": (_i32, []),
    "ctx_prepare_vertices": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ctx_rasterize_ws_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "ctx_rasterize_fwd": (_i32, [_i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _i64, _vp]),
    "ctx_rasterize_fused": (_i32, [_i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ctx_normalize_depth_ws_bytes": (_i64, [_i32]),
    "ctx_normalize_depth": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ctx_texture_mapping_fwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ctx_texture_mapping_bwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ctx_texmap_bwd_plan_bytes": (_i64, [_i32, _i32, _i32]),
    "ctx_texmap_bwd_plan": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ctx_texture_mapping_bwd_binned_ws_bytes": (_i64, [_i32, _i32]),
    "ctx_texture_mapping_bwd_binned": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ctx_texmap_plan_max_res": (_i32, []),
    "ctx_texmap_plan_stale": (_i32, [_vp, _vp]),
    "ctx_uv_scatter_fixed": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "ctx_fixed_to_float": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "ctx_texture_pack4": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "ctx_texture_mapping_packed_fwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ctx_view_weights_max": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ctx_view_weights_mask": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ctx_face_view_map_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "ctx_face_view_map": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ctx_embed_fwd": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "ctx_uvmlp_packed_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "ctx_uvmlp_pack": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ctx_uvmlp_fwd": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ctx_uvmlp_saved_bytes": (_i64, [_i64, _i32, _i32, _i32]),
    "ctx_uvmlp_fwd_save": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ctx_uvmlp_bwd_ws_bytes": (_i64, [_i64, _i32, _i32]),
    "ctx_uvmlp_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ctx_get_rays": (_i32, [_i32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "ctx_raymarch_composite_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctx_unet_create": (_vp, [_vp]),
    "ctx_unet_destroy": (None, [_vp]),
    "ctx_unet_param_count": (_i32, [_vp]),
    "ctx_unet_param_name": (C.c_char_p, [_vp, _i32]),
    "ctx_unet_param_shape": (_i32, [_vp, _i32, _vp]),
    "ctx_unet_weight_bytes": (_i64, [_vp]),
    "ctx_unet_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32, _i32]),
    "ctx_unet_bind": (_i32, [_vp, _vp, _vp, _i64]),
    "ctx_unet_set_param": (_i32, [_vp, _i32, _vp, _vp]),
    "ctx_unet_forward": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ctx_controlnet_create": (_vp, [_vp, _i32]),
    "ctx_controlnet_residual_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ctx_controlnet_cond_cache_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ctx_controlnet_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ctx_unet_set_residuals": (_i32, [_vp, _vp, C.c_float]),
    "ctx_unet_ref_bank_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ctx_unet_workspace_bytes_ref": (_i64, [_vp, _i32, _i32, _i32, _i32, _i32, _i32]),
    "ctx_unet_forward_ref": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "ctx_unet_set_residual_fp32": (_i32, [_vp, _i32]),
    "ctx_unet_stats": (_i32, [_vp, _i32, _vp, _vp]),
    "ctx_unet_set_taps": (_i32, [_vp, _vp, _i64]),
    "ctx_unet_tap_count": (_i32, [_vp]),
    "ctx_unet_tap_info": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "ctx_vae_create": (_vp, [_vp]),
    "ctx_vae_destroy": (None, [_vp]),
    "ctx_vae_param_count": (_i32, [_vp]),
    "ctx_vae_param_name": (C.c_char_p, [_vp, _i32]),
    "ctx_vae_param_shape": (_i32, [_vp, _i32, _vp]),
    "ctx_vae_weight_bytes": (_i64, [_vp]),
    "ctx_vae_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ctx_vae_bind": (_i32, [_vp, _vp, _vp, _i64]),
    "ctx_vae_set_param": (_i32, [_vp, _i32, _vp, _vp]),
    "ctx_vae_decode": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ctx_vae_decoder_param_count": (_i32, [_vp]),
    "ctx_vae_encode_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ctx_vae_encode": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ctx_vae_flops": (C.c_double, [_vp]),
    "ctx_vae_encode_train_workspace_bytes": (_i64, [_vp, _i32, _i32, _i32]),
    "ctx_vae_encode_train": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ctx_vae_encode_bwd": (_i32, [_vp, _vp, _f32, _vp, _vp]),
    "ctx_gemm_f16": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ctx_conv3x3_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ctx_groupnorm_f16": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp]),
    "ctx_layernorm_f16": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp]),
    "ctx_attention_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "ctx_attention_f16": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _i32, _vp, _vp]),
    "ctx_groupnorm_ws_bytes": (_i64, [_i32, _i32]),
    "ctx_geglu_f16": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "ctx_profile_begin": (_i32, []),
    "ctx_profile_end": (_i32, [_i32, _vp, _vp]),
    "ctx_bench_gemm": (C.c_float, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "ctx_gemm_tune": (None, [_i32, _i32]),
    "ctx_probe_mfma": (_i32, [_i32, _vp, _vp, _vp, _vp]),
    "ctx_probe_stage": (C.c_float, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ctx_cfg_plms_step": (_i32, [_vp, _i64, _f32, _vp, _i32, _vp, _f32, _f32, _i32, _vp, _vp, _vp]),
}


class CtxError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared object (no GPU needed to load; compute calls need one)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CtxError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if not hasattr(lib, name) and os.environ.get("CTX_ALLOW_PARTIAL") == "1":
                continue                  # bring-up only; the ABI test runs without this
            fn = getattr(lib, name)       # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise CtxError(f"libctxnerf error {rc}: {load().ctx_last_error().decode()}")


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise CtxError(f"{name}: expected a device tensor (the HIP path has no CPU fallback), got "
                       f"{type(t).__name__} on {getattr(t, 'device', '?')}")
    if not t.is_contiguous():
        raise CtxError(f"{name}: tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise CtxError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def f32c(t, device=None):
    """Contiguous float32 device copy/view."""
    t = t.to(dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    return t.contiguous()

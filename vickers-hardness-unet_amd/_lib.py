"""ctypes binding of libvkunet.so (the C ABI declared in include/vk_unet.h).

There is no CPU / eager fallback anywhere in this package: if the shared library is missing or does
not load, importing the compute entry points raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["VK_LIB"]) if os.environ.get("VK_LIB") else PKG_DIR / "libvkunet.so"   # VK_LIB: diagnostic builds
CSRC = PKG_DIR / "csrc"

VK_F32, VK_BF16, VK_F16 = 0, 1, 2


class VkError(RuntimeError):
    pass


class vk_src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int), ("up", C.c_int), ("scale", C.c_void_p),
                ("shift", C.c_void_p), ("relu", C.c_int)]


class vk_conv_desc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Ho", C.c_int),
                ("Wo", C.c_int), ("K", C.c_int), ("R", C.c_int), ("S", C.c_int), ("stride", C.c_int),
                ("pad", C.c_int), ("transposed", C.c_int), ("src0", vk_src), ("src1", vk_src)]


class vk_bnr(C.Structure):
    _fields_ = [("z", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("sums", C.c_void_p), ("mask", C.c_void_p),
                ("accumulate", C.c_int)]


class vk_letterbox_desc(C.Structure):
    _fields_ = [("h", C.c_int), ("w", C.c_int), ("src_stride", C.c_int), ("size", C.c_int), ("nh", C.c_int),
                ("nw", C.c_int), ("top", C.c_int), ("left", C.c_int), ("pad_value", C.c_int)]


class vk_geom_desc(C.Structure):
    _fields_ = [("h", C.c_int), ("w", C.c_int), ("bin_thresh", C.c_float), ("morph_kernel", C.c_int), ("open_iter", C.c_int),
                ("close_iter", C.c_int), ("min_area", C.c_int), ("max_components", C.c_int)]


class vk_geom_det(C.Structure):
    _fields_ = [("label", C.c_int), ("area", C.c_int), ("box", C.c_int * 8), ("cx", C.c_float), ("cy", C.c_float),
                ("rw", C.c_float), ("rh", C.c_float), ("ux", C.c_float), ("uy", C.c_float), ("hull_n", C.c_int),
                ("reserved", C.c_int), ("d1", C.c_double), ("d2", C.c_double), ("d_mean", C.c_double)]


class vk_geom_quad(C.Structure):
    _fields_ = [("label", C.c_int), ("area", C.c_int), ("box", C.c_int * 8), ("cx", C.c_float), ("cy", C.c_float),
                ("valid", C.c_int), ("branch", C.c_int), ("n_candidates", C.c_int), ("contour_n", C.c_int), ("hull_n", C.c_int),
                ("flags", C.c_int), ("quality", C.c_double), ("d1", C.c_double), ("d2", C.c_double), ("d_mean", C.c_double)]


class vk_aug_params(C.Structure):
    _fields_ = [("d4", C.c_int), ("rotate", C.c_int), ("cos_a", C.c_float), ("sin_a", C.c_float), ("photo", C.c_int),
                ("alpha", C.c_float), ("beta", C.c_float), ("blur_ksize", C.c_int), ("noise_scale", C.c_float),
                ("noise_seed", C.c_uint32), ("clahe_limit", C.c_int)]


class vk_unet_config(C.Structure):
    _fields_ = [("N", C.c_int), ("size", C.c_int), ("dtype", C.c_int), ("training", C.c_int), ("width", C.c_int)]


class vk_tensor_info(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("kind", C.c_int), ("dims", C.c_int * 4), ("ndim", C.c_int),
                ("offset", C.c_int64), ("numel", C.c_int64)]


vp, ci, cf, cd, sz, i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_int64
P = C.POINTER

# name -> (restype, argtypes); every symbol include/vk_unet.h declares
SIGNATURES = {
    "vk_version": (ci, []),
    "vk_last_error_string": (C.c_char_p, []),
    "vk_has_gfx950_code": (ci, []),
    "vk_probe_mfma_rate": (ci, [ci, ci, vp, P(C.c_double), vp]),
    "vk_set_reserved_cus": (ci, [ci]),
    "vk_debug_hold_cus": (ci, [ci, ci, ci, ci, vp, sz, vp, vp]),
    "vk_debug_set_stamp_buffer": (ci, [vp]),
    "vk_prof_enable": (ci, [ci]),
    "vk_prof_collect": (ci, [C.c_char_p, sz]),
    "vk_conv_fwd": (ci, [P(vk_conv_desc), vp, vp, vp, ci, ci, vp, vp]),
    "vk_conv_dgrad_pool2": (ci, [P(vk_conv_desc), vp, vp, vp, ci, ci, vp]),
    "vk_conv_dgrad_fused": (ci, [P(vk_conv_desc), vp, vp, vp, ci, ci, P(vk_bnr), vp]),
    "vk_head_bwd_fused": (ci, [ci, ci, ci, ci, P(vk_src), vp, vp, vp, vp, vp, P(vk_bnr), vp, sz, vp]),
    "vk_halo_pack": (ci, [ci, ci, ci, vp, vp, vp]),
    "vk_conv_uses_halo_pack": (ci, [P(vk_conv_desc)]),
    "vk_conv_fwd_packed": (ci, [P(vk_conv_desc), vp, vp, vp, ci, ci, vp, vp]),
    "vk_conv_fwd_splitk": (ci, [P(vk_conv_desc), vp, vp, vp, sz, vp]),
    "vk_stem_fwd": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp]),
    "vk_conv_wgrad": (ci, [P(vk_conv_desc), vp, vp, vp, sz, vp]),
    "vk_conv_wgrad_batch_supports": (ci, [P(vk_conv_desc)]),
    "vk_conv_wgrad_batch": (ci, [P(vk_conv_desc), P(vp), P(vp), ci, ci, vp, sz, vp, sz, vp]),
    "vk_stem_wgrad": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]),
    "vk_stem_wgrad_bn": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, sz, vp]),
    "vk_letterbox_preprocess": (ci, [P(vk_letterbox_desc), vp, vp, vp]),
    "vk_letterbox_postprocess_mask": (ci, [P(vk_letterbox_desc), vp, cf, vp, vp]),
    "vk_letterbox_postprocess_prob": (ci, [P(vk_letterbox_desc), vp, vp, vp]),
    "vk_geom_workspace_bytes": (i64, [P(vk_geom_desc), ci]),
    "vk_geom_minarearect": (ci, [P(vk_geom_desc), ci, vp, vp, vp, vp, vp, sz, vp]),
    "vk_geom_quadrilateral": (ci, [P(vk_geom_desc), ci, ci, vp, vp, vp, vp, vp, sz, vp]),
    "vk_letterbox_u8": (ci, [P(vk_letterbox_desc), vp, vp, vp]),
    "vk_letterbox_mask_u8": (ci, [P(vk_letterbox_desc), vp, vp, vp]),
    "vk_augment_workspace_bytes": (C.c_size_t, [ci, ci]),
    "vk_augment_batch": (ci, [ci, ci, ci, vp, vp, vp, P(vk_aug_params), vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "vk_input_transform": (ci, [ci, ci, ci, ci, vp, vp, vp]),
    "vk_bn_finalize": (ci, [ci, ci, vp, cd, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp]),
    "vk_bn_relu_maxpool": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]),
    "vk_maxpool_bwd": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    "vk_maxpool_bwd_bn_reduce": (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
    "vk_bn_add_relu": (ci, [ci, sz, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
    "vk_bn_bwd_reduce": (ci, [ci, sz, ci, vp, vp, ci, vp, vp, vp, vp, vp]),
    "vk_bn_bwd_coeffs": (ci, [ci, vp, cd, vp, vp, vp, vp, vp, vp, vp]),
    "vk_bn_bwd_apply": (ci, [ci, sz, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp]),
    "vk_bn_bwd_apply_fused": (ci, [ci, sz, ci, vp, vp, ci, vp, vp, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, ci, vp]),
    "vk_upsample2x_bwd": (ci, [ci, ci, ci, ci, ci, vp, vp, ci, vp]),
    "vk_head_fwd": (ci, [ci, ci, ci, ci, P(vk_src), vp, vp, vp, vp]),
    "vk_dec4_tail_eval": (ci, [ci, ci, ci, ci, P(vk_src), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "vk_head_bwd": (ci, [ci, ci, ci, ci, P(vk_src), vp, vp, vp, vp, vp, vp, sz, vp]),
    "vk_bce_dice_loss": (ci, [sz, vp, vp, vp, vp, vp, cf, cf, cf, vp]),
    "vk_seg_metrics_workspace_bytes": (C.c_size_t, [ci]),
    "vk_seg_metrics": (ci, [ci, sz, vp, vp, ci, cf, cf, vp, sz, vp, vp]),
    "vk_adamw_step": (ci, [sz, vp, vp, vp, vp, cf, cf, cf, cf, cf, ci, cf, vp, vp, ci, vp]),
    "vk_amp_check_inf": (ci, [sz, vp, vp, vp]),
    "vk_amp_unscale_check": (ci, [sz, vp, vp, vp, vp]),
    "vk_adamw_step_amp": (ci, [sz, vp, vp, vp, vp, cf, cf, cf, cf, cf, vp, cf, vp, vp, vp, vp, ci, vp]),
    "vk_unet_create": (ci, [P(vk_unet_config), P(vp)]),
    "vk_unet_destroy": (None, [vp]),
    "vk_unet_set_side_stream": (ci, [vp, ci]),
    "vk_unet_num_tensors": (ci, [vp]),
    "vk_unet_tensor_info": (ci, [vp, ci, P(vk_tensor_info)]),
    "vk_unet_param_numel": (i64, [vp]),
    "vk_unet_buffer_numel": (i64, [vp]),
    "vk_unet_workspace_bytes": (i64, [vp]),
    "vk_unet_num_buckets": (ci, [vp]),
    "vk_unet_bucket_range": (ci, [vp, ci, P(i64), P(i64)]),
    "vk_unet_bind": (ci, [vp, vp, vp, vp, vp, vp, sz]),
    "vk_unet_refresh_weights": (ci, [vp, vp]),
    "vk_unet_forward": (ci, [vp, vp, vp, ci, vp]),
    "vk_unet_loss": (ci, [vp, vp, vp, vp, cf, cf, cf, vp]),
    "vk_unet_backward": (ci, [vp, vp, ci, ci, vp]),
    "vk_unet_zero_grad": (ci, [vp, vp]),
    "vk_unet_debug_tensor": (ci, [vp, C.c_char_p, P(vp), P(ci * 4)]),
    "vk_comm_unique_id": (ci, [vp]),
    "vk_comm_init": (ci, [ci, ci, vp, P(vp)]),
    "vk_comm_world": (ci, [vp]),
    "vk_allreduce_bucket": (ci, [vp, vp, sz, vp]),
    "vk_comm_broadcast": (ci, [vp, vp, sz, ci, vp]),
    "vk_comm_destroy": (ci, [vp]),
}

_lib = None


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 into libvkunet.so (in-tree, next to this file)."""
    srcs = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "vk_unet.h"]
    if not force and LIB_PATH.exists() and all(LIB_PATH.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not Path(hipcc).exists():
        raise VkError(f"hipcc not found at {hipcc}; cannot build libvkunet.so")
    cmd = ["make", "-C", str(CSRC), f"-j{min(4, os.cpu_count() or 1)}", f"HIPCC={hipcc}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose:
        print(r.stdout[-2000:], r.stderr[-2000:])
    if r.returncode != 0 or not LIB_PATH.exists():
        raise VkError("building libvkunet.so failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    return LIB_PATH


def lib():
    """The loaded library with typed entry points.  Raises VkError when it is absent: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise VkError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                      f"(make -C {CSRC}); this package has no CPU fallback")
    try:
        L = C.CDLL(str(LIB_PATH))
    except OSError as e:  # pragma: no cover
        raise VkError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is None:
            raise VkError(f"{LIB_PATH} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if L.vk_version() != 1:
        raise VkError("libvkunet.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().vk_last_error_string().decode(errors="replace")
        kind = "argument/state error" if rc < 0 else f"hipError {rc}"
        raise VkError(f"{what or 'libvkunet'}: {kind}: {msg}")


def dtype_code(dt) -> int:
    import torch
    return {torch.float32: VK_F32, torch.bfloat16: VK_BF16, torch.float16: VK_F16}[dt]


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def prof_collect():
    """Aggregate per-kernel-family timings recorded since vk_prof_enable(1): {tag: dict(n, ms, flops, bytes)}."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().vk_prof_collect(buf, len(buf))
    out = {}
    for line in buf.raw[:max(n, 0)].decode().splitlines():
        tag, cnt, ms, fl, by = line.split()
        out[tag] = dict(n=int(cnt), ms=float(ms), flops=float(fl), bytes=float(by))
    return out

"""ctypes binding of the C-ABI shared library (include/simplenerf_hip.h, include/simplenerf_train.h).

The HIP library is the only implementation of the path: if it is missing or a call fails, a RuntimeError is raised --
there is no PyTorch or CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_uint, c_ulonglong,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsimplenerf_hip.so')
ABI_VERSION = 9


ITERATION_WORDS = 8      # struct snerf_iteration as 64-bit words (include/simplenerf_train.h)


class MlpDesc(ctypes.Structure):
    """struct snerf_mlp_desc"""
    _fields_ = [(name, c_int) for name in (
        'points_net_depth', 'points_net_width', 'views_net_depth', 'views_net_width', 'points_pe_degree',
        'views_pe_degree', 'sigma_pe_degree', 'use_view_dirs', 'view_dependent_rgb', 'predict_visibility')]


class LossTerm(ctypes.Structure):
    """struct snerf_loss_term"""
    _fields_ = [('pred', c_void_p), ('target', c_void_p), ('numerator_mask', c_void_p), ('denominator_mask', c_void_p),
                ('d_pred', c_void_p), ('channels', c_int), ('group', c_int), ('accumulate', c_int), ('weight', c_float)]


class Batch(ctypes.Structure):
    """struct snerf_batch"""
    _fields_ = [(name, c_void_p) for name in (
        'rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc', 'pixel_id', 'target_rgb', 'near', 'far', 'near_ndc',
        'far_ndc', 'sparse_depth_values', 'sparse_depth_errors', 'sparse_depth_values_ndc', 'mask_pixel_rays',
        'mask_sparse_rays', 'global_rows')]


RENDER_LEVELS = 6


class RenderConfig(ctypes.Structure):
    """struct snerf_render_config"""
    _fields_ = [(name, c_int) for name in ('ndc', 'white_bkgd', 'lindisp', 'num_coarse', 'num_fine', 'precision',
                                           'keep_activations', 'fused')]


class RenderMlp(ctypes.Structure):
    """struct snerf_render_mlp"""
    _fields_ = [('desc', POINTER(MlpDesc)), ('packed', c_void_p)]


class RenderRays(ctypes.Structure):
    """struct snerf_render_rays"""
    _fields_ = [(name, c_void_p) for name in ('rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc', 'near', 'far',
                                              't_rand', 'u')] + \
               [('sigma_noise', c_void_p * RENDER_LEVELS), ('depths_fine', c_void_p), ('rays_o2', c_void_p), ('num_other', c_int)]


LEVEL_OUT_FIELDS = ('rgb', 'acc', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc', 'alpha', 'visibility', 'weights',
                    'sigma', 'raw_rgb', 'saved_acts', 'raw_visibility', 'raw_visibility2', 'visibility2', 'view_dirs2')


class RenderLevelOut(ctypes.Structure):
    """struct snerf_render_level_out"""
    _fields_ = [(name, c_void_p) for name in LEVEL_OUT_FIELDS]


class RenderOutputs(ctypes.Structure):
    """struct snerf_render_outputs"""
    _fields_ = [('depths_coarse', c_void_p), ('depths_fine', c_void_p), ('level', RenderLevelOut * RENDER_LEVELS)]


class RenderLevelGrads(ctypes.Structure):
    """struct snerf_render_level_grads"""
    _fields_ = [(name, c_void_p) for name in ('rgb', 'acc', 'depth', 'depth_ndc', 'sigma', 'raw_rgb')] + \
               [('param_grads', POINTER(c_void_p)), ('num_params', c_int), ('accumulate', c_int)]


CAMERA_FLOATS = 24
LOSS_MAX_TERMS = 16
LOSS_MAX_GROUPS = 16

_FP = c_void_p  # device (or host, where the header says so) float*

SIGNATURES = {
    'snerf_abi_version': (c_int, []),
    'snerf_last_error': (c_char_p, []),
    'snerf_generate_rays': (c_int, [c_int, c_int, POINTER(c_float), POINTER(c_float), c_float, c_int, c_float,
                                    c_longlong, c_longlong, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    'snerf_coarse_depths': (c_int, [_FP, _FP, c_longlong, c_int, c_int, _FP, _FP, c_void_p]),
    'snerf_mlp_num_params': (c_int, [POINTER(MlpDesc)]),
    'snerf_mlp_packed_floats': (c_size_t, [POINTER(MlpDesc)]),
    'snerf_mlp_pack': (c_int, [POINTER(MlpDesc), POINTER(c_void_p), c_int, _FP, c_void_p]),
    'snerf_mlp_pack_for': (c_int, [POINTER(MlpDesc), POINTER(c_void_p), c_int, _FP, c_int, c_int, c_void_p]),
    'snerf_mlp_forward': (c_int, [POINTER(MlpDesc), _FP, _FP, _FP, _FP, _FP, c_longlong, c_int, _FP, _FP, _FP, c_int,
                                  c_void_p]),
    'snerf_mlp_saved_floats': (c_size_t, [POINTER(MlpDesc), c_longlong, c_int]),
    'snerf_mlp_forward_train': (c_int, [POINTER(MlpDesc), _FP, _FP, _FP, _FP, _FP, c_longlong, c_int, _FP, _FP, _FP, _FP,
                                        c_int, c_void_p]),
    'snerf_mlp_backward_workspace_floats': (c_size_t, [POINTER(MlpDesc), c_longlong, c_int]),
    'snerf_mlp_backward': (c_int, [POINTER(MlpDesc), _FP, _FP, _FP, _FP, _FP, _FP, c_longlong, c_int, _FP,
                                   POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    'snerf_other_view_dirs': (c_int, [_FP, _FP, _FP, _FP, c_longlong, c_int, c_int, c_int, _FP, c_void_p]),
    'snerf_mlp_forward_visibility': (c_int, [POINTER(MlpDesc), _FP, _FP, _FP, _FP, _FP, c_longlong, c_int, _FP, _FP, c_int, _FP,
                                             _FP, _FP, _FP, _FP, c_int, c_void_p]),
    'snerf_composite_visibility2': (c_int, [_FP, _FP, _FP, c_longlong, c_int, c_int, _FP, c_void_p]),
    'snerf_profile_enable': (c_int, [c_int]),
    'snerf_profile_collect': (c_int, [c_int, POINTER(c_float), POINTER(c_longlong), c_int]),
    'snerf_profile_reset': (c_int, []),
    'snerf_profile_dropped': (c_longlong, []),
    'snerf_range_status': (c_int, [c_int]),
    'snerf_render_workspace_floats': (c_size_t, [POINTER(RenderConfig), c_longlong]),
    'snerf_render_forward': (c_int, [POINTER(RenderConfig), POINTER(RenderMlp), POINTER(RenderRays), c_longlong,
                                     POINTER(RenderOutputs), _FP, c_void_p]),
    'snerf_render_backward_workspace_floats': (c_size_t, [POINTER(RenderConfig), POINTER(RenderMlp), c_longlong]),
    'snerf_render_backward': (c_int, [POINTER(RenderConfig), POINTER(RenderMlp), POINTER(RenderRays), c_longlong,
                                      POINTER(RenderOutputs), POINTER(RenderLevelGrads), _FP, c_void_p]),
    'snerf_composite_backward': (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_longlong, c_int, c_int, c_int, _FP, _FP, _FP, _FP,
                                         _FP, _FP, c_void_p]),
    'snerf_composite': (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_longlong, c_int, c_int, c_int, _FP, _FP, _FP, _FP, _FP,
                                _FP, _FP, _FP, _FP, c_void_p]),
    'snerf_to_display': (c_int, [_FP, _FP, c_longlong, _FP, _FP, c_void_p]),
    'snerf_resample_depths': (c_int, [_FP, _FP, c_longlong, c_int, c_int, _FP, _FP, c_void_p]),
    # include/simplenerf_train.h
    'snerf_loss_workspace_bytes': (c_longlong, []),
    'snerf_loss_forward': (c_int, [POINTER(LossTerm), c_int, c_int, c_longlong, _FP, _FP, c_void_p, c_void_p]),
    'snerf_loss_backward': (c_int, [POINTER(LossTerm), c_int, c_int, c_longlong, _FP, _FP, c_void_p]),
    'snerf_patch_consistency_masks': (c_int, [_FP, _FP, _FP, _FP, c_void_p, c_void_p, c_longlong, _FP, _FP, _FP, c_int,
                                              c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, _FP, _FP,
                                              c_void_p]),
    'snerf_camera_table': (c_int, [_FP, _FP, c_int, c_int, c_int, _FP, c_void_p]),
    'snerf_assemble_batch': (c_int, [c_void_p, c_longlong, c_longlong, _FP, c_int, c_int, c_int, _FP, _FP, _FP, _FP, c_int,
                                     c_float, c_float, c_float, c_float, c_longlong, c_longlong, POINTER(Batch), c_void_p]),
    'snerf_shuffled_indices': (c_int, [c_ulonglong, c_ulonglong, c_longlong, c_longlong, c_longlong, c_void_p, c_int,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'snerf_random_uniform': (c_int, [c_ulonglong, c_uint, c_longlong, c_void_p, c_longlong, c_int, _FP, c_void_p]),
    'snerf_random_normal': (c_int, [c_ulonglong, c_uint, c_longlong, c_void_p, c_longlong, c_int, c_float, _FP, c_void_p]),
    'snerf_adam_step': (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                POINTER(c_longlong), c_int, c_longlong, c_double, c_double, c_double, c_double,
                                c_void_p]),
    # per-iteration scalars in device memory (whole-iteration HIP graphs)
    'snerf_iteration_advance': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'snerf_shuffled_indices_at': (c_int, [c_ulonglong, c_void_p, c_int, c_longlong, c_longlong, c_longlong, c_void_p, c_int,
                                          c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'snerf_random_uniform_at': (c_int, [c_ulonglong, c_void_p, c_int, c_int, c_longlong, c_void_p, c_longlong, c_int, _FP, c_void_p]),
    'snerf_random_normal_at': (c_int, [c_ulonglong, c_void_p, c_int, c_int, c_longlong, c_void_p, c_longlong, c_int, c_float, _FP,
                                       c_void_p]),
    'snerf_adam_step_at': (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                   POINTER(c_longlong), c_int, c_void_p, c_double, c_double, c_double, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: the HIP renderer has not been built. Run `python -m simplenerf_amd.build` '
            f'(needs hipcc; cross-compiles for gfx950 without a GPU). There is no fallback implementation.')
    # The process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as
    # /opt/rocm's); importing torch first makes the dynamic linker bind this library to that already-loaded copy, so
    # device pointers, streams and the primary context are shared with torch's allocator.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.snerf_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f'{LIB_PATH}: ABI version {got}, binding expects {ABI_VERSION}; rebuild the library')
    _lib = lib
    return lib


class Fp16RangeError(RuntimeError):
    """SNERF_E_RANGE: an earlier fp16-mode launch on this device met an activation, encoded input or weight outside the fp16
    range (|v| > 65504); its outputs are invalid.  Reported once, by the next fp16-mode call (include/simplenerf_hip.h)."""


def check(status: int, what: str) -> None:
    if status == -4:
        msg = load().snerf_last_error()
        raise Fp16RangeError(msg.decode() if msg else f'{what}: fp16 range exceeded')
    if status != 0:
        msg = load().snerf_last_error()
        raise RuntimeError(f'{what} failed ({status}): {msg.decode() if msg else "unknown error"}')

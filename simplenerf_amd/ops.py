"""Tensor-level wrappers over the C ABI.  PyTorch supplies device memory and the current HIP stream; every
computation happens in the HIP library.  Inputs must be CUDA(HIP) float32 tensors; outputs are allocated here."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import numpy
import torch

from . import _lib
from ._lib import MlpDesc

Tensor = torch.Tensor
PRECISION_FP32 = 0   # fp32 MFMA
PRECISION_F16X3 = 1  # fp16 hi/lo split, 3 MFMAs per product, fp32 accumulate (forward only)
PRECISIONS = {'fp32': PRECISION_FP32, 'f16x3': PRECISION_F16X3}


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: Optional[Tensor], name: str, shape=None) -> Optional[Tensor]:
    """Validate a device operand; returns a contiguous fp32 view (copying only if the caller's tensor is strided)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f'{name}: expected a tensor on the GPU (the HIP renderer has no CPU path), got '
                           f'{type(t).__name__}{"" if not isinstance(t, torch.Tensor) else " on " + str(t.device)}')
    if t.dtype != torch.float32:
        raise RuntimeError(f'{name}: expected float32, got {t.dtype}')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f'{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
    return t.detach().contiguous()


def _ptr(t: Optional[Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# ---------------------------------------------------------------------------------------------- K1
def generate_rays(resolution, intrinsic, pose, near: float, ndc: bool, device, first_ray: int = 0,
                  num_rays: Optional[int] = None, pixel_offset: float = 0.0) -> Dict[str, Tensor]:
    """rays_o, rays_d, view_dirs (+ rays_o_ndc, rays_d_ndc) for pixels [first_ray, first_ray+num_rays) of a frame.
    intrinsic (3,3) and pose (4,4, processed camera-to-world) are host arrays."""
    lib = _lib.load()
    h, w = int(resolution[0]), int(resolution[1])
    if num_rays is None:
        num_rays = h * w - first_ray
    k = numpy.ascontiguousarray(numpy.asarray(intrinsic, dtype=numpy.float32).reshape(3, 3))
    p = numpy.ascontiguousarray(numpy.asarray(pose, dtype=numpy.float32).reshape(4, 4))
    out = {name: torch.empty((num_rays, 3), dtype=torch.float32, device=device)
           for name in (('rays_o', 'rays_d', 'view_dirs') + (('rays_o_ndc', 'rays_d_ndc') if ndc else ()))}
    if num_rays == 0:
        return out
    with torch.cuda.device(out['rays_o'].device):
        st = lib.snerf_generate_rays(h, w, k.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                     p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), float(pixel_offset), int(ndc),
                                     float(near), int(first_ray), int(num_rays), _ptr(out['rays_o']), _ptr(out['rays_d']),
                                     _ptr(out['view_dirs']), _ptr(out.get('rays_o_ndc')), _ptr(out.get('rays_d_ndc')),
                                     _stream())
    _lib.check(st, 'snerf_generate_rays')
    return out


# ---------------------------------------------------------------------------------------------- K2
def coarse_depths(near: Tensor, far: Tensor, num_samples: int, lindisp: bool = False,
                  t_rand: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    n = near.shape[0]
    near = _dev(near.reshape(-1), 'near', (n,))
    far = _dev(far.reshape(-1), 'far', (n,))
    t_rand = _dev(t_rand, 't_rand', (n, num_samples))
    out = torch.empty((n, num_samples), dtype=torch.float32, device=near.device)
    if n == 0:
        return out
    with torch.cuda.device(near.device):
        st = lib.snerf_coarse_depths(_ptr(near), _ptr(far), n, int(num_samples), int(bool(lindisp)), _ptr(t_rand),
                                     _ptr(out), _stream())
    _lib.check(st, 'snerf_coarse_depths')
    return out


# ---------------------------------------------------------------------------------------------- K3
def mlp_desc(mlp_cfg: dict) -> MlpDesc:
    """snerf_mlp_desc from a reference-style per-MLP config dict (keys of src/models/SimpleNeRF01.py:567-584)."""
    if mlp_cfg.get('predict_visibility', False):
        raise NotImplementedError('predict_visibility is off in every shipped configuration and is not built')
    return MlpDesc(
        points_net_depth=int(mlp_cfg['points_net_depth']), points_net_width=int(mlp_cfg['points_net_width']),
        views_net_depth=int(mlp_cfg.get('views_net_depth', 1)), views_net_width=int(mlp_cfg.get('views_net_width', 0)),
        points_pe_degree=int(mlp_cfg['points_positional_encoding_degree']),
        views_pe_degree=int(mlp_cfg.get('views_positional_encoding_degree', 0)),
        sigma_pe_degree=int(mlp_cfg.get('points_sigma_positional_encoding_degree', -1)),
        use_view_dirs=int(bool(mlp_cfg['use_view_dirs'])), view_dependent_rgb=int(bool(mlp_cfg['view_dependent_rgb'])))


class PackedMlp:
    """Device-resident packed weight stream of one MLP (see csrc/mlp_layout.h)."""

    # When set to a list, every forward() brackets its kernel launch with a pair of events recorded on the launch
    # stream and appends (start, end, num_samples) -- bench.py uses this to time the dominant kernel in situ.
    event_log = None

    def __init__(self, mlp_cfg: dict, device):
        lib = _lib.load()
        self.desc = mlp_desc(mlp_cfg)
        self.num_params = lib.snerf_mlp_num_params(ctypes.byref(self.desc))
        floats = lib.snerf_mlp_packed_floats(ctypes.byref(self.desc))
        if self.num_params == 0 or floats == 0:
            msg = lib.snerf_last_error()
            raise NotImplementedError(f'MLP configuration not built for the HIP path: {msg.decode() if msg else mlp_cfg}')
        self.buffer = torch.zeros(floats, dtype=torch.float32, device=device)
        self.use_view_dirs = bool(self.desc.view_dependent_rgb)

    def pack(self, params: List[Tensor]) -> None:
        """params in C-ABI order: pts_linears.{i}.weight/.bias ..., pts_output_linear.*, [feature_linear.*,
        views_linears.0.*, views_output_linear.*]."""
        lib = _lib.load()
        if len(params) != self.num_params:
            raise RuntimeError(f'expected {self.num_params} parameter tensors, got {len(params)}')
        held = [_dev(p, f'param[{i}]') for i, p in enumerate(params)]
        arr = (ctypes.c_void_p * len(held))(*[p.data_ptr() for p in held])
        with torch.cuda.device(self.buffer.device):
            st = lib.snerf_mlp_pack(ctypes.byref(self.desc), arr, len(held), _ptr(self.buffer), _stream())
        _lib.check(st, 'snerf_mlp_pack')

    def forward(self, origins: Tensor, dirs: Tensor, view_dirs: Optional[Tensor], depths: Tensor,
                sigma_noise: Optional[Tensor] = None, precision: int = 0):
        """-> sigma (n,S,1), rgb (n,S,3).  precision: PRECISION_FP32 (0) or PRECISION_F16X3 (1), see the C header."""
        lib = _lib.load()
        n, s = depths.shape
        origins = _dev(origins, 'origins', (n, 3))
        dirs = _dev(dirs, 'dirs', (n, 3))
        depths = _dev(depths, 'depths')
        view_dirs = _dev(view_dirs, 'view_dirs', (n, 3)) if self.use_view_dirs else None
        if self.use_view_dirs and view_dirs is None:
            raise KeyError('view_dirs')
        if sigma_noise is not None:
            sigma_noise = _dev(sigma_noise.reshape(n, s), 'sigma_noise', (n, s))
        sigma = torch.empty((n, s, 1), dtype=torch.float32, device=depths.device)
        rgb = torch.empty((n, s, 3), dtype=torch.float32, device=depths.device)
        if n == 0:
            return sigma, rgb
        log = PackedMlp.event_log
        with torch.cuda.device(depths.device):
            if log is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
            st = lib.snerf_mlp_forward(ctypes.byref(self.desc), _ptr(self.buffer), _ptr(origins), _ptr(dirs),
                                       _ptr(view_dirs), _ptr(depths), n, s, _ptr(sigma_noise), _ptr(sigma), _ptr(rgb),
                                       int(precision), _stream())
            if log is not None:
                t1.record()
                log.append((t0, t1, n * s))
        _lib.check(st, 'snerf_mlp_forward')
        return sigma, rgb


    # ---- training ------------------------------------------------------------------------------------------
    def forward_train(self, origins: Tensor, dirs: Tensor, view_dirs: Optional[Tensor], depths: Tensor,
                      sigma_noise: Optional[Tensor] = None, precision: int = 0):
        """Forward that also keeps every layer's input for backward().  -> sigma (n,S,1), rgb (n,S,3), saved"""
        lib = _lib.load()
        n, s = depths.shape
        origins = _dev(origins, 'origins', (n, 3))
        dirs = _dev(dirs, 'dirs', (n, 3))
        depths = _dev(depths, 'depths')
        view_dirs = _dev(view_dirs, 'view_dirs', (n, 3)) if self.use_view_dirs else None
        if self.use_view_dirs and view_dirs is None:
            raise KeyError('view_dirs')
        if sigma_noise is not None:
            sigma_noise = _dev(sigma_noise.reshape(n, s), 'sigma_noise', (n, s))
        dev = depths.device
        sigma = torch.empty((n, s, 1), dtype=torch.float32, device=dev)
        rgb = torch.empty((n, s, 3), dtype=torch.float32, device=dev)
        saved = torch.empty(lib.snerf_mlp_saved_floats(ctypes.byref(self.desc), n, s), dtype=torch.float32, device=dev)
        if n == 0:
            return sigma, rgb, saved
        log = PackedMlp.event_log
        with torch.cuda.device(dev):
            if log is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
            st = lib.snerf_mlp_forward_train(ctypes.byref(self.desc), _ptr(self.buffer), _ptr(origins), _ptr(dirs),
                                             _ptr(view_dirs), _ptr(depths), n, s, _ptr(sigma_noise), _ptr(sigma), _ptr(rgb),
                                             _ptr(saved), int(precision), _stream())
            if log is not None:
                t1.record()
                log.append((t0, t1, n * s))
        _lib.check(st, 'snerf_mlp_forward_train')
        return sigma, rgb, saved

    def backward(self, saved: Tensor, sigma: Tensor, rgb: Tensor, d_sigma: Tensor, d_rgb: Tensor,
                 param_shapes: List[tuple], precision: int = 0) -> List[Tensor]:
        """dL/dparam for every parameter (C-ABI order), given dL/dsigma (n,S[,1]) and dL/drgb (n,S,3)."""
        lib = _lib.load()
        n, s = sigma.shape[0], sigma.shape[1]
        dev = sigma.device
        sigma = _dev(sigma.reshape(n, s), 'sigma', (n, s))
        rgb = _dev(rgb, 'rgb', (n, s, 3))
        d_sigma = _dev(d_sigma.reshape(n, s), 'd_sigma', (n, s))
        d_rgb = _dev(d_rgb, 'd_rgb', (n, s, 3))
        if len(param_shapes) != self.num_params:
            raise RuntimeError(f'expected {self.num_params} parameter shapes, got {len(param_shapes)}')
        if n == 0:
            return [torch.zeros(tuple(shape), dtype=torch.float32, device=dev) for shape in param_shapes]
        grads = [torch.empty(tuple(shape), dtype=torch.float32, device=dev) for shape in param_shapes]
        work = torch.empty(lib.snerf_mlp_backward_workspace_floats(ctypes.byref(self.desc), n, s), dtype=torch.float32,
                           device=dev)
        arr = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        with torch.cuda.device(dev):
            st = lib.snerf_mlp_backward(ctypes.byref(self.desc), _ptr(self.buffer), _ptr(saved), _ptr(sigma), _ptr(rgb),
                                        _ptr(d_sigma), _ptr(d_rgb), n, s, _ptr(work), arr, len(grads), int(precision),
                                        _stream())
        _lib.check(st, 'snerf_mlp_backward')
        return grads


# ---------------------------------------------------------------------------------------------- K4
def composite(sigma: Tensor, rgb: Tensor, depths: Tensor, march_dirs: Tensor, ndc: bool, white_bkgd: bool = False,
              rays_o: Optional[Tensor] = None, rays_d: Optional[Tensor] = None,
              per_sample: bool = True) -> Dict[str, Tensor]:
    """Keys as the reference's volume_rendering: rgb, acc, alpha, visibility, weights, depth, depth_var
    (+ depth_ndc, depth_var_ndc when ndc).  per_sample=False skips alpha/visibility/weights."""
    lib = _lib.load()
    n, s = depths.shape
    sigma = _dev(sigma.reshape(n, s), 'sigma', (n, s))
    rgb = _dev(rgb, 'rgb', (n, s, 3))
    depths = _dev(depths, 'depths')
    march_dirs = _dev(march_dirs, 'march_dirs', (n, 3))
    dev = depths.device
    out = {'rgb': torch.empty((n, 3), dtype=torch.float32, device=dev)}
    for k in ('acc', 'depth', 'depth_var') + (('depth_ndc', 'depth_var_ndc') if ndc else ()):
        out[k] = torch.empty((n,), dtype=torch.float32, device=dev)
    if per_sample:
        for k in ('alpha', 'visibility', 'weights'):
            out[k] = torch.empty((n, s), dtype=torch.float32, device=dev)
    if ndc:
        rays_o = _dev(rays_o, 'rays_o', (n, 3))
        rays_d = _dev(rays_d, 'rays_d', (n, 3))
    if n == 0:
        return out
    with torch.cuda.device(dev):
        st = lib.snerf_composite(_ptr(sigma), _ptr(rgb), _ptr(depths), _ptr(march_dirs), _ptr(rays_o if ndc else None),
                                 _ptr(rays_d if ndc else None), n, s, int(bool(ndc)), int(bool(white_bkgd)),
                                 _ptr(out['rgb']), _ptr(out['acc']), _ptr(out.get('alpha')), _ptr(out.get('visibility')),
                                 _ptr(out.get('weights')), _ptr(out['depth']), _ptr(out['depth_var']),
                                 _ptr(out.get('depth_ndc')), _ptr(out.get('depth_var_ndc')), _stream())
    _lib.check(st, 'snerf_composite')
    return out


# ---------------------------------------------------------------------------------------------- K6
def composite_backward(sigma: Tensor, rgb: Tensor, depths: Tensor, march_dirs: Tensor, ndc: bool, white_bkgd: bool,
                       rays_o: Optional[Tensor], rays_d: Optional[Tensor], grad_rgb: Optional[Tensor],
                       grad_acc: Optional[Tensor], grad_depth: Optional[Tensor], grad_depth_ndc: Optional[Tensor]):
    """-> d_sigma (n,S), d_rgb (n,S,3) from the per-ray gradients (None = zero)."""
    lib = _lib.load()
    n, s = depths.shape
    sigma = _dev(sigma.reshape(n, s), 'sigma', (n, s))
    rgb = _dev(rgb, 'rgb', (n, s, 3))
    depths = _dev(depths, 'depths')
    march_dirs = _dev(march_dirs, 'march_dirs', (n, 3))
    if ndc:
        rays_o = _dev(rays_o, 'rays_o', (n, 3))
        rays_d = _dev(rays_d, 'rays_d', (n, 3))
    grad_rgb = _dev(grad_rgb, 'grad_rgb', (n, 3))
    grad_acc = _dev(grad_acc, 'grad_acc', (n,))
    grad_depth = _dev(grad_depth, 'grad_depth', (n,))
    grad_depth_ndc = _dev(grad_depth_ndc, 'grad_depth_ndc', (n,))
    dev = depths.device
    d_sigma = torch.empty((n, s), dtype=torch.float32, device=dev)
    d_rgb = torch.empty((n, s, 3), dtype=torch.float32, device=dev)
    if n == 0:
        return d_sigma, d_rgb
    with torch.cuda.device(dev):
        st = lib.snerf_composite_backward(_ptr(sigma), _ptr(rgb), _ptr(depths), _ptr(march_dirs),
                                          _ptr(rays_o if ndc else None), _ptr(rays_d if ndc else None), n, s,
                                          int(bool(ndc)), int(bool(white_bkgd)), _ptr(grad_rgb), _ptr(grad_acc),
                                          _ptr(grad_depth), _ptr(grad_depth_ndc), _ptr(d_sigma), _ptr(d_rgb), _stream())
    _lib.check(st, 'snerf_composite_backward')
    return d_sigma, d_rgb


# ---------------------------------------------------------------------------------------------- K5
def resample_depths(depths_coarse: Tensor, weights_coarse: Tensor, num_fine: int, u: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    n, s_c = depths_coarse.shape
    depths_coarse = _dev(depths_coarse, 'depths_coarse')
    weights_coarse = _dev(weights_coarse, 'weights_coarse', (n, s_c))
    u = _dev(u, 'u', (n, num_fine))
    out = torch.empty((n, s_c + num_fine), dtype=torch.float32, device=depths_coarse.device)
    if n == 0:
        return out
    with torch.cuda.device(depths_coarse.device):
        st = lib.snerf_resample_depths(_ptr(depths_coarse), _ptr(weights_coarse), n, s_c, int(num_fine), _ptr(u),
                                       _ptr(out), _stream())
    _lib.check(st, 'snerf_resample_depths')
    return out


# ---------------------------------------------------------------------------------------------- f3
def to_display(rgb: Tensor, depth: Optional[Tensor] = None):
    """-> uint8 image (n,3) [clip to [0,1], round(255 x) half-to-even] and depth clipped at 0 (or None)."""
    lib = _lib.load()
    n = rgb.shape[0]
    rgb = _dev(rgb, 'rgb', (n, 3))
    depth = _dev(depth, 'depth', (n,))
    image = torch.empty((n, 3), dtype=torch.uint8, device=rgb.device)
    depth_out = None if depth is None else torch.empty((n,), dtype=torch.float32, device=rgb.device)
    if n == 0:
        return image, depth_out
    with torch.cuda.device(rgb.device):
        st = lib.snerf_to_display(_ptr(rgb), _ptr(depth), n, ctypes.c_void_p(image.data_ptr()), _ptr(depth_out), _stream())
    _lib.check(st, 'snerf_to_display')
    return image, depth_out

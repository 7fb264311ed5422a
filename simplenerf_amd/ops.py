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
PRECISION_F16X3 = 1  # fp16 hi/lo split, 3 MFMAs per product, fp32 accumulate: fp32-grade results
PRECISION_F16 = 2    # 16-bit mode: one fp16 MFMA per product, fp32 accumulate / master weights; 16-bit saved tensors
PRECISION_BF16 = 3   # the 16-bit mode on bf16 operands (BASELINE config 5's literal dtype): no range limit, 8 significand bits
PRECISION_F16S8 = 4  # the 16-bit (fp16) mode with the saved trunk activations as fp8 e4m3: only the weight gradients differ from 'f16'
PRECISION_BF16S8 = 5  # the bf16 mode with the saved trunk activations as fp8 e4m3: only the weight gradients differ from 'bf16'
PRECISIONS = {'fp32': PRECISION_FP32, 'f16x3': PRECISION_F16X3, 'f16': PRECISION_F16, 'bf16': PRECISION_BF16, 'f16s8': PRECISION_F16S8,
              'bf16s8': PRECISION_BF16S8}
FP16_RANGE_PRECISIONS = (PRECISION_F16X3, PRECISION_F16, PRECISION_F16S8)     # the modes whose operands must stay below 65504 (Fp16RangeError)


def _stream() -> ctypes.c_void_p:
    # raw hipStream_t of torch's current stream on the current device (the C-level getter: ~30x cheaper than building a
    # torch.cuda.Stream object, and this runs ~50 times per training iteration)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


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
    # (the common case -- a contiguous tensor outside autograd -- passes through without creating two tensor objects;
    # this function runs ~100 times per training forward)
    if t.requires_grad or not t.is_contiguous():
        return t.detach().contiguous()
    return t


def _ptr(t: Optional[Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# ---------------------------------------------------------------------------------------------- fp16 range flag
Fp16RangeError = _lib.Fp16RangeError
RANGE_ACTIVATION, RANGE_WEIGHT = 1, 2


def range_status(clear: bool = False) -> int:
    """The current device's fp16 range flag (snerf_range_status): 0 = clean, bit 0 = an activation / encoded input of an
    fp16-mode launch left the fp16 range, bit 1 = a weight did.  Launches are asynchronous: synchronise first to be sure
    every enqueued launch has reported.  (Without a query the flag surfaces as Fp16RangeError from the next fp16-mode call.)"""
    status = _lib.load().snerf_range_status(int(bool(clear)))
    if status < 0:
        _lib.check(status, 'snerf_range_status')
    return status


# ---------------------------------------------------------------------------------------------- measurement hook
PROFILE_MLP_FORWARD, PROFILE_MLP_BACKWARD = 0, 1


def profile_enable(capacity: int) -> None:
    """Start (capacity > 0) or stop (0) the library's event timing of its MLP forward launches / backward calls
    (snerf_profile_enable): the events are recorded by the library on the stream each launch is enqueued on, also for
    launches issued from inside the one-call render ops."""
    _lib.check(_lib.load().snerf_profile_enable(int(capacity)), 'snerf_profile_enable')


def profile_reset() -> None:
    _lib.check(_lib.load().snerf_profile_reset(), 'snerf_profile_reset')


def profile_dropped() -> int:
    """Launches since the last enable / reset that found the event table full and were therefore not timed."""
    return int(_lib.load().snerf_profile_dropped())


def profile_collect(kind: int, capacity: int = 65536):
    """-> (milliseconds, samples): per recorded launch of ``kind`` its duration and its number of (ray, sample) rows.
    Waits for the events."""
    ms = (ctypes.c_float * capacity)()
    samples = (ctypes.c_longlong * capacity)()
    count = _lib.load().snerf_profile_collect(int(kind), ms, samples, capacity)
    if count < 0:
        _lib.check(count, 'snerf_profile_collect')
    return list(ms[:count]), list(samples[:count])


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
    return MlpDesc(
        points_net_depth=int(mlp_cfg['points_net_depth']), points_net_width=int(mlp_cfg['points_net_width']),
        views_net_depth=int(mlp_cfg.get('views_net_depth', 1)), views_net_width=int(mlp_cfg.get('views_net_width', 0)),
        points_pe_degree=int(mlp_cfg['points_positional_encoding_degree']),
        views_pe_degree=int(mlp_cfg.get('views_positional_encoding_degree', 0)),
        sigma_pe_degree=int(mlp_cfg.get('points_sigma_positional_encoding_degree', -1)),
        use_view_dirs=int(bool(mlp_cfg['use_view_dirs'])), view_dependent_rgb=int(bool(mlp_cfg['view_dependent_rgb'])),
        predict_visibility=int(bool(mlp_cfg.get('predict_visibility', False))))


class PackedMlp:
    """Device-resident packed weight stream of one MLP (see csrc/mlp_layout.h)."""


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

    def pack(self, params: List[Tensor], precision: Optional[int] = None, training: bool = False) -> None:
        """params in C-ABI order: pts_linears.{i}.weight/.bias ..., pts_output_linear.*, [feature_linear.*,
        views_linears.0.*, views_output_linear.*].  ``precision`` None: every operand format (snerf_mlp_pack); a PRECISION_*
        value: only what calls at that precision in that mode read (snerf_mlp_pack_for) -- re-pack before using the stream
        at another precision or mode."""
        lib = _lib.load()
        if len(params) != self.num_params:
            raise RuntimeError(f'expected {self.num_params} parameter tensors, got {len(params)}')
        held = [_dev(p, f'param[{i}]') for i, p in enumerate(params)]
        arr = (ctypes.c_void_p * len(held))(*[p.data_ptr() for p in held])
        with torch.cuda.device(self.buffer.device):
            if precision is None:
                st = lib.snerf_mlp_pack(ctypes.byref(self.desc), arr, len(held), _ptr(self.buffer), _stream())
            else:
                st = lib.snerf_mlp_pack_for(ctypes.byref(self.desc), arr, len(held), _ptr(self.buffer), int(precision),
                                            1 if training else 0, _stream())
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
        with torch.cuda.device(depths.device):
            st = lib.snerf_mlp_forward(ctypes.byref(self.desc), _ptr(self.buffer), _ptr(origins), _ptr(dirs),
                                       _ptr(view_dirs), _ptr(depths), n, s, _ptr(sigma_noise), _ptr(sigma), _ptr(rgb),
                                       int(precision), _stream())
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
        with torch.cuda.device(dev):
            st = lib.snerf_mlp_forward_train(ctypes.byref(self.desc), _ptr(self.buffer), _ptr(origins), _ptr(dirs),
                                             _ptr(view_dirs), _ptr(depths), n, s, _ptr(sigma_noise), _ptr(sigma), _ptr(rgb),
                                             _ptr(saved), int(precision), _stream())
        _lib.check(st, 'snerf_mlp_forward_train')
        return sigma, rgb, saved

    def backward_workspace_floats(self, n: int, s: int) -> int:
        return int(_lib.load().snerf_mlp_backward_workspace_floats(ctypes.byref(self.desc), n, s))

    def backward(self, saved: Tensor, sigma: Tensor, rgb: Tensor, d_sigma: Tensor, d_rgb: Tensor,
                 param_shapes: List[tuple], precision: int = 0, into: Optional[List[Tensor]] = None,
                 work: Optional[Tensor] = None) -> List[Tensor]:
        """dL/dparam for every parameter (C-ABI order), given dL/dsigma (n,S[,1]) and dL/drgb (n,S,3).  ``into``: existing
        gradient tensors to ADD to (the kernel accumulates; no separate add launches) instead of fresh ones.  ``work``: the
        caller's scratch (at least ``backward_workspace_floats(n, s)`` floats) instead of a fresh allocation."""
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
            return into if into is not None else [torch.zeros(tuple(shape), dtype=torch.float32, device=dev) for shape in param_shapes]
        if into is not None:
            for g, shape in zip(into, param_shapes):
                if tuple(g.shape) != tuple(shape) or g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous():
                    raise RuntimeError(f'backward(into=...): expected a contiguous float32 GPU tensor of shape {tuple(shape)}')
        grads = into if into is not None else [torch.empty(tuple(shape), dtype=torch.float32, device=dev) for shape in param_shapes]
        need = self.backward_workspace_floats(n, s)
        if work is None:
            work = torch.empty(need, dtype=torch.float32, device=dev)
        elif work.dtype != torch.float32 or not work.is_cuda or not work.is_contiguous() or work.numel() < need:
            raise RuntimeError(f'backward(work=...): expected a contiguous float32 GPU tensor of at least {need} floats')
        arr = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        with torch.cuda.device(dev):
            st = lib.snerf_mlp_backward(ctypes.byref(self.desc), _ptr(self.buffer), _ptr(saved), _ptr(sigma), _ptr(rgb),
                                        _ptr(d_sigma), _ptr(d_rgb), n, s, _ptr(work), arr, len(grads), int(precision),
                                        int(into is not None), _stream())
        _lib.check(st, 'snerf_mlp_backward')
        return grads


# ---------------------------------------------------------------------------------------------- one-call render ops
def _row_major_strides(shape):
    strides, step = [], 1
    for d in reversed(shape):
        strides.append(step)
        step *= d
    return tuple(reversed(strides))


class RenderCall:
    """One ``snerf_render_forward`` (and, for training, its ``snerf_render_backward``): the whole of SimpleNeRF.render_rays
    (src/models/SimpleNeRF01.py:108-270) enqueued by ONE call into the library.  Output tensors are carved out of three
    allocations -- per-ray outputs, per-sample outputs, and (training) one saved-activation buffer per level -- instead
    of ~20 separate ones.

    ``mlps``: six entries (C-ABI level order: main / points-aug / views-aug at the coarse level, then at the fine level),
    ``PackedMlp`` or None.  ``rays``: the reference's input_batch tensors.  ``draws``: t_rand, u, per-level sigma noise,
    optional fine-depth override.  ``per_sample``: which (n,S) composite outputs to produce ('alpha', 'visibility',
    'weights')."""

    PER_RAY = (('rgb', 3), ('acc', 1), ('depth', 1), ('depth_var', 1), ('depth_ndc', 1), ('depth_var_ndc', 1))

    def __init__(self, mlps: List[Optional['PackedMlp']], ndc: bool, white_bkgd: bool, lindisp: bool, num_coarse: int,
                 num_fine: int, precision: int, keep_activations: bool, per_sample=('alpha',), fused: bool = False):
        lib = _lib.load()
        self.lib = lib
        self.mlps = list(mlps) + [None] * (_lib.RENDER_LEVELS - len(mlps))
        self.ndc, self.keep = bool(ndc), bool(keep_activations)
        self.num_coarse, self.num_fine = int(num_coarse), int(num_fine) if self.mlps[3] is not None else 0
        self.cfg = _lib.RenderConfig(int(bool(ndc)), int(bool(white_bkgd)), int(bool(lindisp)), self.num_coarse, self.num_fine,
                                     int(precision), int(self.keep), int(bool(fused) and not self.keep))
        self.c_mlps = (_lib.RenderMlp * _lib.RENDER_LEVELS)()
        for l, m in enumerate(self.mlps):
            if m is not None:
                self.c_mlps[l].desc = ctypes.pointer(m.desc)
                self.c_mlps[l].packed = m.buffer.data_ptr()
        self.per_sample = tuple(per_sample)
        self.levels = [l for l, m in enumerate(self.mlps) if m is not None]
        self.c_rays = _lib.RenderRays()
        self.c_out = _lib.RenderOutputs()
        self.held: list = []      # tensors whose pointers sit in the structs

    def samples(self, level: int) -> int:
        return self.num_coarse if level < 3 else self.num_coarse + self.num_fine

    def forward(self, rays: Dict[str, Tensor], draws: Dict[str, Optional[Tensor]]):
        """-> (z_vals_coarse, z_vals_fine or None, {level: {name: tensor}})"""
        lib = self.lib
        n = rays['rays_o'].shape[0]
        dev = rays['rays_o'].device
        self.n, self.device = n, dev
        r = self.c_rays
        held = self.held = []

        def put(name, t, shape):
            t = _dev(t, name, shape)
            held.append(t)
            return 0 if t is None else t.data_ptr()

        r.rays_o, r.rays_d = put('rays_o', rays['rays_o'], (n, 3)), put('rays_d', rays['rays_d'], (n, 3))
        need_dirs = any(self.mlps[l].desc.use_view_dirs for l in self.levels)
        if need_dirs and rays.get('view_dirs') is None:
            raise KeyError('view_dirs')
        r.view_dirs = put('view_dirs', rays.get('view_dirs'), (n, 3)) if need_dirs else 0
        if self.ndc:
            r.rays_o_ndc, r.rays_d_ndc = put('rays_o_ndc', rays['rays_o_ndc'], (n, 3)), put('rays_d_ndc', rays['rays_d_ndc'], (n, 3))
            near, far = rays['near_ndc'], rays['far_ndc']
        else:
            r.rays_o_ndc = r.rays_d_ndc = 0
            near, far = rays['near'], rays['far']
        r.near, r.far = put('near', near.reshape(-1), (n,)), put('far', far.reshape(-1), (n,))
        r.t_rand = put('t_rand', draws.get('t_rand'), (n, self.num_coarse))
        r.u = put('u', draws.get('u'), (n, self.num_fine)) if self.num_fine else 0
        for l in range(_lib.RENDER_LEVELS):
            noise = draws.get(('noise', l)) if l in self.levels else None
            r.sigma_noise[l] = put('sigma_noise', None if noise is None else noise.reshape(n, self.samples(l)), (n, self.samples(l)))
        override = draws.get('z_vals_fine') if self.num_fine else None
        r.depths_fine = put('z_vals_fine', override, (n, self.num_coarse + self.num_fine))
        fine_in = held[-1]        # the validated (contiguous fp32) override, or None: returned as z_vals_fine below
        # predict_visibility: secondary camera centres (n, K, 3), K = num_frames - 1 (sec_views_vis)
        rays_o2 = rays.get('rays_o2') if any(self.mlps[l].desc.predict_visibility for l in self.levels) else None
        k_other = 0 if rays_o2 is None else int(rays_o2.shape[1])
        r.rays_o2 = put('rays_o2', rays_o2, (n, k_other, 3)) if k_other else 0
        r.num_other = k_other

        # ---- outputs: one allocation per group, views carved out of it
        per_ray = [(name, width) for name, width in self.PER_RAY if self.ndc or not name.endswith('_ndc')]
        ray_floats = sum(w for _, w in per_ray) * n * len(self.levels)
        sample_floats = n * self.num_coarse + (n * (self.num_coarse + self.num_fine) if self.num_fine and override is None else 0)
        for l in self.levels:
            sample_floats += n * self.samples(l) * (4 + len(self.per_sample))
            if self.mlps[l].desc.predict_visibility:
                want_weights = 0 if 'weights' in self.per_sample or not k_other else 1
                sample_floats += n * self.samples(l) * (1 + 4 * k_other + want_weights)
                ray_floats += n * k_other
        small = torch.empty((ray_floats,), dtype=torch.float32, device=dev)
        big = torch.empty((sample_floats,), dtype=torch.float32, device=dev)
        pos = {'small': 0, 'big': 0}

        def carve(pool, name, shape):
            buf = small if pool == 'small' else big
            size = 1
            for d in shape:
                size *= d
            t = buf.as_strided(shape, _row_major_strides(shape), pos[pool])     # one view op (slice + view would be two)
            pos[pool] += size
            return t

        o = self.c_out
        z_coarse = carve('big', 'z', (n, self.num_coarse))
        o.depths_coarse = z_coarse.data_ptr()
        z_fine = None
        if self.num_fine:
            z_fine = fine_in if override is not None else carve('big', 'z', (n, self.num_coarse + self.num_fine))
            o.depths_fine = 0 if override is not None else z_fine.data_ptr()
        out: Dict[int, Dict[str, Tensor]] = {}
        for l in range(_lib.RENDER_LEVELS):
            lo = o.level[l]
            if l not in self.levels:
                for f in _lib.LEVEL_OUT_FIELDS:
                    setattr(lo, f, 0)
                continue
            s = self.samples(l)
            d = out[l] = {}
            for name, width in per_ray:
                d[name] = carve('small', name, (n, 3) if width == 3 else (n,))
                setattr(lo, name, d[name].data_ptr())
            if not self.ndc:
                lo.depth_ndc = lo.depth_var_ndc = 0
            for name in ('alpha', 'visibility', 'weights'):
                if name in self.per_sample:
                    d[name] = carve('big', name, (n, s))
                    setattr(lo, name, d[name].data_ptr())
                else:
                    setattr(lo, name, 0)
            d['sigma'], d['raw_rgb'] = carve('big', 'sigma', (n, s, 1)), carve('big', 'raw_rgb', (n, s, 3))
            lo.sigma, lo.raw_rgb = d['sigma'].data_ptr(), d['raw_rgb'].data_ptr()
            lo.raw_visibility = lo.raw_visibility2 = lo.visibility2 = lo.view_dirs2 = 0
            if self.mlps[l].desc.predict_visibility:
                d['raw_visibility'] = carve('big', 'raw_visibility', (n, s, 1))
                lo.raw_visibility = d['raw_visibility'].data_ptr()
                if k_other:
                    d['raw_visibility2'] = carve('big', 'raw_visibility2', (n, s, k_other, 1))
                    d['visibility2'] = carve('small', 'visibility2', (n, k_other))
                    dirs2 = carve('big', 'view_dirs2', (n, s, k_other, 3))
                    lo.raw_visibility2, lo.visibility2, lo.view_dirs2 = d['raw_visibility2'].data_ptr(), d['visibility2'].data_ptr(), dirs2.data_ptr()
                    if not lo.weights:      # visibility2 is composited with the level's weights
                        lo.weights = carve('big', 'weights', (n, s)).data_ptr()
            if self.keep:
                d['saved'] = torch.empty((lib.snerf_mlp_saved_floats(ctypes.byref(self.mlps[l].desc), n, s),), dtype=torch.float32, device=dev)
                lo.saved_acts = d['saved'].data_ptr()
            else:
                lo.saved_acts = 0
        work = None
        if self.num_fine and override is None and 'weights' not in self.per_sample:
            work = torch.empty((lib.snerf_render_workspace_floats(ctypes.byref(self.cfg), n),), dtype=torch.float32, device=dev)
        if n > 0:
            with torch.cuda.device(dev):
                st = lib.snerf_render_forward(ctypes.byref(self.cfg), self.c_mlps, ctypes.byref(r), n, ctypes.byref(o), _ptr(work),
                                              _stream())
            _lib.check(st, 'snerf_render_forward')
        # what backward() needs alive is the memory behind the pointers in the structs: the two pools, the saved
        # activations and the inputs.  The output VIEWS are not kept here: under autograd they carry the node that owns
        # this object, and holding them would close a reference cycle that only the cycle collector frees (GBs per call)
        held += [small, big] + [d['saved'] for d in out.values() if 'saved' in d]
        return z_coarse, z_fine, out

    def backward(self, grads: Dict[int, Dict[str, Optional[Tensor]]], param_grads: Dict[int, List[Tensor]],
                 accumulate: Dict[int, bool]) -> None:
        """``grads[level]``: dL/d(rgb, acc, depth, depth_ndc, sigma, raw_rgb) (missing/None = zero);
        ``param_grads[level]``: the level's gradient tensors in C-ABI order, overwritten or (``accumulate[level]``) added to."""
        lib = self.lib
        n, dev = self.n, self.device
        if n == 0:
            return
        c_grads = (_lib.RenderLevelGrads * _lib.RENDER_LEVELS)()
        keep = []
        for l in self.levels:
            g = grads.get(l) or {}
            if l not in param_grads or not any(g.get(k) is not None for k in ('rgb', 'acc', 'depth', 'depth_ndc', 'sigma', 'raw_rgb')):
                continue
            s = self.samples(l)
            cg = c_grads[l]
            for name, shape in (('rgb', (n, 3)), ('acc', (n,)), ('depth', (n,)), ('depth_ndc', (n,)), ('sigma', (n, s, 1)),
                                ('raw_rgb', (n, s, 3))):
                t = g.get(name)
                if t is not None:
                    t = _dev(t.reshape(shape), f'grad {name}', shape)
                    keep.append(t)
                setattr(cg, name, 0 if t is None else t.data_ptr())
            tensors = param_grads[l]
            if len(tensors) != self.mlps[l].num_params:
                raise RuntimeError(f'level {l}: expected {self.mlps[l].num_params} gradient tensors, got {len(tensors)}')
            arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
            keep.append(arr)
            cg.param_grads = arr
            cg.num_params = len(tensors)
            cg.accumulate = int(bool(accumulate.get(l, False)))
        work = torch.empty((lib.snerf_render_backward_workspace_floats(ctypes.byref(self.cfg), self.c_mlps, n),),
                           dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = lib.snerf_render_backward(ctypes.byref(self.cfg), self.c_mlps, ctypes.byref(self.c_rays), n,
                                           ctypes.byref(self.c_out), c_grads, _ptr(work), _stream())
        _lib.check(st, 'snerf_render_backward')


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
def to_display(rgb: Optional[Tensor], depth: Optional[Tensor] = None, colour: bool = True):
    """-> uint8 image (n,3) [clip to [0,1], round(255 x) half-to-even] and depth clipped at 0 (or None).
    ``colour=False`` converts only the depth column (image is None)."""
    lib = _lib.load()
    n = rgb.shape[0] if colour else depth.shape[0]
    rgb = _dev(rgb, 'rgb', (n, 3)) if colour else None
    depth = _dev(depth, 'depth', (n,))
    dev = rgb.device if colour else depth.device
    image = torch.empty((n, 3), dtype=torch.uint8, device=dev) if colour else None
    depth_out = None if depth is None else torch.empty((n,), dtype=torch.float32, device=dev)
    if n == 0:
        return image, depth_out
    with torch.cuda.device(dev):
        st = lib.snerf_to_display(_ptr(rgb), _ptr(depth), n, ctypes.c_void_p(0 if image is None else image.data_ptr()),
                                  _ptr(depth_out), _stream())
    _lib.check(st, 'snerf_to_display')
    return image, depth_out


# ---------------------------------------------------------------------------------------------- f1 losses
class LossTermSpec:
    """One masked mean-squared-error term of the fused loss evaluation (struct snerf_loss_term)."""
    __slots__ = ('pred', 'target', 'numerator_mask', 'denominator_mask', 'group', 'weight')

    def __init__(self, pred: Tensor, target: Tensor, numerator_mask: Optional[Tensor], denominator_mask: Optional[Tensor],
                 group: int, weight: float):
        n = pred.shape[0]
        self.pred = _dev(pred, 'loss pred')
        self.target = _dev(target, 'loss target', tuple(pred.shape))
        self.numerator_mask = _mask(numerator_mask, 'numerator_mask', n)
        self.denominator_mask = _mask(denominator_mask, 'denominator_mask', n)
        self.group, self.weight = int(group), float(weight)
        if self.pred.dim() not in (1, 2) or (self.pred.dim() == 2 and self.pred.shape[1] > 4):
            raise RuntimeError(f'loss pred: expected (n,) or (n,c<=4), got {tuple(pred.shape)}')


def _mask(t: Optional[Tensor], name: str, n: int) -> Optional[Tensor]:
    if t is None:
        return None
    if not t.is_cuda or t.dtype not in (torch.bool, torch.uint8) or tuple(t.shape) != (n,):
        raise RuntimeError(f'{name}: expected a bool/uint8 GPU tensor of shape ({n},), got {t.dtype} {tuple(t.shape)} on {t.device}')
    return t.contiguous()


# scratch of snerf_loss_forward (block partials + completion counter), one per device; calls on one device are ordered
# by the stream they are enqueued on -- callers that evaluate losses on several streams at once need one buffer each
_loss_workspaces: Dict[torch.device, Tensor] = {}


def _loss_table(terms: List[LossTermSpec], grads: Optional[List[Optional[Tensor]]]):
    if not 1 <= len(terms) <= _lib.LOSS_MAX_TERMS:
        raise RuntimeError(f'fused loss: {len(terms)} terms, the kernel table holds 1..{_lib.LOSS_MAX_TERMS}')
    table = (_lib.LossTerm * len(terms))()
    seen = set()
    for i, t in enumerate(terms):
        e = table[i]
        e.pred, e.target = t.pred.data_ptr(), t.target.data_ptr()
        e.numerator_mask = 0 if t.numerator_mask is None else t.numerator_mask.data_ptr()
        e.denominator_mask = 0 if t.denominator_mask is None else t.denominator_mask.data_ptr()
        e.channels = 1 if t.pred.dim() == 1 else t.pred.shape[1]
        e.group, e.weight = t.group, t.weight
        e.d_pred, e.accumulate = 0, 0
        if grads is not None and grads[i] is not None:
            e.d_pred = grads[i].data_ptr()
            e.accumulate = int(e.d_pred in seen)
            seen.add(e.d_pred)
    return table


def loss_forward(terms: List[LossTermSpec], num_groups: int):
    """-> values (T+G+1: term values, per-loss sums, weighted total), scales (T) for loss_backward.  One launch."""
    lib = _lib.load()
    dev = terms[0].pred.device
    n, count = terms[0].pred.shape[0], len(terms)
    if any(t.pred.shape[0] != n for t in terms):
        raise RuntimeError('fused loss: every term must cover the same rays')
    values = torch.empty((count + num_groups + 1,), dtype=torch.float32, device=dev)
    scales = torch.empty((count,), dtype=torch.float32, device=dev)
    ws = _loss_workspaces.get(dev)
    if ws is None:
        ws = _loss_workspaces[dev] = torch.zeros((int(lib.snerf_loss_workspace_bytes()),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = lib.snerf_loss_forward(_loss_table(terms, None), count, int(num_groups), n, _ptr(values), _ptr(scales),
                                    ctypes.c_void_p(ws.data_ptr()), _stream())
    _lib.check(st, 'snerf_loss_forward')
    return values, scales


def loss_backward(terms: List[LossTermSpec], num_groups: int, scales: Tensor, upstream: Tensor,
                  wanted: List[bool]) -> List[Optional[Tensor]]:
    """Gradient of every term's pred (None where ``wanted[i]`` is false).  Terms that share a pred tensor share one
    buffer holding the sum.  One launch."""
    lib = _lib.load()
    n = terms[0].pred.shape[0]
    buffers: Dict[int, Tensor] = {}
    grads: List[Optional[Tensor]] = []
    for t, want in zip(terms, wanted):
        if not want:
            grads.append(None)
            continue
        key = t.pred.data_ptr()
        if key not in buffers:
            buffers[key] = torch.empty_like(t.pred)
        grads.append(buffers[key])
    upstream = _dev(upstream, 'upstream', (len(terms) + num_groups + 1,))
    if n > 0 and buffers:
        with torch.cuda.device(scales.device):
            st = lib.snerf_loss_backward(_loss_table(terms, grads), len(terms), int(num_groups), n, _ptr(scales),
                                         _ptr(upstream), _stream())
        _lib.check(st, 'snerf_loss_backward')
    return grads


def patch_consistency_masks(rays_o: Tensor, rays_d: Tensor, depth1: Tensor, depth2: Tensor, ray_mask: Optional[Tensor],
                            pixel_id: Tensor, poses: Tensor, intrinsic: Tensor, images: Tensor, patch_size,
                            rmse_threshold: float, with_rmse: bool = False):
    """Decision masks of the patch-reprojection consistency losses: (mask1, mask2[, rmse1, rmse2]) per ray."""
    lib = _lib.load()
    n = rays_o.shape[0]
    v, h, w, c = images.shape
    if c != 3:
        raise RuntimeError(f'images: expected (views,h,w,3), got {tuple(images.shape)}')
    rays_o, rays_d = _dev(rays_o, 'rays_o', (n, 3)), _dev(rays_d, 'rays_d', (n, 3))
    depth1, depth2 = _dev(depth1, 'depth1', (n,)), _dev(depth2, 'depth2', (n,))
    ray_mask = _mask(ray_mask, 'ray_mask', n)
    if not pixel_id.is_cuda or pixel_id.dtype != torch.int32 or tuple(pixel_id.shape) != (n, 3):
        raise RuntimeError(f'pixel_id: expected int32 GPU tensor ({n},3), got {pixel_id.dtype} {tuple(pixel_id.shape)}')
    pixel_id = pixel_id.contiguous()
    poses, images = _dev(poses, 'poses', (v, 4, 4)), _dev(images, 'images')
    intrinsic = _dev(intrinsic, 'intrinsic', (3, 3))
    dev = rays_o.device
    mask1 = torch.empty((n,), dtype=torch.bool, device=dev)
    mask2 = torch.empty((n,), dtype=torch.bool, device=dev)
    rmse1 = torch.empty((n,), dtype=torch.float32, device=dev) if with_rmse else None
    rmse2 = torch.empty((n,), dtype=torch.float32, device=dev) if with_rmse else None
    if n > 0:
        with torch.cuda.device(dev):
            st = lib.snerf_patch_consistency_masks(
                _ptr(rays_o), _ptr(rays_d), _ptr(depth1), _ptr(depth2), _ptr(ray_mask), _ptr(pixel_id), n, _ptr(poses),
                _ptr(intrinsic), _ptr(images), v, h, w, int(patch_size[0]), int(patch_size[1]), float(rmse_threshold),
                _ptr(mask1), _ptr(mask2), _ptr(rmse1), _ptr(rmse2), _stream())
        _lib.check(st, 'snerf_patch_consistency_masks')
    return (mask1, mask2, rmse1, rmse2) if with_rmse else (mask1, mask2)


# ---------------------------------------------------------------------------------------------- f4 optimiser
def adam_step(params: List[Tensor], grads: List[Optional[Tensor]], exp_avg: List[Tensor], exp_avg_sq: List[Tensor],
              step: int, lr: float, beta1: float, beta2: float, eps: float, at: Optional[Tensor] = None) -> None:
    """In-place Adam update of every tensor with a gradient (one launch per 64 tensors).  ``at``: the device-resident
    iteration record (``IterationRing.current``) to take the step-dependent factors from instead of ``step`` / ``lr``."""
    lib = _lib.load()
    n = len(params)
    if n == 0:
        return
    f32 = torch.float32
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        for name, t in (('param', p), ('exp_avg', m), ('exp_avg_sq', v)):
            if not t.is_cuda or t.dtype != f32 or not t.is_contiguous():
                raise RuntimeError(f'adam_step: {name} must be a contiguous float32 GPU tensor, got {t.dtype} on {t.device}')
        if g is not None and (not g.is_cuda or g.dtype != f32 or g.shape != p.shape):
            raise RuntimeError(f'adam_step: grad must be a float32 GPU tensor shaped like its parameter, got {g.dtype} '
                               f'{tuple(g.shape)} on {g.device}')
    grads = [None if g is None else (g if g.is_contiguous() else g.contiguous()) for g in grads]
    ptrs = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    sizes = (ctypes.c_longlong * n)(*[p.numel() for p in params])
    with torch.cuda.device(params[0].device):
        if at is not None:
            st = lib.snerf_adam_step_at(ptrs(params), ptrs(grads), ptrs(exp_avg), ptrs(exp_avg_sq), sizes, n,
                                        ctypes.c_void_p(at.data_ptr()), float(beta1), float(beta2), float(eps), _stream())
        else:
            st = lib.snerf_adam_step(ptrs(params), ptrs(grads), ptrs(exp_avg), ptrs(exp_avg_sq), sizes, n, int(step),
                                     float(lr), float(beta1), float(beta2), float(eps), _stream())
    _lib.check(st, 'snerf_adam_step')


# ---------------------------------------------------------------------------------------------- f2 batch assembly
def camera_table(intrinsics: Tensor, poses: Tensor, resolution) -> Tensor:
    """(V, 24) per-view camera constants (inverse intrinsic, rotation, origin, NDC factors) on the device."""
    lib = _lib.load()
    v = poses.shape[0]
    intrinsics, poses = _dev(intrinsics, 'intrinsics', (v, 3, 3)), _dev(poses, 'poses', (v, 4, 4))
    table = torch.empty((v, _lib.CAMERA_FLOATS), dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        st = lib.snerf_camera_table(_ptr(intrinsics), _ptr(poses), v, int(resolution[0]), int(resolution[1]), _ptr(table), _stream())
    _lib.check(st, 'snerf_camera_table')
    return table


def assemble_batch(indices: Tensor, num_pixel_rays: int, table: Tensor, resolution, images: Tensor, ndc: bool, near: float,
                   far: float, near_ndc: float = 0.0, far_ndc: float = 1.0, sparse_depths: Optional[Tensor] = None,
                   sparse_errors: Optional[Tensor] = None, sparse_depths_ndc: Optional[Tensor] = None,
                   with_sparse_mask: bool = False, first_pixel_row: int = 0, first_sparse_row: Optional[int] = None
                   ) -> Dict[str, Tensor]:
    """One launch: rays (+NDC), view_dirs, pixel_id, target_rgb, near/far columns, sparse-depth columns and the row
    masks for the global pixel ``indices`` (int64 GPU tensor); the first ``num_pixel_rays`` rows are pixel rays.
    ``global_rows`` (int64, (n,)) numbers the rows as the single-process batch would: ``first_pixel_row + i`` for pixel
    rows, ``first_sparse_row + j`` for sparse rows (default: right behind this call's pixel rows)."""
    lib = _lib.load()
    if not indices.is_cuda or indices.dtype != torch.int64 or indices.dim() != 1:
        raise RuntimeError(f'indices: expected a 1-D int64 GPU tensor, got {indices.dtype} {tuple(indices.shape)} on {indices.device}')
    indices = indices.contiguous()
    n, dev = indices.shape[0], indices.device
    h, w = int(resolution[0]), int(resolution[1])
    v = table.shape[0]
    table = _dev(table, 'camera table', (v, _lib.CAMERA_FLOATS))
    images = _dev(images, 'images', (v, h, w, 3))
    tables = [None if t is None else _dev(t.reshape(-1), name, (v * h * w,)) for t, name in
              ((sparse_depths, 'sparse_depths'), (sparse_errors, 'sparse_errors'), (sparse_depths_ndc, 'sparse_depths_ndc'))]
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    out = {'rays_o': f(n, 3), 'rays_d': f(n, 3), 'view_dirs': f(n, 3),
           'pixel_id': torch.empty((n, 3), dtype=torch.int32, device=dev), 'target_rgb': f(n, 3), 'near': f(n, 1), 'far': f(n, 1),
           'indices_mask_nerf': torch.empty((n,), dtype=torch.bool, device=dev),
           'global_rows': torch.empty((n,), dtype=torch.int64, device=dev)}
    if ndc:
        out.update(rays_o_ndc=f(n, 3), rays_d_ndc=f(n, 3), near_ndc=f(n, 1), far_ndc=f(n, 1))
    for key, t in zip(('sparse_depth_values', 'sparse_depth_errors', 'sparse_depth_values_ndc'), tables):
        if t is not None:
            out[key] = f(n, 1)
    if with_sparse_mask:
        out['indices_mask_sparse_depth'] = torch.empty((n,), dtype=torch.bool, device=dev)
    if n == 0:
        return out
    b = _lib.Batch()
    for field, key in (('rays_o', 'rays_o'), ('rays_d', 'rays_d'), ('view_dirs', 'view_dirs'), ('rays_o_ndc', 'rays_o_ndc'),
                       ('rays_d_ndc', 'rays_d_ndc'), ('pixel_id', 'pixel_id'), ('target_rgb', 'target_rgb'), ('near', 'near'),
                       ('far', 'far'), ('near_ndc', 'near_ndc'), ('far_ndc', 'far_ndc'),
                       ('sparse_depth_values', 'sparse_depth_values'), ('sparse_depth_errors', 'sparse_depth_errors'),
                       ('sparse_depth_values_ndc', 'sparse_depth_values_ndc'), ('mask_pixel_rays', 'indices_mask_nerf'),
                       ('mask_sparse_rays', 'indices_mask_sparse_depth'), ('global_rows', 'global_rows')):
        setattr(b, field, out[key].data_ptr() if key in out else None)
    with torch.cuda.device(dev):
        st = lib.snerf_assemble_batch(ctypes.c_void_p(indices.data_ptr()), n, int(num_pixel_rays), _ptr(table), v, h, w,
                                      _ptr(images), _ptr(tables[0]), _ptr(tables[1]), _ptr(tables[2]), int(ndc), float(near),
                                      float(far), float(near_ndc), float(far_ndc), int(first_pixel_row),
                                      int(first_pixel_row + num_pixel_rays if first_sparse_row is None else first_sparse_row),
                                      ctypes.byref(b), _stream())
    _lib.check(st, 'snerf_assemble_batch')
    return out


def shuffled_indices(seed: int, epoch: int, first: int, count: int, domain: int, device, candidates: Optional[Tensor] = None,
                     num_views: int = 0, resolution=(0, 0), crop=None, at: Optional[Tensor] = None, sparse: bool = False) -> Tensor:
    """Positions [first, first+count) of epoch ``epoch``'s permutation of the candidate pixels, as global pixel indices.
    ``at`` (``IterationRing.current``): epoch and position come from the device-resident iteration record (its pixel or
    ``sparse`` pair), ``first`` is then the offset added to that position (a rank's shard of the slice)."""
    lib = _lib.load()
    h, w = int(resolution[0]), int(resolution[1])
    y0, y1, x0, x1 = crop if crop is not None else (0, h, 0, w)
    if candidates is not None:
        if not candidates.is_cuda or candidates.dtype != torch.int64 or tuple(candidates.shape) != (domain,):
            raise RuntimeError(f'candidates: expected int64 GPU tensor ({domain},), got {candidates.dtype} {tuple(candidates.shape)}')
        candidates = candidates.contiguous()
        device = candidates.device
    out = torch.empty((count,), dtype=torch.int64, device=device)
    if count == 0:
        return out
    with torch.cuda.device(out.device):
        if at is not None:
            st = lib.snerf_shuffled_indices_at(int(seed) & (2 ** 64 - 1), ctypes.c_void_p(at.data_ptr()), int(bool(sparse)), int(first),
                                               int(count), int(domain), ctypes.c_void_p(0 if candidates is None else candidates.data_ptr()),
                                               int(num_views), h, w, int(y0), int(y1), int(x0), int(x1),
                                               ctypes.c_void_p(out.data_ptr()), _stream())
        else:
            st = lib.snerf_shuffled_indices(int(seed) & (2 ** 64 - 1), int(epoch), int(first), int(count), int(domain),
                                            ctypes.c_void_p(0 if candidates is None else candidates.data_ptr()), int(num_views), h, w,
                                            int(y0), int(y1), int(x0), int(x1), ctypes.c_void_p(out.data_ptr()), _stream())
    _lib.check(st, 'snerf_shuffled_indices')
    return out


def _draw_target(shape, device, out: Optional[Tensor]) -> Tensor:
    if out is None:
        return torch.empty(tuple(shape), dtype=torch.float32, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
        raise RuntimeError(f'draw target: expected a contiguous float32 GPU tensor of shape {tuple(shape)}')
    return out


def _row_ids(rows: Optional[Tensor], count: int) -> Optional[Tensor]:
    if rows is None:
        return None
    if not rows.is_cuda or rows.dtype != torch.int64 or tuple(rows.shape) != (count,):
        raise RuntimeError(f'global rows: expected an int64 GPU tensor of shape ({count},), got {rows.dtype} {tuple(rows.shape)} on {rows.device}')
    return rows.contiguous()


def random_uniform(seed: int, stream_id: int, first_row: int, shape, device, out: Optional[Tensor] = None,
                   rows: Optional[Tensor] = None, at=None) -> Tensor:
    """[0,1) Philox draws of shape (rows, width...); element (r, c) depends only on (seed, stream_id, global row of r, c),
    where the global row is ``rows[r]`` (int64 GPU tensor, e.g. a batch's ``global_rows``) or ``first_row + r``.
    ``out`` (optional) receives the draws in place (static buffers of a captured graph)."""
    lib = _lib.load()
    out = _draw_target(shape, device, out)
    count = int(shape[0])
    width = out.numel() // count if count else 1
    if out.numel() == 0:
        return out
    rows = _row_ids(rows, count)
    with torch.cuda.device(out.device):
        if at is not None:      # (record, kind, number of kinds): the stream is record.iter_num * kinds + kind, read on the device
            st = lib.snerf_random_uniform_at(int(seed) & (2 ** 64 - 1), ctypes.c_void_p(at[0].data_ptr()), int(at[1]), int(at[2]),
                                             int(first_row), ctypes.c_void_p(0 if rows is None else rows.data_ptr()), count,
                                             max(width, 1), _ptr(out), _stream())
        else:
            st = lib.snerf_random_uniform(int(seed) & (2 ** 64 - 1), int(stream_id) & 0xFFFFFFFF, int(first_row),
                                          ctypes.c_void_p(0 if rows is None else rows.data_ptr()), count, max(width, 1), _ptr(out),
                                          _stream())
    _lib.check(st, 'snerf_random_uniform')
    return out


def random_normal(seed: int, stream_id: int, first_row: int, shape, device, scale: float = 1.0,
                  out: Optional[Tensor] = None, rows: Optional[Tensor] = None, at=None) -> Tensor:
    lib = _lib.load()
    out = _draw_target(shape, device, out)
    count = int(shape[0])
    width = out.numel() // count if count else 1
    if out.numel() == 0:
        return out
    rows = _row_ids(rows, count)
    with torch.cuda.device(out.device):
        if at is not None:
            st = lib.snerf_random_normal_at(int(seed) & (2 ** 64 - 1), ctypes.c_void_p(at[0].data_ptr()), int(at[1]), int(at[2]),
                                            int(first_row), ctypes.c_void_p(0 if rows is None else rows.data_ptr()), count,
                                            max(width, 1), float(scale), _ptr(out), _stream())
        else:
            st = lib.snerf_random_normal(int(seed) & (2 ** 64 - 1), int(stream_id) & 0xFFFFFFFF, int(first_row),
                                         ctypes.c_void_p(0 if rows is None else rows.data_ptr()), count, max(width, 1), float(scale),
                                         _ptr(out), _stream())
    _lib.check(st, 'snerf_random_normal')
    return out


# ---------------------------------------------------------------------------------------------- whole-iteration graphs
class IterationRing:
    """The per-iteration scalars of a replayed training graph (struct snerf_iteration, include/simplenerf_train.h): a ring of
    records in PINNED host memory that the host fills ahead of time, a device-resident counter and the device-resident
    ``current`` record.  ``advance()`` (captured as the graph's first node) makes record ``counter mod slots`` current."""

    def __init__(self, device, slots: int = 8):
        self.slots = int(slots)
        self.device = torch.device(device)
        self.ring = torch.zeros((self.slots, _lib.ITERATION_WORDS), dtype=torch.int64).pin_memory()
        self.view = self.ring.numpy()
        self.counter = torch.zeros((1,), dtype=torch.int64, device=self.device)
        self.current = torch.zeros((_lib.ITERATION_WORDS,), dtype=torch.int64, device=self.device)
        self.filled = 0                       # records written so far = index of the next replay

    def fill(self, iter_num: int, pixel: tuple, sparse: tuple, adam_neg_step_size: float, adam_bias2_sqrt: float) -> int:
        """Write the record of the NEXT replay; ``pixel`` / ``sparse`` = (epoch, first).  -> its slot."""
        slot = self.filled % self.slots
        row = self.view[slot]
        row[0], row[1], row[2], row[3], row[4] = int(iter_num), int(pixel[0]), int(pixel[1]), int(sparse[0]), int(sparse[1])
        row[5] = int(numpy.array([adam_neg_step_size, adam_bias2_sqrt], dtype=numpy.float32).view(numpy.int64)[0])
        self.filled += 1
        return slot

    def advance(self) -> None:
        lib = _lib.load()
        with torch.cuda.device(self.device):
            st = lib.snerf_iteration_advance(ctypes.c_void_p(self.ring.data_ptr()), self.slots, ctypes.c_void_p(self.counter.data_ptr()),
                                             ctypes.c_void_p(self.current.data_ptr()), _stream())
        _lib.check(st, 'snerf_iteration_advance')

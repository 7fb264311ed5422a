"""MI355X renderer behind the reference's model interface.

``SimpleNeRFHip(configs, model_configs)`` is a ``torch.nn.Module`` whose constructor arguments, parameter names and
shapes, ``forward(input_batch, retraw=False, sec_views_vis=False) -> dict`` signature and output keys are those of
the reference's ``SimpleNeRF`` (src/models/SimpleNeRF01.py:11-75, :108-270), so the reference's Trainer/Tester call
sites (src/Trainer01.py:93,194,328; src/Tester01.py:63) work unchanged.  All arithmetic runs in the HIP library
(simplenerf_amd/csrc) through the C ABI in include/simplenerf_hip.h; PyTorch only owns device memory, the stream and
the parameters.  There is no eager/CPU fallback: CPU tensors or an unbuilt library raise.

One extra, optional config key: ``configs['model']['hip_precision']`` = ``'fp32'`` (default; fp32 matrix cores) or
``'f16x3'`` (fp16 hi/lo split, three MFMAs per product, fp32 accumulate: fp32-grade results ~2.5x faster; covers the
forward, the activation-keeping training forward, the backward dgrad chain and the large weight-gradient products), or
``'f16'`` (16-bit mode: one fp16 MFMA per product, 16-bit saved activations / layer gradients, fp32 master weights and
accumulation; ~4x faster training than fp32 at ~1e-3 agreement -- outside the fp32 parity bar, tests/test_gpu_f16.py).

Differences from the reference that a caller can observe:
  * ``model.chunk`` / ``model.netchunk`` are accepted and ignored -- the kernels tile the work themselves and the
    results do not depend on chunking;
  * training-mode randomness (stratified jitter, inverse-CDF ``u``, density noise) is drawn on the device by the
    library's counter-based Philox4x32-10 (snerf_random_uniform / snerf_random_normal) instead of on the CPU
    generator: element (ray, sample) of a draw is a function of (``configs['seed']``, number of training forwards so
    far, kind of draw, GLOBAL row of the ray, sample) only, so a rank that holds a part of a batch draws what a single
    process would for those rows.  The global rows come from ``input_batch['row_offset']`` (rows are
    [row_offset, row_offset+n)) or, when a rank's rows are not contiguous in the global batch -- a pixel-ray shard
    followed by a sparse-depth shard, as ``BatchAssembler`` produces -- from ``input_batch['row_segments']``, a list of
    (first local row, count, first global row).  ``set_random_draws`` injects explicit draws (used by the parity tests
    to replay the reference's CPU stream);
  * ``predict_visibility`` (off in every shipped config) is not built;
  * gradients flow from ``rgb_*``, ``acc_*``, ``depth_*``, ``depth_ndc_*`` (incl. the augmentation-prefixed ones) and
    ``raw_sigma_*`` / ``raw_rgb*_*`` to the parameters -- a superset of what the shipped losses read (SURVEY 8a row
    9); ``alpha_*``, ``visibility_*``, ``weights_*``, ``depth_var*`` and ``z_vals_*`` are returned without a
    gradient path (the reference detaches the sample depths, :312; no loss reads the others), so a loss built only
    on them raises instead of silently training nothing.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops

Tensor = torch.Tensor
_SKIP_AFTER = 4  # reference: self.skips = [4]
_DRAW_KINDS = ('t_rand', 'u', 'noise_coarse', 'noise_points_augmentation', 'noise_views_augmentation', 'noise_fine',
               'noise_points_augmentation_fine', 'noise_views_augmentation_fine')


def row_segments(batch: dict, n: int) -> List[tuple]:
    """[(first local row, count, first global row)] covering rows 0..n of ``batch``: its ``row_segments`` entry if
    present, else one segment at ``row_offset`` (default 0).  Adjacent segments that are contiguous globally are merged."""
    segs = batch.get('row_segments')
    if segs is None:
        return [(0, n, int(batch.get('row_offset', 0)))]
    out: List[tuple] = []
    pos = 0
    for first, count, glob in segs:
        first, count, glob = int(first), int(count), int(glob)
        if first != pos or count < 0:
            raise RuntimeError(f'row_segments must tile the batch rows in order, got {list(segs)} for {n} rows')
        pos += count
        if count == 0:
            continue
        if out and out[-1][2] + out[-1][1] == glob:
            out[-1] = (out[-1][0], out[-1][1] + count, out[-1][2])
        else:
            out.append((first, count, glob))
    if pos != n:
        raise RuntimeError(f'row_segments cover {pos} rows, the batch has {n}')
    return out or [(0, 0, 0)]


def slice_row_segments(segments: List[tuple], start: int, count: int) -> List[tuple]:
    """Segments of the sub-batch made of local rows [start, start+count)."""
    out = []
    for first, cnt, glob in segments:
        lo, hi = max(first, start), min(first + cnt, start + count)
        if hi > lo:
            out.append((lo - start, hi - lo, glob + lo - first))
    return out or [(0, 0, 0)]


def _draw_segments(fn, segments, shape, device, out, *args):
    """One keyed draw of ``shape`` whose rows follow ``segments``: a launch per segment into that segment's row block."""
    if len(segments) == 1:
        return fn(*args[:2], segments[0][2], shape, device, *args[2:], out=out)
    target = out if out is not None else torch.empty(tuple(shape), dtype=torch.float32, device=device)
    for first, count, glob in segments:
        fn(*args[:2], glob, (count,) + tuple(shape[1:]), device, *args[2:], out=target[first:first + count])
    return target


class MlpParameters(torch.nn.Module):
    """Parameters of one NeRF MLP under the reference's names, shapes and construction order (MLP.__init__
    src/models/SimpleNeRF01.py:561-609), so ``state_dict()`` round-trips with reference checkpoints and the default
    initialisation consumes the RNG identically."""

    def __init__(self, configs: dict, mlp_configs: dict):
        super().__init__()
        self.mlp_configs = mlp_configs
        if mlp_configs.get('predict_visibility', False):
            raise NotImplementedError('predict_visibility is not built in the HIP renderer')
        dp, wp = mlp_configs['points_net_depth'], mlp_configs['points_net_width']
        full_pe = 3 + 6 * mlp_configs['points_positional_encoding_degree']
        pts_in = full_pe
        views_in = 0
        if mlp_configs['use_view_dirs']:
            views_in = 3 + 6 * mlp_configs['views_positional_encoding_degree']
        if 'points_sigma_positional_encoding_degree' in mlp_configs:
            pts_in = (2 * mlp_configs['points_sigma_positional_encoding_degree'] + 1) * 3
            views_in += full_pe - pts_in
        self.view_dependent = bool(mlp_configs['view_dependent_rgb'])
        lin = torch.nn.Linear
        self.pts_linears = torch.nn.ModuleList(
            [lin(pts_in, wp)] + [lin(wp + pts_in if i == _SKIP_AFTER else wp, wp) for i in range(dp - 1)])
        if self.view_dependent:
            dv, wv = mlp_configs['views_net_depth'], mlp_configs['views_net_width']
            self.views_linears = torch.nn.ModuleList([lin(views_in + wp, wv)] + [lin(wv, wv) for _ in range(dv - 1)])
        self.pts_output_linear = lin(wp, 1 if self.view_dependent else 4)
        if self.view_dependent:
            self.feature_linear = lin(wp, wp)
            self.views_output_linear = lin(wv, 3)

    def abi_params(self) -> List[Tensor]:
        """Parameters in the order of the C ABI.  The list is cached (it is asked for ~16 times per training iteration);
        Parameter objects survive ``.to()`` / ``load_state_dict`` / optimiser steps, and a re-assigned first or last layer
        invalidates the cache."""
        cached = self.__dict__.get('_abi_params')
        last = self.views_output_linear.bias if self.view_dependent else self.pts_output_linear.bias
        if cached is not None and cached[0] is self.pts_linears[0].weight and cached[-1] is last and \
                cached[1] is self.pts_linears[0].bias:
            return cached
        out = self._build_abi_params()
        self.__dict__['_abi_params'] = out
        return out

    def _build_abi_params(self) -> List[Tensor]:
        out = []
        for layer in self.pts_linears:
            out += [layer.weight, layer.bias]
        out += [self.pts_output_linear.weight, self.pts_output_linear.bias]
        if self.view_dependent:
            out += [self.feature_linear.weight, self.feature_linear.bias, self.views_linears[0].weight,
                    self.views_linears[0].bias, self.views_output_linear.weight, self.views_output_linear.bias]
        return out

    def forward(self, *args, **kwargs):
        raise RuntimeError('MlpParameters only holds weights; evaluation happens in the fused HIP kernel')


class _ShadeFunction(torch.autograd.Function):
    """One MLP evaluated on (N,S) samples + compositing, with hand-written backward kernels (K6, K7).

    forward  = snerf_mlp_forward_train + snerf_composite
    backward = snerf_composite_backward -> (d sigma, d rgb) -> snerf_mlp_backward -> parameter gradients
    """
    COMPOSITE_KEYS = ('rgb', 'acc', 'alpha', 'visibility', 'weights', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc')

    @staticmethod
    def forward(ctx, packed, precision, ndc, white, march_o, march_d, view_dirs, depths, noise, rays_o, rays_d, *params):
        sigma, rgb, saved = packed.forward_train(march_o, march_d, view_dirs, depths, noise, precision)
        comp = ops.composite(sigma, rgb, depths, march_d, ndc, white, rays_o, rays_d)
        ctx.packed, ctx.ndc, ctx.white, ctx.precision = packed, ndc, white, precision
        ctx.param_shapes = [tuple(p.shape) for p in params]
        ctx.save_for_backward(sigma, rgb, saved, depths, march_d, rays_o, rays_d)
        outs = [comp.get(k) for k in _ShadeFunction.COMPOSITE_KEYS if k in comp]
        ctx.keys = [k for k in _ShadeFunction.COMPOSITE_KEYS if k in comp] + ['sigma', 'rgb_raw']
        no_grad = [comp[k] for k in ('alpha', 'visibility', 'weights', 'depth_var', 'depth_var_ndc') if k in comp]
        ctx.mark_non_differentiable(*no_grad)
        ctx.set_materialize_grads(False)
        return tuple(outs) + (sigma, rgb)

    @staticmethod
    def backward(ctx, *grad_outputs):
        sigma, rgb, saved, depths, march_d, rays_o, rays_d = ctx.saved_tensors
        g = dict(zip(ctx.keys, grad_outputs))
        d_sigma, d_rgb = ops.composite_backward(sigma, rgb, depths, march_d, ctx.ndc, ctx.white, rays_o, rays_d,
                                                g.get('rgb'), g.get('acc'), g.get('depth'), g.get('depth_ndc'))
        if g.get('sigma') is not None:
            d_sigma = d_sigma + g['sigma'].reshape(d_sigma.shape)
        if g.get('rgb_raw') is not None:
            d_rgb = d_rgb + g['rgb_raw']
        grads = ctx.packed.backward(saved, sigma, rgb, d_sigma, d_rgb, ctx.param_shapes, ctx.precision)
        return (None,) * 11 + tuple(grads)


class SimpleNeRFHip(torch.nn.Module):
    def __init__(self, configs: dict, model_configs: dict = None):
        super().__init__()
        self.configs = configs
        self.model_configs = model_configs
        mcfg = configs['model']
        self.ndc = configs['data_loader']['ndc']
        self.coarse_mlp_needed = 'coarse_mlp' in mcfg
        self.fine_mlp_needed = 'fine_mlp' in mcfg
        if not self.coarse_mlp_needed:
            raise NotImplementedError('a coarse_mlp is required (the reference cannot sample a fine pass without one)')
        # sub-model presence and construction order follow the reference (:45-65)
        self.coarse_model = MlpParameters(configs, mcfg['coarse_mlp'])
        self.fine_model = MlpParameters(configs, mcfg['fine_mlp']) if self.fine_mlp_needed else None
        self._train_only: List[tuple] = []  # (output prefix, level, attribute name)
        for key, short in (('points_augmentation', 'pts_aug'), ('views_augmentation', 'views_aug')):
            if key in mcfg:
                for level in ('coarse', 'fine'):
                    if f'{level}_mlp' in mcfg[key]:
                        name = f'{short}_{level}_model'
                        setattr(self, name, MlpParameters(configs, mcfg[key][f'{level}_mlp']))
                        self._train_only.append((f'{key}_', level, name))
        precision = mcfg.get('hip_precision', 'fp32')
        if precision not in ops.PRECISIONS:
            raise KeyError(f"model.hip_precision must be one of {sorted(ops.PRECISIONS)}, got {precision!r}")
        self.precision = ops.PRECISIONS[precision]
        self._packed: Dict[str, tuple] = {}
        self._draws: Optional[dict] = None
        self.seed = int(configs.get('seed', 0))
        self._train_calls = 0   # training-mode forwards so far: selects the Philox stream of each draw

    # ------------------------------------------------------------------------------------------
    def set_random_draws(self, draws: Optional[dict]) -> None:
        """Use these tensors for the next training-mode forward instead of device RNG.  Keys (all optional):
        ``t_rand`` (N,S_c); ``u`` (N,S_f); ``noise_coarse``, ``noise_points_augmentation``,
        ``noise_views_augmentation`` (N,S_c,1); ``noise_fine`` (N,S_c+S_f,1) -- noise already scaled by
        raw_noise_std.  A missing key means that draw is skipped (no jitter / no noise).
        Parity-test hook: ``z_vals_fine`` (N,S_c+S_f) replaces the resampled fine depths altogether (the reference's
        sample_pdf has a rounding-dependent discontinuity; works in eval mode too)."""
        self._draws = draws

    def invalidate_packed(self) -> None:
        """Force the next forward to re-pack every MLP's weights (used before a graph capture, so that the re-pack is
        part of the captured work and not skipped because the weights happen to be current at capture time)."""
        self._packed = {name: (None, entry[1]) for name, entry in self._packed.items()}

    def training_draw_shapes(self, n: int) -> Dict[str, tuple]:
        """Shapes of the draws one training-mode forward over ``n`` rays consumes, keyed like ``set_random_draws``."""
        mcfg = self.configs['model']
        s_c = mcfg['coarse_mlp']['num_samples']
        shapes: Dict[str, tuple] = {}
        if mcfg['perturb'] > 0.:
            shapes['t_rand'] = (n, s_c)
        noisy = float(mcfg['raw_noise_std']) > 0.
        if noisy:
            shapes['noise_coarse'] = (n, s_c, 1)
            for prefix, level, _ in self._train_only:
                if level == 'coarse':
                    shapes[f'noise_{prefix[:-1]}'] = (n, s_c, 1)
        if self.fine_mlp_needed:
            s_f = mcfg['fine_mlp']['num_samples']
            if mcfg['perturb'] > 0.:
                shapes['u'] = (n, s_f)
            if noisy:
                shapes['noise_fine'] = (n, s_c + s_f, 1)
                for prefix, level, _ in self._train_only:
                    if level == 'fine':
                        shapes[f'noise_{prefix[:-1]}_fine'] = (n, s_c + s_f, 1)
        return shapes

    def draw_training_randomness(self, n: int, row_offset, device, out: Optional[Dict[str, Tensor]] = None
                                 ) -> Dict[str, Tensor]:
        """The draws the next training-mode forward would make itself, produced up front (and counted as that forward's):
        pass the result to ``set_random_draws``.  ``row_offset``: the first global row (int) or a ``row_segments`` list.
        ``out`` supplies pre-allocated tensors to fill (static graph inputs)."""
        call = self._train_calls
        self._train_calls += 1
        noise_std = float(self.configs['model']['raw_noise_std'])
        segments = row_segments({'row_segments': row_offset}, n) if isinstance(row_offset, (list, tuple)) else [(0, n, int(row_offset))]
        draws = {}
        for key, shape in self.training_draw_shapes(n).items():
            stream = call * len(_DRAW_KINDS) + _DRAW_KINDS.index(key)
            target = None if out is None else out[key]
            if key.startswith('noise'):
                draws[key] = _draw_segments(ops.random_normal, segments, shape, device, target, self.seed, stream, noise_std)
            else:
                draws[key] = _draw_segments(ops.random_uniform, segments, shape, device, target, self.seed, stream)
        return draws

    def _packed_mlp(self, name: str) -> ops.PackedMlp:
        module: MlpParameters = getattr(self, name)
        params = module.abi_params()
        # staleness stamp: every parameter's in-place version counter (optimiser steps, load_state_dict, manual edits)
        # plus the storage address of the first one (a .to(device) / re-materialisation moves them all)
        stamp = (params[0].data_ptr(),) + tuple(p._version for p in params)
        entry = self._packed.get(name)
        if entry is None or entry[0] != stamp or entry[1].buffer.device != params[0].device:
            packed = entry[1] if entry is not None and entry[1].buffer.device == params[0].device \
                else ops.PackedMlp(module.mlp_configs, params[0].device)
            packed.pack(params)
            self._packed[name] = (stamp, packed)
        return self._packed[name][1]

    # ------------------------------------------------------------------------------------------
    def forward(self, input_batch: dict, retraw: bool = False, sec_views_vis: bool = False) -> Dict[str, Tensor]:
        batch = dict(input_batch)  # the caller's dict is never mutated (reference: deep_dict_copy :68)
        training = self.training
        retraw = retraw or training
        with_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        mcfg = self.configs['model']
        rays_o, rays_d = batch['rays_o'], batch['rays_d']
        if self.ndc:
            march_o, march_d = batch['rays_o_ndc'], batch['rays_d_ndc']
            near, far = batch['near_ndc'], batch['far_ndc']
        else:
            march_o, march_d = rays_o, rays_d
            near, far = batch['near'], batch['far']
        need_dirs = mcfg['coarse_mlp']['use_view_dirs'] or (self.fine_mlp_needed and mcfg['fine_mlp']['use_view_dirs'])
        view_dirs = batch['view_dirs'] if need_dirs else batch.get('view_dirs')
        n = rays_o.shape[0]
        dev = rays_o.device
        draws = self._draws
        self._draws = None
        noise_std = float(mcfg['raw_noise_std'])
        perturb = bool(mcfg['perturb'] > 0.) and training

        segments = row_segments(batch, n) if training else None
        call = self._train_calls
        if training and draws is None:
            self._train_calls += 1

        def draw(key, shape, normal):
            if not training:
                return None
            if draws is not None:
                t = draws.get(key)
                return None if t is None else t.to(dev)
            stream = call * len(_DRAW_KINDS) + _DRAW_KINDS.index(key)
            if normal:
                return _draw_segments(ops.random_normal, segments, shape, dev, None, self.seed, stream, noise_std) \
                    if noise_std > 0. else None
            return _draw_segments(ops.random_uniform, segments, shape, dev, None, self.seed, stream) if perturb else None

        out: Dict[str, Tensor] = {}

        def shade(name, prefix, level, depths, noise_key):
            s = depths.shape[1]
            packed = self._packed_mlp(name)
            noise = draw(noise_key, (n, s, 1), True)
            if with_grad:
                res = _ShadeFunction.apply(packed, self.precision, self.ndc, bool(mcfg['white_bkgd']), march_o, march_d, view_dirs, depths,
                                           noise, rays_o, rays_d, *getattr(self, name).abi_params())
                keys = [k for k in _ShadeFunction.COMPOSITE_KEYS if self.ndc or not k.endswith('_ndc')]
                comp = dict(zip(keys, res[:len(keys)]))
                sigma, rgb = res[len(keys)], res[len(keys) + 1]
            else:
                sigma, rgb = packed.forward(march_o, march_d, view_dirs, depths, noise, self.precision)
                comp = ops.composite(sigma, rgb, depths, march_d, self.ndc, mcfg['white_bkgd'], rays_o, rays_d)
            # key order as volume_rendering's return_dict (:465-477)
            for k in ('rgb', 'acc', 'alpha', 'visibility', 'weights', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc'):
                if k in comp:
                    out[f'{prefix}{k}_{level}'] = comp[k]
            if retraw:
                out[f'{prefix}raw_sigma_{level}'] = sigma
                variant = 'rgb_view_dependent' if packed.use_view_dirs else 'rgb_view_independent'
                out[f'{prefix}raw_{variant}_{level}'] = rgb
                out[f'{prefix}raw_rgb_{level}'] = rgb
            return comp

        s_c = mcfg['coarse_mlp']['num_samples']
        z_coarse = ops.coarse_depths(near, far, s_c, mcfg['lindisp'], draw('t_rand', (n, s_c), False))
        comp_c = shade('coarse_model', '', 'coarse', z_coarse, 'noise_coarse')
        out['z_vals_coarse'] = z_coarse
        if training:
            for prefix, level, name in self._train_only:
                if level == 'coarse':
                    shade(name, prefix, 'coarse', z_coarse, f'noise_{prefix[:-1]}')
        if self.fine_mlp_needed:
            s_f = mcfg['fine_mlp']['num_samples']
            if draws is not None and draws.get('z_vals_fine') is not None:
                z_fine = draws['z_vals_fine'].to(dev)
            else:
                z_fine = ops.resample_depths(z_coarse, comp_c['weights'].detach(), s_f, draw('u', (n, s_f), False))
            shade('fine_model', '', 'fine', z_fine, 'noise_fine')
            out['z_vals_fine'] = z_fine
            if training:
                for prefix, level, name in self._train_only:
                    if level == 'fine':
                        shade(name, prefix, 'fine', z_fine, f'noise_{prefix[:-1]}_fine')
        if not retraw:
            for level in ('coarse', 'fine'):
                for k in ('z_vals', 'visibility', 'weights'):
                    out.pop(f'{k}_{level}', None)
        return out

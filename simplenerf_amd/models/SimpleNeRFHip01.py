"""MI355X renderer behind the reference's model interface.

``SimpleNeRFHip(configs, model_configs)`` is a ``torch.nn.Module`` whose constructor arguments, parameter names and
shapes, ``forward(input_batch, retraw=False, sec_views_vis=False) -> dict`` signature and output keys are those of
the reference's ``SimpleNeRF`` (src/models/SimpleNeRF01.py:11-75, :108-270), so the reference's Trainer/Tester call
sites (src/Trainer01.py:93,194,328; src/Tester01.py:63) work unchanged.  All arithmetic runs in the HIP library
(simplenerf_amd/csrc) through the C ABI in include/simplenerf_hip.h; PyTorch only owns device memory, the stream and
the parameters.  There is no eager/CPU fallback: CPU tensors or an unbuilt library raise.

Host binding: ``configs['model']['hip_host_binding']`` = ``'torch_ext'`` (default: ``torch.ops.snerf.render``, the
TORCH_LIBRARY op with a C++ ``torch::autograd::Function`` that SURVEY 8b names -- csrc_torch/snerf_torch.cpp, built by
torch.utils.cpp_extension) or ``'ctypes'`` (the torch-free binding of the same two C-ABI calls, ops.RenderCall).  Same
kernels, same results bit for bit; the extension spends ~1 ms less host time per training iteration.

One extra, optional config key: ``configs['model']['hip_precision']`` = ``'fp32'`` (default; fp32 matrix cores) or
``'f16x3'`` (fp16 hi/lo split, three MFMAs per product, fp32 accumulate: fp32-grade results ~2.5x faster; covers the
forward, the activation-keeping training forward, the backward dgrad chain and the large weight-gradient products), or
``'f16'`` (16-bit mode: one fp16 MFMA per product, 16-bit saved activations / layer gradients, fp32 master weights and
accumulation; ~4x faster training than fp32 at ~1e-3 agreement -- outside the fp32 parity bar, tests/test_gpu_f16.py),
``'bf16'`` (the 16-bit mode on bf16 operands: no range limit, three significand bits fewer, tests/test_gpu_bf16.py) or
``'f16s8'`` (the fp16 16-bit mode with the saved trunk activations of 256-wide MLPs kept as fp8 e4m3 for the weight
gradients: rendering and the forward are ``'f16'``'s bit for bit, the training iteration moves 15 % fewer HBM bytes;
``'bf16s8'``: the same for the bf16 mode).
``configs['model']['hip_fused_render']`` = ``True``: eval-mode renders of a plain coarse + fine fp32 model as one launch
(csrc/render_fused.hip; bit-identical, off by default).

Differences from the reference that a caller can observe:
  * ``model.chunk`` / ``model.netchunk`` are accepted and ignored -- the kernels tile the work themselves and the
    results do not depend on chunking;
  * training-mode randomness (stratified jitter, inverse-CDF ``u``, density noise) is drawn on the device by the
    library's counter-based Philox4x32-10 (snerf_random_uniform / snerf_random_normal) instead of on the CPU
    generator: element (ray, sample) of a draw is a function of (``configs['seed']``, training iteration
    ``input_batch['iter_num']``, kind of draw, GLOBAL row of the ray, sample) only, so a rank that holds a part of a
    batch -- or a trainer that cuts it into any number of sub-batches -- draws what a single process would for those rows
    (a batch without ``iter_num`` falls back to the number of training forwards this module has run, which is only
    shard-invariant when every process makes the same number of calls).  The global rows come from ``input_batch['global_rows']`` (int64 (n,), emitted by
    ``BatchAssembler``; a per-row tensor, because a rank's rows are a pixel-ray shard followed by a sparse-depth shard of
    the global batch and because the reference's trainer slices every tensor of the batch into sub-batches,
    src/Trainer01.py:82-90) or, without it, are ``input_batch.get('row_offset', 0)`` + ray.  ``set_random_draws`` injects
    explicit draws (used by the parity tests to replay the reference's CPU stream);
  * ``predict_visibility`` (off in every shipped config): built for view-dependent MLPs in the fp32 mode -- the
    ``raw_visibility_*`` / ``raw_visibility2_*`` / ``visibility2_*`` outputs of src/models/SimpleNeRF01.py:646-649, :479-482
    are produced (``sec_views_vis``, ``rays_o2`` or ``common_data['poses']`` + ``pixel_id`` + ``num_frames`` as in :120-133)
    but carry no gradient (the losses that would read them are not among the shipped ones); the f16 modes raise;
  * gradients flow from ``rgb_*``, ``acc_*``, ``depth_*``, ``depth_ndc_*`` (incl. the augmentation-prefixed ones) and
    ``raw_sigma_*`` / ``raw_rgb*_*`` to the parameters -- a superset of what the shipped losses read (SURVEY 8a row
    9); ``alpha_*``, ``visibility_*``, ``weights_*``, ``depth_var*`` and ``z_vals_*`` are returned without a
    gradient path (the reference detaches the sample depths, :312; no loss reads the others), so a loss built only
    on them raises instead of silently training nothing.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import _torch_ext, ops

Tensor = torch.Tensor
_SKIP_AFTER = 4  # reference: self.skips = [4]
_DRAW_KINDS = ('t_rand', 'u', 'noise_coarse', 'noise_points_augmentation', 'noise_views_augmentation', 'noise_fine',
               'noise_points_augmentation_fine', 'noise_views_augmentation_fine')


def global_rows(batch: dict, n: int):
    """(first_row, rows): the global rows of the batch's rays in the single-process batch, which key the training draws.
    ``batch['global_rows']`` (int64 GPU tensor (n,), e.g. from BatchAssembler -- a per-row tensor, so a trainer that slices
    sub-batches slices it along) wins; else rows are ``batch['row_offset']`` (default 0) + r."""
    rows = batch.get('global_rows')
    if rows is not None:
        if not isinstance(rows, torch.Tensor) or tuple(rows.shape) != (n,):
            raise RuntimeError(f"input_batch['global_rows'] must be an int64 tensor of shape ({n},)")
        return 0, rows
    return int(batch.get('row_offset', 0)), None


class MlpParameters(torch.nn.Module):
    """Parameters of one NeRF MLP under the reference's names, shapes and construction order (MLP.__init__
    src/models/SimpleNeRF01.py:561-609), so ``state_dict()`` round-trips with reference checkpoints and the default
    initialisation consumes the RNG identically."""

    def __init__(self, configs: dict, mlp_configs: dict):
        super().__init__()
        self.mlp_configs = mlp_configs
        self.predict_visibility = bool(mlp_configs.get('predict_visibility', False))
        if self.predict_visibility and not mlp_configs['view_dependent_rgb']:
            raise NotImplementedError('predict_visibility is built for view_dependent_rgb MLPs (a views head with rgb + '
                                      'visibility rows); the visibility-only views head is not')
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
            self.views_output_linear = lin(wv, 4 if self.predict_visibility else 3)   # rgb (+ visibility), :594-602

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
            out += [self.feature_linear.weight, self.feature_linear.bias]
            for layer in self.views_linears:          # (views_net_depth > 1: the layered path, csrc/mlp_generic.hip)
                out += [layer.weight, layer.bias]
            out += [self.views_output_linear.weight, self.views_output_linear.bias]
        return out

    def forward(self, *args, **kwargs):
        raise RuntimeError('MlpParameters only holds weights; evaluation happens in the fused HIP kernel')


# C-ABI level order (include/simplenerf_hip.h, enum snerf_render_level): attribute name and output-key prefix
_LEVELS = (('coarse_model', '', 'coarse'), ('pts_aug_coarse_model', 'points_augmentation_', 'coarse'),
           ('views_aug_coarse_model', 'views_augmentation_', 'coarse'), ('fine_model', '', 'fine'),
           ('pts_aug_fine_model', 'points_augmentation_', 'fine'), ('views_aug_fine_model', 'views_augmentation_', 'fine'))
_NOISE_KEYS = ('noise_coarse', 'noise_points_augmentation', 'noise_views_augmentation', 'noise_fine',
               'noise_points_augmentation_fine', 'noise_views_augmentation_fine')
_DIFF_KEYS = ('rgb', 'acc', 'depth', 'depth_ndc', 'sigma', 'raw_rgb')          # outputs with a gradient path
_PLAIN_KEYS = ('alpha', 'visibility', 'weights', 'depth_var', 'depth_var_ndc',     # returned without one
               'raw_visibility', 'raw_visibility2', 'visibility2')


class _RenderFunction(torch.autograd.Function):
    """The whole of render_rays as ONE autograd node: forward = snerf_render_forward (training variant: activations
    kept), backward = snerf_render_backward (per level: compositing backward K6 -> MLP backward K7).

    Parameter gradients are written by the kernels straight into ``p.grad`` -- overwritten when the parameter has none
    yet, ADDED to it otherwise (the trainer's second sub-batch, src/Trainer01.py:82-96) -- instead of being returned for
    autograd to accumulate with one add launch per tensor; the Function therefore returns None for its parameter inputs.
    ``model.return_param_grads = True`` restores the plain autograd contract (needed for torch.autograd.grad)."""

    @staticmethod
    def forward(ctx, model, call, rays, draws, *params):
        z_coarse, z_fine, out = call.forward(rays, draws)
        ctx.model, ctx.call = model, call
        ctx.layout = []
        flat, plain, plain_layout = [], [], []
        for level in call.levels:
            d = out[level]
            for k in _DIFF_KEYS:
                if k in d:
                    ctx.layout.append((level, k))
                    flat.append(d[k])
            for k in _PLAIN_KEYS:
                if k in d:
                    plain_layout.append((level, k))
                    plain.append(d[k])
        plain += [z_coarse] + ([z_fine] if z_fine is not None and draws.get('z_vals_fine') is None else [])
        call.output_layout = (list(ctx.layout), plain_layout)      # (level, key) of the returned tensors, in order
        ctx.mark_non_differentiable(*plain)
        ctx.set_materialize_grads(False)
        ctx.num_plain = len(plain)
        return tuple(flat) + tuple(plain)

    @staticmethod
    def backward(ctx, *grad_outputs):
        call, model = ctx.call, ctx.model
        grads: Dict[int, Dict[str, Tensor]] = {}
        for (level, key), g in zip(ctx.layout, grad_outputs):
            if g is not None:
                grads.setdefault(level, {})[key] = g
        direct = not model.return_param_grads
        param_grads, accumulate, returned = {}, {}, []
        for level in call.levels:
            params = getattr(model, _LEVELS[level][0]).abi_params()
            if level not in grads or not any(p.requires_grad for p in params):
                returned += [None] * len(params)
                continue
            if direct and all(p.grad is not None for p in params):
                param_grads[level], accumulate[level] = [p.grad for p in params], True
            else:
                param_grads[level] = [torch.empty_like(p, memory_format=torch.contiguous_format) for p in params]
                accumulate[level] = False
            returned += [None] * len(params) if direct else param_grads[level]
        call.backward(grads, param_grads, accumulate)
        if direct:
            for level, tensors in param_grads.items():
                if not accumulate[level]:
                    for p, g in zip(getattr(model, _LEVELS[level][0]).abi_params(), tensors):
                        if p.requires_grad:
                            p.grad = g if p.grad is None else p.grad.add_(g)
        return (None, None, None, None) + tuple(returned)


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
        # eval-mode renders of a plain coarse + fine model as ONE launch with the ray group's sample tile in LDS
        # (csrc/render_fused.hip; bit-identical outputs; calls outside its scope take the stage-by-stage path)
        self.fused_render = bool(mcfg.get('hip_fused_render', False))
        self._packed: Dict[str, tuple] = {}
        self._draws: Optional[dict] = None
        self.seed = int(configs.get('seed', 0))
        self._train_calls = 0   # training-mode forwards so far: the Philox stream of a batch that carries no 'iter_num'
        # False (default): the backward kernels write / accumulate parameter gradients straight into ``p.grad``;
        # True: they are returned to autograd (torch.autograd.grad, hooks), which then accumulates them itself
        self.return_param_grads = bool(mcfg.get('hip_return_param_grads', False))
        self.host_binding = mcfg.get('hip_host_binding', 'torch_ext')
        if self.host_binding not in ('torch_ext', 'ctypes'):
            raise KeyError(f"model.hip_host_binding must be 'torch_ext' or 'ctypes', got {self.host_binding!r}")
        self._desc_ints: Dict[tuple, list] = {}

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

    def draw_training_randomness(self, n: int, row_offset, device, out: Optional[Dict[str, Tensor]] = None,
                                 iter_num: Optional[int] = None, at: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """The draws the next training-mode forward would make itself, produced up front (and counted as that forward's):
        pass the result to ``set_random_draws``.  ``row_offset``: the first global row (int) or the batch's
        ``global_rows`` tensor.  ``out`` supplies pre-allocated tensors to fill (static graph inputs).  ``iter_num``: the
        batch's training iteration (what ``forward`` keys the draws by); None = this module's call counter.  ``at``: the
        device-resident iteration record (``ops.IterationRing.current``) to read the iteration from -- inside a captured graph."""
        call = self._train_calls if iter_num is None else int(iter_num)
        self._train_calls += 1
        noise_std = float(self.configs['model']['raw_noise_std'])
        first, rows = (0, row_offset) if isinstance(row_offset, torch.Tensor) else (int(row_offset), None)
        draws = {}
        for key, shape in self.training_draw_shapes(n).items():
            stream = call * len(_DRAW_KINDS) + _DRAW_KINDS.index(key)
            target = None if out is None else out[key]
            where = None if at is None else (at, _DRAW_KINDS.index(key), len(_DRAW_KINDS))
            if key.startswith('noise'):
                draws[key] = ops.random_normal(self.seed, stream, first, shape, device, noise_std, out=target, rows=rows, at=where)
            else:
                draws[key] = ops.random_uniform(self.seed, stream, first, shape, device, out=target, rows=rows, at=where)
        return draws

    @staticmethod
    def _secondary_origins(batch: dict) -> Tensor:
        """Camera centres of the other training views per ray (render_rays :122-133): for ray r of image i the j-th entry is
        the centre of view j + (j >= i)."""
        common = batch['common_data']
        poses = common['poses']
        if poses.dim() == 4:            # the per-GPU replica axis the trainer's loader adds (forward strips it, :69-72)
            poses = poses[0]
        image_id = batch['pixel_id'][:, 0].long()
        num_frames = int(batch['num_frames'])
        cols = []
        for i in range(num_frames - 1):
            other = i + (i >= image_id).long()
            cols.append(poses[other][:, :3, 3])
        return torch.stack(cols, dim=1).contiguous()

    def _packed_mlp(self, name: str, keeps_activations: bool) -> ops.PackedMlp:
        module: MlpParameters = getattr(self, name)
        params = module.abi_params()
        # staleness stamp: every parameter's in-place version counter (optimiser steps, load_state_dict, manual edits)
        # plus the storage address of the first one (a .to(device) / re-materialisation moves them all)
        # ... and what the stream was packed FOR: the optimiser step of every training iteration re-packs, and writing only
        # the operand formats this precision reads in this mode (snerf_mlp_pack_for) leaves 9 of 25 pack launches
        # (``keeps_activations``: this call saves activations for a backward -- the storing forward and the backward read the
        # training layout, a plain render reads the rendering layout as well)
        stamp = (params[0].data_ptr(), self.precision, bool(keeps_activations)) + tuple(p._version for p in params)
        entry = self._packed.get(name)
        if entry is None or entry[0] != stamp or entry[1].buffer.device != params[0].device:
            packed = entry[1] if entry is not None and entry[1].buffer.device == params[0].device \
                else ops.PackedMlp(module.mlp_configs, params[0].device)
            packed.pack(params, self.precision, bool(keeps_activations))
            self._packed[name] = (stamp, packed)
        return self._packed[name][1]

    # ------------------------------------------------------------------------------------------
    def _render_with_ctypes(self, batch, draws, present, packed, s_c, s_f, per_sample, with_grad):
        """ops.RenderCall (ctypes over the C ABI) + the Python autograd.Function."""
        mcfg = self.configs['model']
        call = ops.RenderCall(packed, self.ndc, bool(mcfg['white_bkgd']), bool(mcfg['lindisp']), s_c, s_f, self.precision,
                              keep_activations=with_grad, per_sample=per_sample, fused=self.fused_render)
        if not with_grad:
            return call.forward(batch, draws)
        params = [p for name in present if name for p in getattr(self, name).abi_params()]
        res = _RenderFunction.apply(self, call, batch, draws, *params)
        out_levels, it = {level: {} for level in call.levels}, iter(res)
        for layout in call.output_layout:
            for level, k in layout:
                out_levels[level][k] = next(it)
        z_coarse = next(it)
        z_fine = draws['z_vals_fine'] if 'z_vals_fine' in draws else (next(it) if self.fine_mlp_needed else None)
        return z_coarse, z_fine, out_levels

    _EXT_KEYS = ('rgb', 'acc', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc', 'alpha', 'visibility', 'weights', 'sigma',
                 'raw_rgb', 'raw_visibility', 'raw_visibility2', 'visibility2')     # order of `enum Key` in snerf_torch.cpp

    def _render_with_extension(self, batch, draws, present, packed, s_c, s_f, per_sample, with_grad):
        """torch.ops.snerf.render: one dispatcher call; validation, output allocation, pointer tables and the autograd node
        live in C++ (csrc_torch/snerf_torch.cpp)."""
        snerf = _torch_ext.load()
        mcfg = self.configs['model']
        key = tuple(present)
        descs = self._desc_ints.get(key)
        if descs is None:
            descs = []
            for mlp in packed:
                d = mlp.desc if mlp is not None else None
                descs += [0] * 10 if d is None else [d.points_net_depth, d.points_net_width, d.views_net_depth, d.views_net_width,
                                                      d.points_pe_degree, d.views_pe_degree, d.sigma_pe_degree, d.use_view_dirs,
                                                      d.view_dependent_rgb, d.predict_visibility]
            self._desc_ints[key] = descs
        mask = (1 if 'alpha' in per_sample else 0) | (2 if 'visibility' in per_sample else 0) | (4 if 'weights' in per_sample else 0)
        cfg = [int(self.ndc), int(bool(mcfg['white_bkgd'])), int(bool(mcfg['lindisp'])), s_c, s_f, self.precision, mask,
               int(self.fused_render)]
        near, far = (batch['near_ndc'], batch['far_ndc']) if self.ndc else (batch['near'], batch['far'])
        predicts = any(m is not None and m.desc.predict_visibility for m in packed)
        rays = [batch['rays_o'], batch['rays_d'], batch.get('view_dirs'), batch.get('rays_o_ndc') if self.ndc else None,
                batch.get('rays_d_ndc') if self.ndc else None, near, far, batch.get('rays_o2') if predicts else None]
        draw_list = [draws.get('t_rand'), draws.get('u')] + [draws.get(('noise', l)) for l in range(6)] + [draws.get('z_vals_fine')]
        params, counts = [], [0] * 6
        if with_grad:
            for l, name in enumerate(present):
                if name:
                    level_params = getattr(self, name).abi_params()
                    params += level_params
                    counts[l] = len(level_params)
        try:
            res = snerf.render(cfg, descs, [None if m is None else m.buffer for m in packed], rays, draw_list, params, counts,
                               with_grad, self.return_param_grads)
        except RuntimeError as error:        # (c10::Error; the range report keeps its own type through either binding)
            if 'outside the fp16 range' in str(error):
                raise ops.Fp16RangeError(str(error).split('\n')[0]) from None
            raise
        it = iter(res)
        z_coarse = next(it)
        z_fine = next(it) if self.fine_mlp_needed else None
        k_other = rays[7].shape[1] if rays[7] is not None else 0
        out_levels: Dict[int, Dict[str, Tensor]] = {}
        for l, mlp in enumerate(packed):
            if mlp is None:
                continue
            d = out_levels[l] = {}
            vis = bool(mlp.desc.predict_visibility)
            for k in self._EXT_KEYS:
                if k in ('depth_ndc', 'depth_var_ndc') and not self.ndc:
                    continue
                if k in ('alpha', 'visibility', 'weights') and k not in per_sample:
                    continue
                if k == 'raw_visibility' and not vis:
                    continue
                if k in ('raw_visibility2', 'visibility2') and not (vis and k_other):
                    continue
                d[k] = next(it)
        return z_coarse, z_fine, out_levels

    # ------------------------------------------------------------------------------------------
    def forward(self, input_batch: dict, retraw: bool = False, sec_views_vis: bool = False) -> Dict[str, Tensor]:
        batch = dict(input_batch)  # the caller's dict is never mutated (reference: deep_dict_copy :68)
        training = self.training
        retraw = retraw or training
        with_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        mcfg = self.configs['model']
        rays_o = batch['rays_o']
        n = rays_o.shape[0]
        dev = rays_o.device
        injected = self._draws
        self._draws = None
        noise_std = float(mcfg['raw_noise_std'])
        perturb = bool(mcfg['perturb'] > 0.) and training
        first_row, rows = global_rows(batch, n) if training else (0, None)
        # the Philox stream: the batch's training iteration (shard- and sub-batch-invariant together with the global
        # rows); a batch without one falls back to this module's count of training forwards
        call_index = self._train_calls if batch.get('iter_num') is None else int(batch['iter_num'])
        if training and injected is None:
            self._train_calls += 1

        # which of the six MLPs take part: the augmentation models run in training mode only (:170-199, :234-263)
        present = [None] * 6
        present[0] = 'coarse_model'
        if self.fine_mlp_needed:
            present[3] = 'fine_model'
        if training:
            for prefix, level, name in self._train_only:
                if level == 'coarse' or self.fine_mlp_needed:
                    present[[row[0] for row in _LEVELS].index(name)] = name
        s_c = mcfg['coarse_mlp']['num_samples']
        s_f = mcfg['fine_mlp']['num_samples'] if self.fine_mlp_needed else 0

        def draw(key, shape, normal):
            if not training:
                return None
            if injected is not None:
                t = injected.get(key)
                return None if t is None else t.to(dev)
            stream = call_index * len(_DRAW_KINDS) + _DRAW_KINDS.index(key)
            if normal:
                return ops.random_normal(self.seed, stream, first_row, shape, dev, noise_std, rows=rows) if noise_std > 0. else None
            return ops.random_uniform(self.seed, stream, first_row, shape, dev, rows=rows) if perturb else None

        # draws in the reference's consumption order (A.3): jitter, coarse-level noises, u, fine-level noises
        draws: dict = {'t_rand': draw('t_rand', (n, s_c), False)}
        for level in (0, 1, 2):
            if present[level]:
                draws[('noise', level)] = draw(_NOISE_KEYS[level], (n, s_c, 1), True)
        if self.fine_mlp_needed:
            override = injected.get('z_vals_fine') if injected is not None else None
            if override is not None:
                draws['z_vals_fine'] = override.to(dev)
            else:
                draws['u'] = draw('u', (n, s_f), False)
            for level in (3, 4, 5):
                if present[level]:
                    draws[('noise', level)] = draw(_NOISE_KEYS[level], (n, s_c + s_f, 1), True)

        # (Re-packing the stale streams side by side on forked torch streams -- they are independent, ~25 us of small dependent
        # launches each -- was built and measured in round 5: neutral at 512 rows, and 5 % SLOWER on the graphed 4096-row
        # iteration, whose graph then starts with four parallel branches; profiles/r05_share_ab.jsonl.  They stay in order.)
        packed = [self._packed_mlp(name, with_grad) if name else None for name in present]
        per_sample = ('alpha', 'visibility', 'weights') if retraw else ('alpha',)
        predicts = [name is not None and getattr(self, name).predict_visibility for name in present]
        if any(predicts):
            if self.precision != ops.PRECISION_FP32:
                raise NotImplementedError("predict_visibility is built for hip_precision='fp32' only")
            batch.pop('rays_o2', None) if not (sec_views_vis or training) else None
            if (sec_views_vis or training) and 'rays_o2' not in batch:
                batch['rays_o2'] = self._secondary_origins(batch)          # (n, num_frames - 1, 3), :122-133
        if self.host_binding == 'torch_ext':
            z_coarse, z_fine, out_levels = self._render_with_extension(batch, draws, present, packed, s_c, s_f, per_sample, with_grad)
        else:
            z_coarse, z_fine, out_levels = self._render_with_ctypes(batch, draws, present, packed, s_c, s_f, per_sample, with_grad)

        out: Dict[str, Tensor] = {}

        def emit(level):
            _, prefix, tag = _LEVELS[level]
            d = out_levels[level]
            # key order as volume_rendering's return_dict (:465-477)
            for k in ('rgb', 'acc', 'alpha', 'visibility', 'weights', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc'):
                if k in d:
                    out[f'{prefix}{k}_{tag}'] = d[k]
            if 'visibility2' in d:
                out[f'{prefix}visibility2_{tag}'] = d['visibility2']               # volume_rendering's last key (:479-482)
            if retraw:
                # key order of MLP.forward's output_batch (:626-654): sigma, rgb_view_*, visibility, visibility2, rgb
                out[f'{prefix}raw_sigma_{tag}'] = d['sigma']
                variant = 'rgb_view_dependent' if packed[level].use_view_dirs else 'rgb_view_independent'
                out[f'{prefix}raw_{variant}_{tag}'] = d['raw_rgb']
                if 'raw_visibility' in d:
                    out[f'{prefix}raw_visibility_{tag}'] = d['raw_visibility']
                if 'raw_visibility2' in d:
                    out[f'{prefix}raw_visibility2_{tag}'] = d['raw_visibility2']
                out[f'{prefix}raw_rgb_{tag}'] = d['raw_rgb']

        emit(0)
        out['z_vals_coarse'] = z_coarse
        for level in (1, 2):
            if present[level]:
                emit(level)
        if self.fine_mlp_needed:
            emit(3)
            out['z_vals_fine'] = z_fine
            for level in (4, 5):
                if present[level]:
                    emit(level)
        if not retraw:
            for level in ('coarse', 'fine'):
                for k in ('z_vals', 'visibility', 'weights'):
                    out.pop(f'{k}_{level}', None)
        return out

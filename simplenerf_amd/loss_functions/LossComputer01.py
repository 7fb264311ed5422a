"""Training losses evaluated by the HIP library, behind the reference's ``LossComputer`` interface.

``LossComputer(configs).compute_losses(input_dict, output_dict, return_loss_maps=False)`` takes and returns what the
reference's class does (src/loss_functions/LossComputer01.py:12-53): a dictionary ``{loss name: {'loss_value': t},
'TotalLoss': t}`` whose ``TotalLoss.backward()`` drives the training step (src/Trainer01.py:93-96).  The nine loss
classes every shipped experiment enables are built: MSE01-03, SparseDepthMSE01-03, PointsAugmentationDepthLoss02,
ViewsAugmentationDepthLoss02, CoarseFineConsistencyLoss02; any other name raises at construction, as an unknown
module does in the reference (:24-31).

Instead of one Python object and a few dozen torch kernels per loss, every loss contributes rows to ONE table of
masked mean-squared-error terms; the table is evaluated by one launch (snerf_loss_forward) and differentiated by one
launch (snerf_loss_backward); the three patch-consistency losses add one launch each for their decision masks
(snerf_patch_consistency_masks).  See include/simplenerf_train.h.  There is no torch fallback.

Reference behaviour kept on purpose (pinned by tests/golden/losses_*.npz):
  * a mean over zero rays is 0, not NaN (MSE01.py:62);
  * SparseDepthMSE02/03 read ``depth_fine`` of the MAIN model when the augmentation has a fine MLP
    (SparseDepthMSE02.py:44);
  * of the two symmetric terms of a consistency loss only the one on the first estimate (main / coarse model) is
    non-zero: the reference computes the second from tensors its first call zeroed in place through ``detach()``
    aliases (PointsAugmentationDepthLoss02.py:172-173, :205-207), so it is identically zero in value and gradient.
    The augmented / fine depth therefore receives no gradient from these losses, here as there;
  * ``compute_losses`` replaces the tensors of ``input_dict['common_data']`` by their first (per-GPU) replica in
    place (LossComputer01.py:34-38).
Loss maps (``return_loss_maps=True``, validation only) are assembled with torch ops from the kernel's masks -- off
the training path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops

Tensor = torch.Tensor
_AUGMENTATION = {'01': '', '02': 'points_augmentation', '03': 'views_augmentation'}
_PATCH_LOSSES = {'PointsAugmentationDepthLoss02': 'points_augmentation', 'ViewsAugmentationDepthLoss02': 'views_augmentation',
                 'CoarseFineConsistencyLoss02': None}
SUPPORTED = tuple(f'{stem}{nn}' for stem in ('MSE', 'SparseDepthMSE') for nn in _AUGMENTATION) + tuple(_PATCH_LOSSES)


class _FusedLossFunction(torch.autograd.Function):
    """values = snerf_loss_forward(table);  d pred = snerf_loss_backward(table, d values)."""

    @staticmethod
    def forward(ctx, terms, num_groups, owner, *unique_preds):
        values, scales = ops.loss_forward(terms, num_groups)
        ctx.terms, ctx.num_groups, ctx.owner, ctx.scales = terms, num_groups, owner, scales
        return values

    @staticmethod
    def backward(ctx, grad_values):
        needs = ctx.needs_input_grad[3:]
        wanted = [needs[j] for j in ctx.owner]
        grads = ops.loss_backward(ctx.terms, ctx.num_groups, ctx.scales, grad_values.contiguous(), wanted)
        per_input: List[Optional[Tensor]] = [None] * len(needs)
        for g, j in zip(grads, ctx.owner):
            if g is not None and per_input[j] is None:
                per_input[j] = g          # terms sharing a pred share one buffer that already holds their sum
        return (None, None, None) + tuple(per_input)


class LossComputer:
    def __init__(self, configs: dict):
        self.configs = configs
        self.losses: Dict[str, dict] = {}
        for loss_configs in configs['losses']:
            name = loss_configs['name']
            if name not in SUPPORTED:
                raise RuntimeError(f'Unknown Loss Function: {name} (the HIP loss evaluation builds {", ".join(SUPPORTED)})')
            self.losses[name] = loss_configs
        if len(self.losses) > 16:
            raise RuntimeError('at most 16 losses fit the fused loss table')

    @staticmethod
    def get_loss_weight(loss_configs: dict, iter_num: int):
        """Constant ``weight`` or the ``iter_weights`` entry with the largest start <= iter_num (reference :55-69)."""
        weight = None
        if 'weight' in loss_configs:
            weight = loss_configs['weight']
        elif 'iter_weights' in loss_configs:
            for start in sorted((int(k) for k in loss_configs['iter_weights']), reverse=True):
                if iter_num >= start:
                    weight = loss_configs['iter_weights'][str(start)]
                    break
        if weight is None:
            raise RuntimeError(f"loss_weight is None for {loss_configs['name']} at iter {iter_num}")
        return weight

    # --------------------------------------------------------------------------------------------------
    def compute_losses(self, input_dict: dict, output_dict: dict, return_loss_maps: bool = False) -> dict:
        if 'common_data' in input_dict:
            common = input_dict['common_data']
            for key in common:
                if isinstance(common[key], torch.Tensor):
                    common[key] = common[key][0]
        model = self.configs['model']
        iter_num = input_dict['iter_num']
        mask_nerf = input_dict['indices_mask_nerf']
        mask_sd = input_dict.get('indices_mask_sparse_depth')
        terms: List[ops.LossTermSpec] = []
        preds: List[Tensor] = []          # the caller's tensors (with their autograd history), one per term
        maps: Dict[str, Dict[str, Tensor]] = {name: {} for name in self.losses}

        def add(group, weight, pred, target, numerator, denominator):
            terms.append(ops.LossTermSpec(pred, target, numerator, denominator, group, weight))
            preds.append(pred)

        for group, (name, loss_configs) in enumerate(self.losses.items()):
            weight = self.get_loss_weight(loss_configs, iter_num)
            if name in _PATCH_LOSSES:
                self._patch_terms(name, loss_configs, group, weight, input_dict, output_dict, mask_nerf, mask_sd, add,
                                  maps[name] if return_loss_maps else None)
                continue
            aug = _AUGMENTATION[name[-2:]]
            section = model[aug] if aug else model
            prefix = f'{aug}_' if aug else ''
            if name.startswith('MSE'):
                for level in ('coarse', 'fine'):
                    key = f'{prefix}rgb_{level}'
                    if f'{level}_mlp' in section and (not aug or key in output_dict):
                        add(group, weight, output_dict[key], input_dict['target_rgb'], mask_nerf, mask_nerf)
                        if return_loss_maps:
                            err = output_dict[key][mask_nerf] - input_dict['target_rgb'][mask_nerf]
                            maps[name][f'{name}_{level}'] = torch.mean(torch.square(err), dim=1)
            elif mask_sd is not None:     # SparseDepthMSE: only for batches that carry sparse-depth rays
                key = 'depth_fine' if 'fine_mlp' in section else f'{prefix}depth_coarse'
                add(group, weight, output_dict[key], input_dict['sparse_depth_values'][:, 0], mask_sd, mask_sd)

        num_groups = len(self.losses)
        device = input_dict['rays_o'].device
        if terms:
            unique: List[Tensor] = []
            owner: List[int] = []
            for p in preds:
                for j, q in enumerate(unique):
                    if q is p:
                        owner.append(j)
                        break
                else:
                    owner.append(len(unique))
                    unique.append(p)
            values = _FusedLossFunction.apply(terms, num_groups, owner, *unique)
            count = len(terms)
        else:
            values = torch.zeros((num_groups + 1,), dtype=torch.float32, device=device)
            count = 0
        loss_values: Dict[str, object] = {}
        for group, name in enumerate(self.losses):
            loss_values[name] = {'loss_value': values[count + group]}
            if return_loss_maps:
                loss_values[name]['loss_maps'] = maps[name]
        loss_values['TotalLoss'] = values[count + num_groups]
        return loss_values

    # --------------------------------------------------------------------------------------------------
    def _patch_terms(self, name, loss_configs, group, weight, input_dict, output_dict, mask_nerf, mask_sd, add, maps):
        model = self.configs['model']
        aug = _PATCH_LOSSES[name]
        if aug is None:
            if 'coarse_mlp' not in model or 'fine_mlp' not in model:
                return
            pairs = [('depth_coarse', 'depth_fine', f'{name}_coarse', f'{name}_fine')]
        else:
            pairs = [(f'depth_{level}', f'{aug}_depth_{level}', f'{name}_{level}_main', f'{name}_{level}_augmented')
                     for level in ('coarse', 'fine') if f'{level}_mlp' in model and f'{level}_mlp' in model[aug]]
        common = input_dict['common_data']
        h, w = common['resolution']
        if tuple(common['images'].shape[1:3]) != (int(h), int(w)):
            raise RuntimeError(f"common_data images {tuple(common['images'].shape)} do not match resolution {(h, w)}")
        for key1, key2, map1_name, map2_name in pairs:
            depth1, depth2 = output_dict[key1], output_dict[key2]
            _, better2 = ops.patch_consistency_masks(
                input_dict['rays_o'], input_dict['rays_d'], depth1, depth2, mask_nerf, input_dict['pixel_id'],
                common['poses'], common['intrinsics'][0], common['images'], loss_configs['patch_size'],
                loss_configs['rmse_threshold'])
            # estimate 1 is pulled towards estimate 2 where 2 reprojects better; mean over ALL pixel rays
            add(group, weight, depth1, depth2, better2, mask_nerf)
            if maps is not None:
                keep = better2[mask_nerf].to(depth1.dtype)
                maps[map1_name] = torch.square((depth1[mask_nerf] - depth2[mask_nerf].detach()) * keep)
                maps[map2_name] = torch.zeros_like(maps[map1_name])
        if aug is None and 'sparse_depth' in self.configs['data_loader'] and mask_sd is not None:
            add(group, weight, output_dict['depth_coarse'], output_dict['depth_fine'], mask_sd, mask_sd)

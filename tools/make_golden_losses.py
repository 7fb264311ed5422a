#!/usr/bin/env python3
"""G8: golden vectors for the loss row, made by RUNNING THE REFERENCE's loss classes here.

    PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden_losses.py

Inputs come from the build-owned generators ``synth.synth_scene`` / ``synth.loss_batch`` (seeds recorded in the
fixture); the fixture stores the reference's loss values, the gradient of ``TotalLoss`` with respect to every model
output, the per-ray loss maps and the un-rounded reprojected pixel positions (``CommonUtils.reproject``).

Cases:  full   9 shipped losses, iteration 20000 (consistency losses weighted 0.1), 448 pixel rays + 64 sparse rays
        early  same batch at iteration 0 (consistency weights 0)
        nosd   no sparse depth in the batch or the config, 512 pixel rays, another seed
        empty  pixel-ray mask all false (8 sparse rays only)
"""
import os
import sys
import types

import numpy
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, 'src'))
for name in ('skimage', 'skimage.io', 'skimage.transform'):
    sys.modules.setdefault(name, types.ModuleType(name))

from loss_functions.LossComputer01 import LossComputer  # noqa: E402  (the reference)
from utils import CommonUtils01  # noqa: E402  (the reference)

from simplenerf_amd import synth  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
OUTPUT_KEYS = ('rgb_coarse', 'rgb_fine', 'points_augmentation_rgb_coarse', 'views_augmentation_rgb_coarse',
               'depth_coarse', 'depth_fine', 'points_augmentation_depth_coarse', 'views_augmentation_depth_coarse')


def run_case(name, scene_seed, batch_seed, num_rays, num_sparse, iter_num, sparse_in_batch=True):
    scene = synth.synth_scene(scene_seed)
    batch = synth.loss_batch(scene, num_rays, num_sparse, batch_seed)
    configs = synth.make_configs('config3')
    configs['losses'] = synth.loss_configs()
    if sparse_in_batch:
        configs['data_loader']['sparse_depth'] = {}
    t = lambda a: torch.from_numpy(numpy.ascontiguousarray(a))
    input_dict = {
        'iter_num': iter_num,
        'rays_o': t(batch['rays_o']), 'rays_d': t(batch['rays_d']), 'pixel_id': t(batch['pixel_id']),
        'target_rgb': t(batch['target_rgb']), 'indices_mask_nerf': t(batch['indices_mask_nerf']),
        # the loader replicates the shared tensors once per GPU; compute_losses takes [0] (LossComputer01.py:34-38)
        'common_data': {'poses': t(scene['poses'])[None], 'images': t(scene['images'])[None],
                        'intrinsics': t(scene['intrinsics'])[None], 'resolution': scene['resolution']},
    }
    if sparse_in_batch:
        input_dict['indices_mask_sparse_depth'] = t(batch['indices_mask_sparse_depth'])
        input_dict['sparse_depth_values'] = t(batch['sparse_depth_values'])
    output_dict = {k: t(batch[k]).clone().requires_grad_(True) for k in OUTPUT_KEYS}
    # with sparse depth in the batch the reference's own loss-map path raises (CoarseFineConsistencyLoss02.py:90 adds a
    # (num_sparse,) map to a (num_rays,) one), so maps are only recorded for the batch without sparse rays
    losses = LossComputer(configs).compute_losses(input_dict, output_dict, return_loss_maps=not sparse_in_batch)
    total = losses['TotalLoss']
    arrays = {'scene_seed': scene_seed, 'batch_seed': batch_seed, 'num_rays': num_rays, 'num_sparse': num_sparse,
              'iter_num': iter_num, 'sparse_in_batch': sparse_in_batch, 'TotalLoss': float(total)}
    if isinstance(total, torch.Tensor) and total.requires_grad:
        total.backward()
    for k in OUTPUT_KEYS:
        g = output_dict[k].grad
        arrays[f'grad_{k}'] = (g if g is not None else torch.zeros_like(output_dict[k])).numpy()
    for loss_name, entry in losses.items():
        if loss_name == 'TotalLoss':
            continue
        arrays[f'value_{loss_name}'] = float(entry['loss_value'])
        for map_name, loss_map in entry.get('loss_maps', {}).items():
            arrays[f'map_{loss_name}_{map_name}'] = loss_map.detach().numpy()
    # reprojected (un-rounded) pixel positions of the main coarse depth into each ray's nearest other view
    with torch.no_grad():
        mask = input_dict['indices_mask_nerf']
        poses = t(scene['poses'])
        origins = poses[:, :3, 3]
        image_ids = t(batch['pixel_id'])[:, 0].long()
        dist = torch.sqrt(torch.sum(torch.square(origins[image_ids][:, None] - origins[None]), dim=2))
        closest = torch.kthvalue(dist, 2, dim=1)[1]
        arrays['closest_view'] = closest.numpy()
        if int(mask.sum()) > 1:   # reproject() squeezes; a single ray would lose its batch axis
            pts = t(batch['rays_o'])[mask] + t(batch['rays_d'])[mask] * t(batch['depth_coarse'])[mask][:, None]
            arrays['reprojected_depth_coarse'] = CommonUtils01.reproject(pts, poses[closest[mask]], t(scene['intrinsics'])).numpy()
    path = os.path.join(OUT, f'losses_{name}.npz')
    numpy.savez_compressed(path, **arrays)
    print(f'losses_{name}.npz: {os.path.getsize(path) / 1024:.0f} KiB; TotalLoss {float(total):.6f}; '
          + ', '.join(f"{k[6:]}={float(v):.5f}" for k, v in arrays.items() if k.startswith('value_')))


if __name__ == '__main__':
    torch.set_num_threads(8)
    run_case('full', 0, 0, 448, 64, 20000)
    run_case('early', 0, 0, 448, 64, 0)
    run_case('nosd', 1, 3, 512, 0, 20000, sparse_in_batch=False)
    run_case('empty', 0, 5, 0, 8, 20000)

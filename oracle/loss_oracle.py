"""CPU restatement (torch, fp32) of the reference's training losses -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product path (simplenerf_amd/) never does.  Pinned: ``tests/test_oracle_golden.py`` holds every function here to the
G8 fixtures (``tests/golden/losses_*.npz``), which ``tools/make_golden_losses.py`` produced by running the
reference's own loss classes in the build container.

Each function cites the reference lines it follows (all under /root/reference/src).  The restatement is functional
(no in-place masking, one vectorised patch gather instead of the reference's 75 indexed assignments) but keeps the
reference's arithmetic, including its quirks:
  * a masked mean over zero rays is the integer 0 (MSE01.py:62, SparseDepthMSE01.py:63);
  * the patch-consistency depth error is averaged over ALL pixel rays of the batch, not over the rays the mask
    keeps (compute_depth_mse zeroes the excluded entries and then takes the plain mean,
    PointsAugmentationDepthLoss02.py:205-211);
  * of the two symmetric depth terms only the first survives: the second is computed from tensors the first call
    zeroed in place through ``detach()`` aliases, and comes out identically zero (see consistency_loss);
  * every ray is reprojected with the FIRST view's intrinsic matrix (utils/CommonUtils01.py:66);
  * the ground-truth images are zero-padded on the bottom/right only and negative patch coordinates wrap round into
    that padding (PointsAugmentationDepthLoss02.py:152-158), which amounts to zero padding on all four sides for the
    square patches every shipped config uses.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor


def loss_weight(loss_configs: dict, iter_num: int) -> float:
    """LossComputer.get_loss_weight (loss_functions/LossComputer01.py:55-69): constant ``weight`` or the entry of
    ``iter_weights`` with the largest key <= iter_num."""
    if 'weight' in loss_configs:
        return loss_configs['weight']
    for start in sorted((int(k) for k in loss_configs['iter_weights']), reverse=True):
        if iter_num >= start:
            return loss_configs['iter_weights'][str(start)]
    raise RuntimeError(f'no loss weight for iteration {iter_num}')


def masked_mse(pred: Tensor, target: Tensor, mask: Tensor):
    """MSE.compute_mse (loss_functions/MSE01.py:57-67) for (N,3) colours and SparseDepthMSE.compute_depth_loss
    (loss_functions/SparseDepthMSE01.py:58-71) for (N,) depths: mean squared error over the rays ``mask`` keeps."""
    error = pred[mask] - target[mask]
    if error.numel() == 0:
        return 0
    return torch.mean(torch.square(error))   # mean over channels then rays == mean over all kept elements (equal counts)


def closest_other_view(poses: Tensor, image_ids: Tensor) -> Tensor:
    """Index of the nearest other camera for each ray's source view (PointsAugmentationDepthLoss02.py:128-132):
    second-smallest origin distance (the smallest is the view itself at 0)."""
    origins = poses[:, :3, 3]
    dist = torch.sqrt(torch.sum(torch.square(origins[:, None] - origins[None]), dim=2))   # (V,V): same rows the
    order = torch.argsort(dist, dim=1, stable=True)                                        # reference builds per ray
    return order[:, 1][image_ids]


def reproject(points: Tensor, poses: Tensor, intrinsic: Tensor) -> Tensor:
    """CommonUtils.reproject (utils/CommonUtils01.py:45-71): pixel position (x,y) of world points in the cameras
    ``poses`` (camera-to-world, y/z flipped convention), using ONE 3x3 intrinsic for all."""
    flip = torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    rel = points - poses[:, :3, 3]
    cam = (intrinsic[None] @ flip[None] @ poses[:, :3, :3].transpose(1, 2) @ rel[..., None])[..., 0]
    return cam[:, :2] / cam[:, 2:]


def gather_patches(images: Tensor, image_ids: Tensor, x: Tensor, y: Tensor, half_x: int, half_y: int) -> Tensor:
    """(n, 2*half_y+1, 2*half_x+1, C) patches centred on (x,y), zero outside the image
    (PointsAugmentationDepthLoss02.py:148-158; see the module docstring for the padding equivalence)."""
    v, h, w, c = images.shape
    padded = torch.zeros((v, h + 2 * half_y, w + 2 * half_x, c), dtype=images.dtype)
    padded[:, half_y:half_y + h, half_x:half_x + w] = images
    oy = torch.arange(0, 2 * half_y + 1)[None, :, None]
    ox = torch.arange(0, 2 * half_x + 1)[None, None, :]
    return padded[image_ids[:, None, None], y[:, None, None] + oy, x[:, None, None] + ox]


def patch_rmse(a: Tensor, b: Tensor) -> Tensor:
    """compute_patch_rmse (PointsAugmentationDepthLoss02.py:181-194)."""
    return torch.sqrt(torch.mean(torch.square(a - b), dim=(1, 2, 3)))


def consistency_masks(depth1: Tensor, depth2: Tensor, rays_o: Tensor, rays_d: Tensor, pixel_id: Tensor, poses: Tensor,
                      images: Tensor, intrinsics: Tensor, resolution: Tuple[int, int], patch_size, rmse_threshold: float
                      ) -> Dict[str, Tensor]:
    """The decision part of compute_loss_nerf (PointsAugmentationDepthLoss02.py:119-169) for rays that are already
    restricted to the pixel-ray mask: which of two depth estimates reprojects onto the better-matching patch in the
    nearest other view.  mask1[i] = estimate 1 is the more accurate one (and is then used to supervise estimate 2)."""
    h, w = resolution
    px, py = patch_size
    hx, hy = px // 2, py // 2
    pixel_id = pixel_id.long()
    image_a, x_a, y_a = pixel_id[:, 0], pixel_id[:, 1], pixel_id[:, 2]
    image_b = closest_other_view(poses, image_a)
    poses_b = poses[image_b]

    def land(depth):
        pos = reproject((rays_o + rays_d * depth[:, None]).detach(), poses_b, intrinsics[0]).round().long()
        x, y = pos[:, 0], pos[:, 1]
        valid = (x >= hx) & (x < w - hx) & (y >= hy) & (y < h - hy)
        return torch.clip(x, 0, w - 1), torch.clip(y, 0, h - 1), valid

    x1, y1, valid1 = land(depth1)
    x2, y2, valid2 = land(depth2)
    valid_a = (x_a >= hx) & (x_a < w - hx) & (y_a >= hy) & (y_a < h - hy)
    patch_a = gather_patches(images, image_a, x_a, y_a, hx, hy)
    rmse1 = patch_rmse(patch_a, gather_patches(images, image_b, x1, y1, hx, hy))
    rmse2 = patch_rmse(patch_a, gather_patches(images, image_b, x2, y2, hx, hy))
    mask1 = ((rmse1 < rmse2) | ~valid2) & (rmse1 < rmse_threshold) & valid1 & valid_a
    mask2 = ((rmse2 < rmse1) | ~valid1) & (rmse2 < rmse_threshold) & valid2 & valid_a
    return {'mask1': mask1, 'mask2': mask2, 'rmse1': rmse1, 'rmse2': rmse2, 'closest_view': image_b}


def consistency_loss(depth1: Tensor, depth2: Tensor, ray_mask: Tensor, rays_o: Tensor, rays_d: Tensor, pixel_id: Tensor,
                     poses: Tensor, images: Tensor, intrinsics: Tensor, resolution, patch_size, rmse_threshold: float):
    """compute_loss_nerf (PointsAugmentationDepthLoss02.py:100-175; the Views / CoarseFine variants are the same
    function): each estimate is pulled towards the other (detached) wherever the other one is the more accurate."""
    d1, d2 = depth1[ray_mask], depth2[ray_mask]
    if d1.numel() == 0:
        return torch.tensor(0.0), None
    m = consistency_masks(d1, d2, rays_o[ray_mask], rays_d[ray_mask], pixel_id[ray_mask], poses, images, intrinsics,
                          resolution, patch_size, rmse_threshold)
    keep2, keep1 = m['mask2'].to(d1.dtype), m['mask1'].to(d1.dtype)
    map1 = torch.square(d1 * keep2 - d2.detach() * keep2)    # loss on estimate 1, where estimate 2 is better
    # The second call of compute_depth_mse (:173) sees tensors the first call already zeroed IN PLACE outside mask2:
    # ``depth2.detach()`` shares storage with depth2 and ``pred_depth[~mask] = 0`` (:206-207) writes through it.  Both
    # operands therefore carry mask2 AND mask1, which are disjoint by construction, so this term and its gradient
    # are identically zero in the reference: only estimate 1 is ever supervised.
    both = keep1 * keep2
    map2 = torch.square(d2 * both - d1.detach() * both)
    m['map1'], m['map2'] = map1, map2
    return torch.mean(map1) + torch.mean(map2), m


def compute_losses(configs: dict, input_dict: dict, output_dict: dict) -> Dict[str, Tensor]:
    """LossComputer.compute_losses (loss_functions/LossComputer01.py:33-53) over the loss classes the shipped
    experiments enable: MSE01-03, SparseDepthMSE01-03, Points/ViewsAugmentationDepthLoss02,
    CoarseFineConsistencyLoss02.  ``input_dict['common_data']`` holds the un-replicated tensors."""
    model = configs['model']
    mask_nerf = input_dict['indices_mask_nerf']
    mask_sd = input_dict.get('indices_mask_sparse_depth')
    common = input_dict.get('common_data', {})
    values: Dict[str, Tensor] = {}
    total = 0
    zero = torch.tensor(0.0)

    def aug_key(name):   # 'points_augmentation' / 'views_augmentation' / '' by the two-digit suffix
        return {'01': '', '02': 'points_augmentation', '03': 'views_augmentation'}[name[-2:]]

    for loss_configs in configs['losses']:
        name = loss_configs['name']
        stem = name[:-2]
        value = zero.clone()
        if stem == 'MSE':                                                     # MSE01.py:25-55, MSE02/03 likewise
            aug = aug_key(name)
            section = model[aug] if aug else model
            prefix = f'{aug}_' if aug else ''
            for level in ('coarse', 'fine'):
                key = f'{prefix}rgb_{level}'
                if f'{level}_mlp' in section and key in output_dict:
                    value = value + masked_mse(output_dict[key], input_dict['target_rgb'], mask_nerf)
        elif stem == 'SparseDepthMSE':                                        # SparseDepthMSE01.py:27-56
            if mask_sd is not None:
                aug = aug_key(name)
                section = model[aug] if aug else model
                key = 'depth_fine' if 'fine_mlp' in section else (f'{aug}_depth_coarse' if aug else 'depth_coarse')
                value = value + masked_mse(output_dict[key], input_dict['sparse_depth_values'][:, 0], mask_sd)
        elif name in ('PointsAugmentationDepthLoss02', 'ViewsAugmentationDepthLoss02'):   # :33-77
            aug = 'points_augmentation' if name.startswith('Points') else 'views_augmentation'
            for level in ('coarse', 'fine'):
                if f'{level}_mlp' in model and f'{level}_mlp' in model[aug]:
                    value = value + consistency_loss(
                        output_dict[f'depth_{level}'], output_dict[f'{aug}_depth_{level}'], mask_nerf,
                        input_dict['rays_o'], input_dict['rays_d'], input_dict['pixel_id'], common['poses'],
                        common['images'], common['intrinsics'], common['resolution'], loss_configs['patch_size'],
                        loss_configs['rmse_threshold'])[0]
        elif name == 'CoarseFineConsistencyLoss02':                           # CoarseFineConsistencyLoss02.py:32-92
            if 'coarse_mlp' in model and 'fine_mlp' in model:
                value = value + consistency_loss(
                    output_dict['depth_coarse'], output_dict['depth_fine'], mask_nerf, input_dict['rays_o'],
                    input_dict['rays_d'], input_dict['pixel_id'], common['poses'], common['images'],
                    common['intrinsics'], common['resolution'], loss_configs['patch_size'],
                    loss_configs['rmse_threshold'])[0]
                if 'sparse_depth' in configs['data_loader'] and mask_sd is not None:      # compute_loss_sd :174-189
                    value = value + masked_mse(output_dict['depth_coarse'], output_dict['depth_fine'].detach(), mask_sd)
        else:
            raise KeyError(f'loss {name} is not part of the restated set')
        values[name] = value
        total = total + loss_weight(loss_configs, input_dict['iter_num']) * value
    values['TotalLoss'] = total
    return values

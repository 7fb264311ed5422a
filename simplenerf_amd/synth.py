"""Deterministic synthetic workloads for the ray-marching path.

Everything here is build-owned: experiment-config dictionaries in the shape the
reference's harness passes to ``get_model`` (key names as read by
``src/models/SimpleNeRF01.py:16-41,567-584``; values as shipped in
``src/NerfLlffTrainerTester01.py:270-350``), PyTorch-Linear-like random
weights drawn from the frozen legacy ``numpy.random.RandomState`` stream, and
seeded ray batches.  No datasets or checkpoints ship with the reference, so
tests, ``bench.py`` and the golden-vector generator all draw from here; a
fixture therefore only needs to store seeds and expected outputs.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Dict, Optional

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
CAMERAS_JSON = os.path.join(os.path.dirname(_HERE), 'tests', 'golden', 'cameras.json')


# --------------------------------------------------------------------------------------
# experiment configs
# --------------------------------------------------------------------------------------
def mlp_config(num_samples: Optional[int] = None, depth: int = 8, width: int = 256, views_width: int = 128,
               use_view_dirs: bool = True, view_dependent_rgb: bool = True,
               sigma_pe_degree: Optional[int] = None, predict_visibility: bool = False, views_depth: int = 1) -> dict:
    cfg = {
        'points_net_depth': depth,
        'views_net_depth': views_depth,
        'points_net_width': width,
        'views_net_width': views_width,
        'points_positional_encoding_degree': 10,
        'use_view_dirs': use_view_dirs,
        'view_dependent_rgb': view_dependent_rgb,
        'predict_visibility': bool(predict_visibility),
    }
    if use_view_dirs:
        cfg['views_positional_encoding_degree'] = 4
    if sigma_pe_degree is not None:
        cfg['points_sigma_positional_encoding_degree'] = sigma_pe_degree
    if num_samples is not None:
        cfg['num_samples'] = num_samples
    return cfg


def make_configs(kind: str, model_name: str = 'SimpleNeRFHip01') -> dict:
    """Experiment config for one of the BASELINE.json workloads.

    kind:
      'config1'   1024 random rays, 64 coarse samples, 4x128 MLP, non-NDC, no fine MLP
      'config2'   LLFF fern, 64+128, 8x256 coarse+fine, NDC
      'config4'   the same renderer for the RealEstate-10K full-frame render (camera('re10k'))
      'headline'  128+128, 8x256 coarse+fine, NDC (the metric configuration)
      'config3'   config2 + points-augmentation + views-augmentation coarse MLPs
      'headline_world'  headline but non-NDC (world-space rays)
      'config3f'  config3 with fine-level augmentation MLPs as well
    """
    model = {
        'name': model_name,
        'chunk': 4 * 1024,
        'lindisp': False,
        'netchunk': 16 * 1024,
        'perturb': True,
        'raw_noise_std': 1.0,
        'white_bkgd': False,
    }
    ndc = True
    if kind == 'config1':
        ndc = False
        model['coarse_mlp'] = mlp_config(64, depth=4, width=128, views_width=64)
    elif kind in ('config2', 'config4'):      # config 4 = the config-2 renderer on the RealEstate-10K camera
        model['coarse_mlp'] = mlp_config(64)
        model['fine_mlp'] = mlp_config(128)
    elif kind == 'headline':
        model['coarse_mlp'] = mlp_config(128)
        model['fine_mlp'] = mlp_config(128)
    elif kind == 'headline_world':
        ndc = False
        model['coarse_mlp'] = mlp_config(128)
        model['fine_mlp'] = mlp_config(128)
    elif kind == 'config3':
        model['coarse_mlp'] = mlp_config(64)
        model['fine_mlp'] = mlp_config(128)
        model['points_augmentation'] = {'coarse_mlp': mlp_config(sigma_pe_degree=3)}
        model['views_augmentation'] = {'coarse_mlp': mlp_config(use_view_dirs=False, view_dependent_rgb=False)}
    elif kind == 'config3f':      # config3 + FINE augmentation MLPs (supported by the reference, not in its shipped configs)
        model['coarse_mlp'] = mlp_config(64)
        model['fine_mlp'] = mlp_config(128)
        model['points_augmentation'] = {'coarse_mlp': mlp_config(sigma_pe_degree=3), 'fine_mlp': mlp_config(sigma_pe_degree=3)}
        model['views_augmentation'] = {'coarse_mlp': mlp_config(use_view_dirs=False, view_dependent_rgb=False),
                                       'fine_mlp': mlp_config(use_view_dirs=False, view_dependent_rgb=False)}
    else:
        raise KeyError(kind)
    return {
        'data_loader': {'ndc': ndc},
        'model': model,
        'device': [0],
    }


def with_overrides(configs: dict, **model_overrides) -> dict:
    out = copy.deepcopy(configs)
    out['model'].update(model_overrides)
    return out


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def synth_state_dict(shapes: Dict[str, tuple], seed: int, sigma_gain: float = 1.0, sigma_shift: float = 0.0
                     ) -> Dict[str, numpy.ndarray]:
    """uniform(+-1/sqrt(fan_in)) weights and biases, one RandomState stream, keys visited in sorted order.

    ``sigma_gain``/``sigma_shift`` rescale the density head (row 0 of every ``pts_output_linear``) so that a
    randomly initialised field is not almost transparent: with the plain init sigma*delta ~ 1e-3 and every
    ray composites to ~0, which would make an absolute 1e-4 colour tolerance vacuous.
    """
    rng = numpy.random.RandomState(seed)
    out = {}
    fan_in = {}
    for key in sorted(shapes):
        if key.endswith('.weight'):
            fan_in[key[:-len('.weight')]] = shapes[key][1]
    for key in sorted(shapes):
        stem = key.rsplit('.', 1)[0]
        bound = 1.0 / numpy.sqrt(fan_in[stem])
        value = rng.uniform(-bound, bound, size=shapes[key]).astype(numpy.float32)
        if stem.endswith('pts_output_linear') and (sigma_gain != 1.0 or sigma_shift != 0.0):
            if key.endswith('.weight'):
                value[0] *= numpy.float32(sigma_gain)
            else:
                value[0] = value[0] * numpy.float32(sigma_gain) + numpy.float32(sigma_shift)
        out[key] = value
    return out


# --------------------------------------------------------------------------------------
# rays
# --------------------------------------------------------------------------------------
def random_world_rays(num_rays: int, seed: int = 1, near: float = 2.0, far: float = 6.0) -> Dict[str, numpy.ndarray]:
    """BASELINE config 1 inputs: origins 0.1*N(0,1), unit directions looking down -z."""
    rng = numpy.random.RandomState(seed)
    rays_o = (0.1 * rng.standard_normal((num_rays, 3))).astype(numpy.float32)
    d = rng.standard_normal((num_rays, 3)).astype(numpy.float32)
    d[:, 2] = -numpy.abs(d[:, 2]) - numpy.float32(0.5)
    d = d / numpy.linalg.norm(d, axis=1, keepdims=True)
    rays_d = d.astype(numpy.float32)
    view_dirs = (rays_d / numpy.linalg.norm(rays_d, axis=1, keepdims=True)).astype(numpy.float32)
    ones = numpy.ones((num_rays, 1), dtype=numpy.float32)
    return {
        'rays_o': rays_o, 'rays_d': rays_d, 'view_dirs': view_dirs,
        'near': numpy.float32(near) * ones, 'far': numpy.float32(far) * ones,
    }


def load_cameras() -> dict:
    with open(CAMERAS_JSON) as f:
        return json.load(f)


def camera(scene: str = 'fern', pose_index: int = 0, downscale: int = 1, resolution: Optional[tuple] = None) -> dict:
    """Camera metadata for a full-frame render: resolution, intrinsic (3x3), processed pose (4x4), near, far.

    Values originate from the reference's saved run metadata (see tests/golden/cameras.json header);
    ``downscale=2`` halves resolution and intrinsics (BASELINE config 2 names fern at 504x378);
    ``resolution=(h, w)`` re-targets the camera to another frame size: focal lengths scaled by w / w0, principal point
    at the new centre (BASELINE config 4 names the RE10K render at 1008x756; the scene's own frames are 1024x576,
    SURVEY 8d).
    """
    cams = load_cameras()[scene]
    h, w = cams['resolution']
    k = numpy.array(cams['intrinsic'], dtype=numpy.float64)
    if downscale != 1:
        h, w = h // downscale, w // downscale
        k = k.copy()
        k[:2] /= downscale
    if resolution is not None:
        scale = resolution[1] / w
        h, w = int(resolution[0]), int(resolution[1])
        k = k.copy()
        k[0, 0] *= scale
        k[1, 1] *= scale
        k[0, 2], k[1, 2] = w / 2, h / 2
    return {
        'resolution': (h, w),
        'intrinsic': k.astype(numpy.float32),
        'pose': numpy.array(cams['processed_poses'][pose_index], dtype=numpy.float32),
        'near': float(cams['near']), 'far': float(cams['far']),
        'near_ndc': float(cams['near_ndc']), 'far_ndc': float(cams['far_ndc']),
    }


# --------------------------------------------------------------------------------------
# a small multi-view scene for the loss / batch-assembly rows
# --------------------------------------------------------------------------------------
def _rotation(rx: float, ry: float, rz: float) -> numpy.ndarray:
    cx, sx, cy, sy, cz, sz = numpy.cos(rx), numpy.sin(rx), numpy.cos(ry), numpy.sin(ry), numpy.cos(rz), numpy.sin(rz)
    mx = numpy.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    my = numpy.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    mz = numpy.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return mz @ my @ mx


def synth_scene(seed: int = 0, num_views: int = 3, height: int = 48, width: int = 64, plane_depth: float = 4.0) -> dict:
    """A textured plane z = -plane_depth seen by ``num_views`` pinhole cameras (camera-to-world poses in the
    reference's processed convention: x right, y up, looking down -z; src/data_preprocessors/DataPreprocessor01.py
    :351-368).  Images are rendered exactly (ray/plane intersection), so a ray's true depth reprojects onto a matching
    patch in the neighbouring view and a wrong depth does not -- what the patch-consistency losses measure.

    Returns float32 arrays: images (V,h,w,3) in [0,1], poses (V,4,4), intrinsics (V,3,3), resolution (h,w),
    true_depth (V,h,w) = ray parameter t with o + t*d on the plane (d as produced by get_rays, d_z = -1 in camera space).
    """
    rng = numpy.random.RandomState(seed)
    focal = 0.9 * width
    k = numpy.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1]], dtype=numpy.float64)
    poses = numpy.zeros((num_views, 4, 4))
    for v in range(num_views):
        angles = 0.04 * rng.standard_normal(3) if v else numpy.zeros(3)
        shift = numpy.array([0.35, 0.25, 0.05]) * rng.standard_normal(3) if v else numpy.zeros(3)
        poses[v, :3, :3] = _rotation(*angles)
        poses[v, :3, 3] = shift
        poses[v, 3, 3] = 1
    phase = rng.uniform(0, 2 * numpy.pi, size=(3, 4))
    xs, ys = numpy.meshgrid(numpy.arange(width, dtype=numpy.float64), numpy.arange(height, dtype=numpy.float64))
    dirs_cam = numpy.stack([(xs - k[0, 2]) / focal, -(ys - k[1, 2]) / focal, -numpy.ones_like(xs)], -1)
    images = numpy.zeros((num_views, height, width, 3))
    depth = numpy.zeros((num_views, height, width))
    for v in range(num_views):
        d = dirs_cam @ poses[v, :3, :3].T
        o = poses[v, :3, 3]
        t = (-plane_depth - o[2]) / d[..., 2]
        px, py = o[0] + t * d[..., 0], o[1] + t * d[..., 1]
        depth[v] = t
        for c in range(3):
            images[v, ..., c] = 0.5 + 0.2 * numpy.sin(2.3 * px + phase[c, 0]) * numpy.cos(1.9 * py + phase[c, 1]) \
                + 0.15 * numpy.sin(5.1 * px + 3.7 * py + phase[c, 2]) + 0.1 * numpy.cos(9.0 * py - 4.0 * px + phase[c, 3])
    images = numpy.clip(images, 0, 1)
    return {
        'images': images.astype(numpy.float32), 'poses': poses.astype(numpy.float32),
        'intrinsics': numpy.repeat(k[None], num_views, 0).astype(numpy.float32), 'resolution': (height, width),
        'true_depth': depth.astype(numpy.float32),
    }


def loss_configs(iter_weighted: bool = True) -> list:
    """The nine losses every shipped experiment enables (src/NerfLlffTrainerTester01.py:351-430)."""
    late = {'iter_weights': {'0': 0, '10000': 0.1}} if iter_weighted else {'weight': 0.1}
    patch = {'rmse_threshold': 0.1, 'patch_size': [5, 5]}
    return [
        {'name': 'MSE01', 'weight': 1}, {'name': 'SparseDepthMSE01', 'weight': 0.1},
        {'name': 'MSE02', 'weight': 1}, {'name': 'SparseDepthMSE02', 'weight': 0.1},
        {'name': 'MSE03', 'weight': 1}, {'name': 'SparseDepthMSE03', 'weight': 0.1},
        {'name': 'PointsAugmentationDepthLoss02', **late, **patch},
        {'name': 'ViewsAugmentationDepthLoss02', **late, **patch},
        {'name': 'CoarseFineConsistencyLoss02', **late, **patch},
    ]


def loss_batch(scene: dict, num_rays: int, num_sparse: int, seed: int = 0) -> Dict[str, numpy.ndarray]:
    """Seeded stand-ins for one training batch and the model outputs the losses read: ``num_rays`` pixel rays followed
    by ``num_sparse`` sparse-depth rays (the layout of load_cached_next_batch, DataPreprocessor01.py:514-584), depth
    estimates scattered around the true plane depth (some exact ties, some wildly off, some border pixels)."""
    rng = numpy.random.RandomState(seed)
    v, (h, w) = scene['images'].shape[0], scene['resolution']
    n = num_rays + num_sparse
    image_id = rng.randint(0, v, size=n)
    x = rng.randint(0, w, size=n)
    y = rng.randint(0, h, size=n)
    x[:8] = [0, 1, 2, w - 1, w - 2, w - 3, 5, 6][:min(8, n)] if n >= 8 else x[:8]
    y[:8] = [5, 6, 7, 8, 0, 1, h - 1, h - 2][:min(8, n)] if n >= 8 else y[:8]
    k = scene['intrinsics'][0].astype(numpy.float64)
    dirs = numpy.stack([(x - k[0, 2]) / k[0, 0], -(y - k[1, 2]) / k[1, 1], -numpy.ones(n)], -1)
    rays_d = numpy.einsum('nj,nij->ni', dirs, scene['poses'][image_id, :3, :3].astype(numpy.float64))
    rays_o = scene['poses'][image_id, :3, 3]
    true = scene['true_depth'][image_id, y, x]

    def estimate(sigma_small, sigma_large):
        pick = rng.uniform(size=n) < 0.6
        return (true + numpy.where(pick, sigma_small, sigma_large) * rng.standard_normal(n)).astype(numpy.float32)

    depths = {name: estimate(0.03, 0.6) for name in ('depth_coarse', 'depth_fine', 'points_augmentation_depth_coarse',
                                                      'views_augmentation_depth_coarse')}
    ties = rng.uniform(size=n) < 0.1
    depths['points_augmentation_depth_coarse'][ties] = depths['depth_coarse'][ties]
    if n >= 16:
        depths['depth_fine'][8:12] = [0.0, -1.5, 1e6, 3e-3]
        depths['views_augmentation_depth_coarse'][12:16] = [1e-4, -40.0, 2.5e4, 0.0]
    target = scene['images'][image_id, y, x]
    colours = {name: numpy.clip(target + 0.1 * rng.standard_normal((n, 3)), 0, 1).astype(numpy.float32)
               for name in ('rgb_coarse', 'rgb_fine', 'points_augmentation_rgb_coarse', 'views_augmentation_rgb_coarse')}
    sparse = numpy.full((n, 1), -1.0, dtype=numpy.float32)
    sparse[num_rays:, 0] = true[num_rays:] + 0.05 * rng.standard_normal(num_sparse)
    target_rgb = target.astype(numpy.float32).copy()
    target_rgb[num_rays:] = -1.0          # sparse-depth rows keep the loader's -1 fill (:598)
    return {
        'rays_o': rays_o.astype(numpy.float32), 'rays_d': rays_d.astype(numpy.float32),
        'pixel_id': numpy.stack([image_id, x, y], 1).astype(numpy.int32), 'target_rgb': target_rgb,
        'indices_mask_nerf': numpy.arange(n) < num_rays, 'indices_mask_sparse_depth': numpy.arange(n) >= num_rays,
        'sparse_depth_values': sparse, **depths, **colours,
    }


def optim_case(seed: int = 0) -> dict:
    """Seeded parameters and per-step gradients for the optimiser fixtures: a few tensor shapes (odd sizes included)
    with gradients spanning many magnitudes (zeros and denormal-sized second moments included), stepped at the
    trainer iterations ``iters``; ``record`` lists the step counts whose state the fixture keeps."""
    rng = numpy.random.RandomState(seed)
    shapes = [(37,), (64, 63), (5,), (1,), (256, 3)]
    params = [rng.uniform(-0.5, 0.5, size=s).astype(numpy.float32) for s in shapes]
    iters = [0, 1, 2, 1000, 250000, 250001]
    grads = []
    for _ in iters:
        step_grads = []
        for s in shapes:
            g = rng.standard_normal(s) * 10.0 ** rng.uniform(-6, 1, size=s)
            g[rng.uniform(size=s) < 0.05] = 0.0
            step_grads.append(g.astype(numpy.float32))
        grads.append(step_grads)
    grads[0][0][:4] = [1e-22, -3e-20, 0.0, 1e-30]
    return {'params': params, 'grads': grads, 'iters': iters, 'record': (1, 2, 6)}


def training_scene(seed: int = 0, num_views: int = 3, height: int = 756, width: int = 1008, sparse_fraction: float = 2e-3,
                   sparse_points: int = None) -> dict:
    """``synth_scene`` at a training resolution in the form ``BatchAssembler`` takes, plus dense sparse-depth tables
    (true plane depth + noise on a random ``sparse_fraction`` of the pixels, -1 elsewhere).
    ``sparse_points``: exactly that many pixels carry a sparse depth instead (a benchmark wants an epoch that is a whole
    number of batches: with the default ~4 570 points every third 2048-row batch of the epoch is 476 rows short)."""
    scene = synth_scene(seed, num_views, height, width)
    rng = numpy.random.RandomState(seed + 50)
    has = rng.uniform(size=scene['true_depth'].shape) < sparse_fraction
    if sparse_points is not None:
        has = numpy.zeros(scene['true_depth'].size, dtype=bool)
        has[rng.choice(has.size, size=int(sparse_points), replace=False)] = True
        has = has.reshape(scene['true_depth'].shape)
    depth = numpy.where(has, scene['true_depth'] + 0.05 * rng.standard_normal(has.shape), -1.0).astype(numpy.float32)
    error = numpy.where(has, rng.uniform(0.1, 2.0, size=has.shape), -1.0).astype(numpy.float32)
    return {**scene, 'near': 1.0, 'far': 6.0, 'near_ndc': 0.0, 'far_ndc': 1.0, 'frame_nums': list(range(num_views)),
            'sparse_depths': depth.reshape(-1), 'sparse_errors': error.reshape(-1),
            'sparse_depths_ndc': numpy.where(has, 1.0 - 1.0 / numpy.maximum(depth, 1e-3), -1.0).astype(numpy.float32).reshape(-1)}


def training_configs(precision: str = 'fp32', num_rays: int = 2048, num_sparse: int = 2048, seed: int = 0) -> dict:
    """The shipped LLFF experiment (src/NerfLlffTrainerTester01.py:245-440) restricted to the keys this build reads:
    config-3 model, 2048 pixel rays + 2048 sparse-depth rays per iteration in sub-batches of 2048, the nine losses,
    Adam(5e-4, 0.9, 0.999) with the NeRF exponential decay."""
    cfg = make_configs('config3')
    cfg['model']['hip_precision'] = precision
    cfg['data_loader'].update(num_rays=num_rays, precrop_fraction=1, precrop_iterations=-1)
    if num_sparse:
        cfg['data_loader']['sparse_depth'] = {'num_rays': num_sparse}
    cfg['losses'] = loss_configs()
    cfg['optimizer'] = {'lr_decayer_name': 'NeRFLearningRateDecayer01', 'lr_initial': 5e-4, 'lr_decay': 250,
                        'beta1': 0.9, 'beta2': 0.999}
    cfg.update(sub_batch_size=2048, num_iterations=100000, seed=seed)
    return cfg


def abi_param_list(params: dict, prefix: str = ''):
    """Parameter tensors of one MLP in the C ABI's order (include/simplenerf_hip.h, snerf_mlp_pack) out of a state dict."""
    names = []
    i = 0
    while f'{prefix}pts_linears.{i}.weight' in params:
        names += [f'pts_linears.{i}.weight', f'pts_linears.{i}.bias']
        i += 1
    names += ['pts_output_linear.weight', 'pts_output_linear.bias']
    if f'{prefix}feature_linear.weight' in params:
        names += ['feature_linear.weight', 'feature_linear.bias']
        j = 0
        while f'{prefix}views_linears.{j}.weight' in params:
            names += [f'views_linears.{j}.weight', f'views_linears.{j}.bias']
            j += 1
        names += ['views_output_linear.weight', 'views_output_linear.bias']
    return [params[prefix + n] for n in names]

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
               sigma_pe_degree: Optional[int] = None) -> dict:
    cfg = {
        'points_net_depth': depth,
        'views_net_depth': 1,
        'points_net_width': width,
        'views_net_width': views_width,
        'points_positional_encoding_degree': 10,
        'use_view_dirs': use_view_dirs,
        'view_dependent_rgb': view_dependent_rgb,
        'predict_visibility': False,
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
      'headline'  128+128, 8x256 coarse+fine, NDC (the metric configuration)
      'config3'   config2 + points-augmentation + views-augmentation coarse MLPs
      'headline_world'  headline but non-NDC (world-space rays)
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
    elif kind == 'config2':
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


def camera(scene: str = 'fern', pose_index: int = 0, downscale: int = 1) -> dict:
    """Camera metadata for a full-frame render: resolution, intrinsic (3x3), processed pose (4x4), near, far.

    Values originate from the reference's saved run metadata (see tests/golden/cameras.json header);
    ``downscale=2`` halves resolution and intrinsics (BASELINE config 2 names fern at 504x378).
    """
    cams = load_cameras()[scene]
    h, w = cams['resolution']
    k = numpy.array(cams['intrinsic'], dtype=numpy.float64)
    if downscale != 1:
        h, w = h // downscale, w // downscale
        k = k.copy()
        k[:2] /= downscale
    return {
        'resolution': (h, w),
        'intrinsic': k.astype(numpy.float32),
        'pose': numpy.array(cams['processed_poses'][pose_index], dtype=numpy.float32),
        'near': float(cams['near']), 'far': float(cams['far']),
        'near_ndc': float(cams['near_ndc']), 'far_ndc': float(cams['far_ndc']),
    }

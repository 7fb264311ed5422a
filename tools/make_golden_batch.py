#!/usr/bin/env python3
"""G9: golden vectors for the batch-assembly row, made by RUNNING THE REFERENCE's DataPreprocessor.

    PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden_batch.py

A train-mode ``DataPreprocessor`` is built over a synthetic 3-view scene (``synth.synth_scene`` images, seeded
world-to-camera matrices, seeded sparse-depth points) and asked for consecutive batches
(``get_next_batch`` -> ``load_cached_next_batch``, src/data_preprocessors/DataPreprocessor01.py:507-551).  The fixture
stores the processed camera data the assembler needs (poses, intrinsics, near/far, images, dense sparse-depth tables),
the index lists the reference drew (its numpy shuffles cannot be replayed on a device) and every tensor of each batch.
A second preprocessor with ``precrop_fraction`` records which pixel indices are candidates during the pre-crop phase.
"""
import os
import sys
import types

import numpy
import pandas
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference/src')
for name in ('skimage', 'skimage.io', 'skimage.transform'):
    sys.modules.setdefault(name, types.ModuleType(name))

from data_preprocessors.DataPreprocessor01 import DataPreprocessor  # noqa: E402  (the reference)

from simplenerf_amd import synth  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def raw_data(seed=0):
    scene = synth.synth_scene(seed)
    rng = numpy.random.RandomState(seed + 100)
    v, (h, w) = 3, scene['resolution']
    extrinsics = numpy.zeros((v, 4, 4))
    for i in range(v):
        rot = synth._rotation(*(0.05 * rng.standard_normal(3)))
        extrinsics[i, :3, :3] = rot
        extrinsics[i, :3, 3] = [0.4 * rng.standard_normal(), 0.3 * rng.standard_normal(), 0.1 * rng.standard_normal()]
        extrinsics[i, 3, 3] = 1
    sparse = {}
    for frame in (0, 1, 2):
        count = 40
        sparse[frame] = pandas.DataFrame({
            'x': rng.uniform(0, w - 1, count), 'y': rng.uniform(0, h - 1, count),
            'depth': rng.uniform(2.5, 6.0, count), 'reprojection_error': rng.uniform(0.1, 2.0, count)})
    return {
        'frame_nums': numpy.array([0, 1, 2]),
        'nerf_data': {'images': numpy.round(scene['images'] * 255).astype(numpy.uint8), 'extrinsics': extrinsics,
                      'intrinsics': scene['intrinsics'].astype(numpy.float64), 'bounds': numpy.array([2.0, 7.0]),
                      'resolution': (h, w)},
        'sparse_depth_data': sparse,
    }


def configs(**loader):
    return {'data_loader': {'bd_factor': 0.75, 'batching': True, 'ndc': True, 'downsampling_factor': 1, 'num_rays': 96,
                            'recenter_camera_poses': True, 'spherify': False, 'precrop_iterations': 0, **loader},
            'model': {'white_bkgd': False}, 'device': 'cpu'}


def main():
    arrays = {}
    numpy.random.seed(7)
    pp = DataPreprocessor(configs(sparse_depth={'num_rays': 32}), mode='train', raw_data_dict=raw_data())
    data = pp.preprocessed_data_dict
    nerf = data['nerf_data']
    arrays.update(poses=nerf['poses'], intrinsics=nerf['intrinsics'], images=nerf['images'].astype(numpy.float32),
                  near=nerf['near'], far=nerf['far'], near_ndc=nerf['near_ndc'], far_ndc=nerf['far_ndc'],
                  resolution=numpy.array(nerf['resolution']),
                  sparse_depths=data['sparse_depth_data']['depths'].numpy(),
                  sparse_errors=data['sparse_depth_data']['reprojection_errors'].numpy(),
                  sparse_depths_ndc=data['sparse_depth_data']['depths_ndc'].numpy(),
                  sparse_candidates=numpy.sort(data['sparse_depth_data']['indices']))
    for b in range(3):
        batch = pp.get_next_batch(iter_num=b)
        for key, value in batch.items():
            if isinstance(value, torch.Tensor):
                arrays[f'batch{b}_{key}'] = value.numpy()
            elif key != 'common_data':
                arrays[f'batch{b}_{key}'] = numpy.asarray(value)
        if b == 0:   # the shared tensors are the processed camera data above, replicated once per configured device
            for key, value in batch['common_data'].items():
                if isinstance(value, torch.Tensor):
                    assert torch.equal(value[0], torch.as_tensor(arrays[key])), key
                    arrays[f'common_{key}_shape'] = numpy.array(value.shape)
                else:
                    arrays[f'common_{key}'] = numpy.asarray(value)
    # a full image (validation path of the trainer: get_next_batch(iter, image_num), :564-567)
    batch = pp.get_next_batch(iter_num=5, image_num=1)
    for key in ('indices', 'rays_o', 'rays_d_ndc', 'target_rgb', 'pixel_id', 'indices_mask_nerf'):
        arrays[f'image1_{key}'] = batch[key].numpy()
    arrays['image1_has_sparse'] = 'indices_mask_sparse_depth' in batch

    numpy.random.seed(8)
    crop = DataPreprocessor(configs(precrop_fraction=0.5, precrop_iterations=10), mode='train', raw_data_dict=raw_data())
    arrays['precrop_candidates'] = numpy.sort(crop.preprocessed_data_dict['indices'])
    for _ in range(10):
        crop.get_next_batch(iter_num=_)
    crop.get_next_batch(iter_num=10)         # :557-558 regenerates the full index list at precrop_iterations
    arrays['after_precrop_count'] = crop.preprocessed_data_dict['indices'].size

    path = os.path.join(OUT, 'batch_assembly.npz')
    numpy.savez_compressed(path, **arrays)
    print(f'batch_assembly.npz: {os.path.getsize(path) / 1024:.0f} KiB;', sorted(k for k in arrays if k.startswith('batch0_')))
    print({k: arrays[k] for k in ('near', 'far', 'near_ndc', 'far_ndc', 'after_precrop_count')}, arrays['precrop_candidates'].size,
          arrays['sparse_candidates'].size)


if __name__ == '__main__':
    main()

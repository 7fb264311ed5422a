#!/usr/bin/env python3
"""Round-2 golden vectors, made by RUNNING THE REFERENCE in the build container (same rules as tools/make_golden.py:
the reference is imported read-only from /root/reference/src, inputs/weights come from simplenerf_amd/synth.py).

    PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden_r02.py

Fixtures written:
    e2e_config4_<profile>.npz   G6 for BASELINE config 4: the RealEstate-10K camera of
                                runs/training/train0011/00000/ModelConfigs.json (1024x576, f = 493.9, near 1, far 133.3)
                                through SimpleNeRF.forward (eval, NDC, 64+128, 8x256 coarse+fine), 128 rays
    display.npz                 f3: DataPreprocessor.post_process_image / post_process_depth
                                (src/data_preprocessors/DataPreprocessor01.py:1106-1114) on seeded arrays with exact .5
                                ties, negatives, values > 1, +-inf and NaN
    inference_outputs.npz       f3: DataPreprocessor.retrieve_inference_outputs (:897-925) on a seeded network-output
                                dictionary of a small frame: which outputs leave the device and in what form
    lr_schedules.npz            f4: both learning-rate decayers sampled every 37 iterations up to 500 000
    e2e_visibility_<case>.npz   predict_visibility (src/models/SimpleNeRF01.py:317-326, :646-649, :691-714, :479-482):
                                'ndc_eval' = fern NDC rays, 64+128, 8x256 coarse+fine both predicting visibility, eval with
                                sec_views_vis and explicit rays_o2 (two secondary views); 'world_train' = world rays, 4x128
                                coarse-only, training mode (rays_o2 derived from common_data poses / pixel_id / num_frames)
"""
import os
import sys
import warnings

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as base  # noqa: E402  (imports the reference, stubs skimage)
from make_golden import DataPreprocessor, cams_from_disk, save, synth  # noqa: E402


def re10k_batch(pixel_seed, n, cams, pose_index=0):
    mc = {k: cams['re10k'][k] for k in ('resolution', 'intrinsic', 'near', 'far', 'near_ndc', 'far_ndc',
                                        'average_pose', 'translation_scale')}
    pp = base.preprocessor(mc)
    batch = pp.create_test_data(numpy.array(cams['re10k']['raw_poses'][pose_index]), preprocess_pose=True)
    h, w = mc['resolution']
    pix = numpy.sort(numpy.random.RandomState(pixel_seed).choice(h * w, n, replace=False))
    pix[0], pix[-1] = 0, h * w - 1          # the two frame corners: largest |x|, |y| of the NDC warp
    return {k: v[pix].contiguous() for k, v in batch.items()}, pix


def make_config4(cams):
    n = 128
    for profile in ('dense', 'consistent'):
        cfg = synth.make_configs('config2')
        model = base.ref_model(cfg, 107, training=False)
        batch, pix = re10k_batch(53, n, cams)
        overrides = base.calibrate_density(model, batch, train_mode=False)
        if profile == 'consistent':
            overrides = {k: v for k, v in overrides.items() if not k.startswith('ovr_fine_model')}
            overrides.update(base.tie_fine_to_coarse(model))
        with torch.no_grad():
            out = model(batch, retraw=True)
            out_plain = model(batch)
        assert all(torch.equal(out[k], out_plain[k]) for k in out_plain)
        arrays = {'seed': 107, 'pixel_indices': pix, 'eval_keys': numpy.array(sorted(out_plain.keys()))}
        arrays.update(overrides)
        arrays.update({f'in_{k}': v for k, v in batch.items()})
        arrays.update({f'out_{k}': v for k, v in out.items()})
        save(f'e2e_config4_{profile}.npz', **arrays)
        print('   acc_coarse mean %.3f  acc_fine mean %.3f  depth_fine max %.1f' % (
            out['acc_coarse'].mean(), out['acc_fine'].mean(), out['depth_fine'].max()))


def display_inputs():
    rng = numpy.random.RandomState(4)
    n = 4096
    rgb = rng.uniform(-0.2, 1.2, (n, 3)).astype(numpy.float32)
    rgb[:512] = (numpy.arange(512 * 3).reshape(512, 3) % 511 + 0.5).astype(numpy.float32) / 255   # x*255 lands on k + 0.5
    rgb[512:520] = numpy.array([0.0, -0.0, 1.0, 1.0000001, -1e-30, 0.5, 0.0019607844, 0.99803925], dtype=numpy.float32)[:, None]
    rgb[520] = [numpy.inf, -numpy.inf, numpy.nan]
    rgb[521] = [numpy.nan, 0.25, 2.0]
    depth = rng.uniform(-1, 6, n).astype(numpy.float32)
    depth[:6] = [0.0, -0.0, numpy.inf, -numpy.inf, numpy.nan, -1e-38]
    return rgb, depth


def make_display():
    rgb, depth = display_inputs()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')         # NaN -> uint8 cast warns; the value numpy produces is what we record
        image = DataPreprocessor.post_process_image(rgb)
    depth_out = DataPreprocessor.post_process_depth(depth)
    save('display.npz', rgb=rgb, depth=depth, image=image, depth_out=depth_out)
    print('   NaN colour ->', image[520, 2], image[521, 0], ' NaN depth ->', depth_out[4], ' -inf depth ->', depth_out[3])


def make_inference_outputs(cams):
    """retrieve_inference_outputs on seeded per-ray outputs of a 12 x 20 frame, for a coarse+fine NDC config, a
    coarse-only world config, and the fine NDC config again with extra per-sample keys present (they must be dropped)."""
    arrays = {}
    for case, kind, ndc in (('fine_ndc', 'config2', True), ('coarse_world', 'config1', False)):
        h, w = 12, 20
        mc = {k: cams['fern'][k] for k in ('intrinsic', 'near', 'far', 'near_ndc', 'far_ndc', 'average_pose', 'translation_scale')}
        mc['resolution'] = [h, w]
        pp = base.preprocessor(mc, ndc=ndc)
        pp.configs['model'] = synth.make_configs(kind)['model']
        rng = numpy.random.RandomState(17)
        net = {}
        for level in (('coarse', 'fine') if kind == 'config2' else ('coarse',)):
            net[f'rgb_{level}'] = rng.uniform(-0.1, 1.1, (h * w, 3)).astype(numpy.float32)
            for k in ('depth', 'depth_var') + (('depth_ndc', 'depth_var_ndc') if ndc else ()):
                net[f'{k}_{level}'] = rng.uniform(-0.5, 5.0, h * w).astype(numpy.float32)
            net[f'acc_{level}'] = rng.uniform(0, 1, h * w).astype(numpy.float32)
            net[f'alpha_{level}'] = rng.uniform(0, 1, (h * w, 8)).astype(numpy.float32)
        out = pp.retrieve_inference_outputs({k: torch.from_numpy(v) for k, v in net.items()})
        arrays[f'{case}_keys'] = numpy.array(list(out.keys()))
        for k, v in net.items():
            arrays[f'{case}_net_{k}'] = v
        for k, v in out.items():
            arrays[f'{case}_out_{k}'] = v
    save('inference_outputs.npz', **arrays)


def make_lr_schedules():
    """Both reference decayers sampled every 37 iterations over their whole horizon (the round-1 fixture held 7 probes,
    which missed that numpy's exp/sin differ from libm's by 1 ulp on ~4 % of iterations)."""
    sys.path.insert(0, '/root/reference/src')
    from lr_decayers.LearningRateDecayerFactory import get_lr_decayer
    iters = numpy.arange(0, 500001, 37, dtype=numpy.int64)
    nerf = get_lr_decayer({'optimizer': {'lr_decayer_name': 'NeRFLearningRateDecayer01', 'lr_initial': 5e-4, 'lr_decay': 250}})
    mip = get_lr_decayer({'num_iterations': 500000,
                          'optimizer': {'lr_decayer_name': 'MipNeRFLearningRateDecayer01', 'lr_initial': 5e-4,
                                        'lr_final': 5e-6, 'lr_decay_steps': 2500, 'lr_decay_mult': 0.01}})
    save('lr_schedules.npz', iters=iters,
         nerf_lr=numpy.array([nerf.get_updated_learning_rate(int(i)) for i in iters], dtype=numpy.float64),
         mip_lr=numpy.array([mip.get_updated_learning_rate(int(i)) for i in iters], dtype=numpy.float64))


def sharpen_visibility(model, names, gain=60.0):
    """Random-init visibility logits are ~0 +- 0.02 (sigmoid 0.5 +- 0.006): scale the 4th row of the views head so that the
    predicted visibilities spread over (0,1) and a wrong direction or weight row shows.  Returned as ovr_* arrays."""
    out = {}
    for name in names:
        lin = dict(model.named_modules())[f'{name}.views_output_linear']
        with torch.no_grad():
            lin.weight[3] *= gain
            lin.bias[3] *= gain
        out[f'ovr_{name}.views_output_linear.weight'] = lin.weight.detach().clone()
        out[f'ovr_{name}.views_output_linear.bias'] = lin.bias.detach().clone()
    return out


def make_visibility(cams):
    # (a) eval, NDC, coarse + fine, explicit secondary camera centres
    n = 64
    cfg = synth.make_configs('config2')
    cfg['model']['coarse_mlp'] = synth.mlp_config(64, predict_visibility=True)
    cfg['model']['fine_mlp'] = synth.mlp_config(128, predict_visibility=True)
    model = base.ref_model(cfg, 108, training=False)
    batch, pix = base.fern_batch(59, n, cams)
    overrides = base.calibrate_density(model, batch, train_mode=False)
    overrides = {k: v for k, v in overrides.items() if not k.startswith('ovr_fine_model')}
    overrides.update(sharpen_visibility(model, ['coarse_model']))
    overrides.update(base.tie_fine_to_coarse(model))
    rng = numpy.random.RandomState(23)
    centres = numpy.array(cams['fern']['processed_poses'])[1:3, :3, 3].astype(numpy.float32)          # two other spiral poses
    batch['rays_o2'] = torch.from_numpy(numpy.broadcast_to(centres[None], (n, 2, 3)).copy() +
                                        0.01 * rng.standard_normal((n, 2, 3)).astype(numpy.float32))
    with torch.no_grad():
        out = model(batch, retraw=True, sec_views_vis=True)
        plain = model(batch, sec_views_vis=True)
        blind = model(batch, retraw=True)                      # sec_views_vis False: no *visibility2* keys
    assert 'visibility2_fine' in plain and 'raw_visibility2_coarse' in out and 'raw_visibility_fine' in blind
    assert not any('visibility2' in k for k in blind)
    arrays = {'seed': 108, 'pixel_indices': pix, 'eval_keys': numpy.array(sorted(plain.keys())),
              'blind_keys': numpy.array(sorted(blind.keys())), 'key_order': numpy.array(list(out.keys()))}
    arrays.update(overrides)
    arrays.update({f'in_{k}': v for k, v in batch.items()})
    arrays.update({f'out_{k}': v for k, v in out.items()})
    save('e2e_visibility_ndc_eval.npz', **arrays)
    print('   visibility2_fine range %.3f..%.3f  raw_visibility_coarse std %.3f' % (
        out['visibility2_fine'].min(), out['visibility2_fine'].max(), out['raw_visibility_coarse'].std()))

    # (b) training mode, world rays, coarse only, secondary centres derived from the training poses
    n = 96
    cfg = synth.with_overrides(synth.make_configs('config1'), perturb=False, raw_noise_std=0.0)
    cfg['model']['coarse_mlp'] = synth.mlp_config(64, depth=4, width=128, views_width=64, predict_visibility=True)
    model = base.ref_model(cfg, 109, training=True)
    batch = {k: torch.from_numpy(v) for k, v in synth.random_world_rays(n, seed=6).items()}
    poses = numpy.stack([numpy.eye(4, dtype=numpy.float32)] * 3)
    poses[:, :3, 3] = rng.uniform(-0.5, 0.5, (3, 3)).astype(numpy.float32)
    batch['pixel_id'] = torch.from_numpy(numpy.stack([rng.randint(0, 3, n), rng.randint(0, 64, n), rng.randint(0, 48, n)], 1).astype(numpy.int32))
    batch['num_frames'] = 3
    batch['common_data'] = {'poses': torch.from_numpy(poses)[None]}     # leading replica axis, as the trainer's loader adds
    overrides = base.calibrate_density(model, batch, train_mode=True)
    overrides.update(sharpen_visibility(model, ['coarse_model']))
    with torch.no_grad():
        out = model(batch)
    assert 'visibility2_coarse' in out and tuple(out['raw_visibility2_coarse'].shape) == (n, 64, 2, 1)
    arrays = {'seed': 109, 'key_order': numpy.array(list(out.keys())), 'poses': poses, 'num_frames': 3}
    arrays.update(overrides)
    arrays.update({f'in_{k}': v for k, v in batch.items() if isinstance(v, torch.Tensor)})
    arrays.update({f'out_{k}': v for k, v in out.items()})
    save('e2e_visibility_world_train.npz', **arrays)
    print('   visibility2_coarse range %.3f..%.3f  raw_visibility2_coarse std %.3f' % (
        out['visibility2_coarse'].min(), out['visibility2_coarse'].max(), out['raw_visibility2_coarse'].std()))


if __name__ == '__main__':
    only = sys.argv[1:]
    cams = cams_from_disk()
    if not only or 'config4' in only:
        make_config4(cams)
    if not only or 'display' in only:
        make_display()
        make_inference_outputs(cams)
    if not only or 'lr' in only:
        make_lr_schedules()
    if not only or 'visibility' in only:
        make_visibility(cams)

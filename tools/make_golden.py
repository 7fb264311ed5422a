#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this build container.

Usage (build container only -- /root/reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden.py

The reference (``/root/reference/src``) is imported read-only via sys.path; nothing of it is copied.  Inputs and
weights come from the build-owned deterministic generators in ``simplenerf_amd/synth.py``, so each fixture stores
only seeds/small inputs and the reference's outputs.  ``skimage`` (absent here, and not used by the functions we
call) is stubbed with empty modules so that ``DataPreprocessor01`` imports.

Fixtures written (names follow SURVEY.md 8c):
    cameras.json            camera metadata for fern / RE10K-00000 + raw and reference-processed poses
    raygen_<scene>.npz      G1  get_rays / get_view_dirs / get_ndc_rays at 1024 fixed pixels x 3 poses
    zvals.npz               G2  get_z_vals_coarse (eval, lindisp, perturb with seeded draws)
    mlp_<layout>.npz        G3  MLP.forward for the main / points-aug / views-aug layouts (8x256 and 4x128)
    composite.npz           G4  volume_rendering (NDC, world, white background; S in 64/192/256)
    resample.npz            G5  sample_pdf + sort (deterministic and with injected u)
    e2e_<config>.npz        G6  SimpleNeRF.forward end-to-end (eval: config1/config2/headline; train: config3)
    grads_<config>.npz      G7  parameter gradients of a fixed scalar loss through the training-mode forward
"""
import json
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

from models.SimpleNeRF01 import SimpleNeRF, MLP  # noqa: E402  (the reference)
from data_preprocessors.DataPreprocessor01 import DataPreprocessor  # noqa: E402  (the reference)

from simplenerf_amd import synth  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    arrays = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else numpy.asarray(v)) for k, v in arrays.items()}
    numpy.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays')


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed, **kw):
    sd = synth.synth_state_dict(shapes_of(module), seed, **kw)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return module


# ------------------------------------------------------------------------------------------------
# cameras + G1 ray generation
# ------------------------------------------------------------------------------------------------
def preprocessor(model_configs, ndc=True):
    configs = {
        'data_loader': {'bd_factor': 0.75, 'batching': True, 'ndc': ndc, 'downsampling_factor': 1, 'num_rays': 2048,
                        'recenter_camera_poses': True, 'spherify': False},
        'model': {}, 'device': 'cpu',
    }
    return DataPreprocessor(configs, mode='test', model_configs=model_configs)


def make_cameras():
    cams = {'_source': 'reference runs/training/train1011/fern/ModelConfigs.json, runs/training/train0011/00000/'
                       'ModelConfigs.json, data/databases/NeRF_LLFF/data/train_test_sets/set02/video_poses01/fern.csv '
                       '(rows 0, 39, 79); RE10K raw poses are synthetic small-baseline world-to-camera matrices. '
                       'processed_poses = reference DataPreprocessor.preprocess_poses(train_mode=False).'}
    with open(f'{REF}/runs/training/train1011/fern/ModelConfigs.json') as f:
        fern = json.load(f)
    spiral = numpy.loadtxt(f'{REF}/data/databases/NeRF_LLFF/data/train_test_sets/set02/video_poses01/fern.csv',
                           delimiter=',').reshape(-1, 4, 4)
    fern_raw = [spiral[i] for i in (0, 39, 79)]
    with open(f'{REF}/runs/training/train0011/00000/ModelConfigs.json') as f:
        re10k = json.load(f)
    rng = numpy.random.RandomState(11)
    re_raw = []
    for _ in range(3):
        ang = 0.05 * rng.standard_normal(3)
        cx, cy, cz = numpy.cos(ang)
        sx, sy, sz = numpy.sin(ang)
        rx = numpy.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        ry = numpy.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        rz = numpy.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        m = numpy.eye(4)
        m[:3, :3] = rz @ ry @ rx
        m[:3, 3] = 0.2 * rng.standard_normal(3)
        re_raw.append(m)
    for scene, mc, raws in (('fern', fern, fern_raw), ('re10k', re10k, re_raw)):
        pp = preprocessor(mc)
        processed = [pp.preprocess_poses({'poses': numpy.array(r)[None].copy(),
                                          'translation_scale': mc['translation_scale'],
                                          'average_pose': numpy.array(mc['average_pose'])},
                                         train_mode=False)['poses'][0] for r in raws]
        cams[scene] = {
            'resolution': mc['resolution'], 'intrinsic': mc['intrinsic'],
            'near': mc['near'], 'far': mc['far'], 'near_ndc': mc['near_ndc'], 'far_ndc': mc['far_ndc'],
            'average_pose': mc['average_pose'], 'translation_scale': mc['translation_scale'],
            'raw_poses': [numpy.asarray(r).tolist() for r in raws],
            'processed_poses': [p.astype(numpy.float64).tolist() for p in processed],
        }
    with open(os.path.join(OUT, 'cameras.json'), 'w') as f:
        json.dump(cams, f, indent=1)
    print('cameras.json written')
    return cams


def cams_from_disk():
    """cameras.json as make_cameras() wrote it (for generators that add fixtures without re-making the others)."""
    with open(os.path.join(OUT, 'cameras.json')) as f:
        return json.load(f)


def make_raygen(cams):
    for scene in ('fern', 're10k'):
        mc = {k: cams[scene][k] for k in ('resolution', 'intrinsic', 'near', 'far', 'near_ndc', 'far_ndc',
                                          'average_pose', 'translation_scale')}
        pp = preprocessor(mc)
        h, w = mc['resolution']
        pix = numpy.sort(numpy.random.RandomState(7).choice(h * w, 1024, replace=False))
        pix[0], pix[-1] = 0, h * w - 1
        arrays = {'pixel_indices': pix}
        for pi, raw in enumerate(cams[scene]['raw_poses']):
            batch = pp.create_test_data(numpy.array(raw), preprocess_pose=True)
            for k in ('rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc', 'near', 'far', 'near_ndc', 'far_ndc'):
                arrays[f'pose{pi}_{k}'] = batch[k].numpy()[pix]
        save(f'raygen_{scene}.npz', **arrays)


# ------------------------------------------------------------------------------------------------
# helpers to drive reference model pieces
# ------------------------------------------------------------------------------------------------
def ref_model(configs, seed, training=False):
    model = SimpleNeRF(configs, None)
    load_synth(model, seed)
    model.train(training)
    return model


def fern_batch(pixel_seed, n, cams, pose_index=0):
    mc = {k: cams['fern'][k] for k in ('resolution', 'intrinsic', 'near', 'far', 'near_ndc', 'far_ndc',
                                       'average_pose', 'translation_scale')}
    pp = preprocessor(mc)
    batch = pp.create_test_data(numpy.array(cams['fern']['raw_poses'][pose_index]), preprocess_pose=True)
    h, w = mc['resolution']
    pix = numpy.sort(numpy.random.RandomState(pixel_seed).choice(h * w, n, replace=False))
    return {k: v[pix].contiguous() for k, v in batch.items()}, pix


# ------------------------------------------------------------------------------------------------
# G2 coarse depths
# ------------------------------------------------------------------------------------------------
def make_zvals():
    arrays = {}
    rng = numpy.random.RandomState(3)
    n = 64
    near_w = torch.from_numpy((2.0 + rng.uniform(0, 0.5, (n, 1))).astype(numpy.float32))
    far_w = torch.from_numpy((6.0 + rng.uniform(0, 2.0, (n, 1))).astype(numpy.float32))
    arrays['near_world'], arrays['far_world'] = near_w, far_w
    for ndc in (False, True):
        for lindisp in (False, True):
            if ndc and lindisp:
                continue  # 1/near_ndc = 1/0
            for s in (64, 128):
                cfg = synth.with_overrides(synth.make_configs('config1'), lindisp=lindisp)
                cfg['data_loader']['ndc'] = ndc
                cfg['model']['coarse_mlp']['num_samples'] = s
                m = SimpleNeRF(cfg, None).eval()
                batch = {'rays_o': torch.zeros(n, 3), 'near': near_w, 'far': far_w,
                         'near_ndc': torch.zeros(n, 1), 'far_ndc': torch.ones(n, 1)}
                arrays[f'eval_ndc{int(ndc)}_lindisp{int(lindisp)}_s{s}'] = m.get_z_vals_coarse(batch)
                m.train()
                torch.manual_seed(1234)
                arrays[f'train_seed1234_ndc{int(ndc)}_lindisp{int(lindisp)}_s{s}'] = m.get_z_vals_coarse(batch)
    save('zvals.npz', **arrays)


# ------------------------------------------------------------------------------------------------
# G3 MLP layouts
# ------------------------------------------------------------------------------------------------
def make_mlp():
    rng = numpy.random.RandomState(5)
    n = 512
    pts_ndc = rng.uniform(-1.5, 1.5, (n, 3)).astype(numpy.float32)
    pts_world = rng.uniform(-6, 6, (n, 3)).astype(numpy.float32)
    pts = numpy.concatenate([pts_ndc, pts_world], 0)
    vd = rng.standard_normal((2 * n, 3)).astype(numpy.float32)
    vd /= numpy.linalg.norm(vd, axis=1, keepdims=True)
    layouts = {
        'main': dict(),
        'ptsaug': dict(sigma_pe_degree=3),
        'viewsaug': dict(use_view_dirs=False, view_dependent_rgb=False),
    }
    base = synth.make_configs('config2')
    for lname, kw in layouts.items():
        for (depth, width, vwidth) in ((8, 256, 128), (4, 128, 64)):
            mcfg = synth.mlp_config(64, depth=depth, width=width, views_width=vwidth, **kw)
            for mode, seed, gain, shift in (('plain', 21, 1.0, 0.0), ('dense', 22, 400.0, 10.0)):
                mlp = load_synth(MLP(base, mcfg), seed, sigma_gain=gain, sigma_shift=shift).eval()
                with torch.no_grad():
                    out = mlp({'pts': torch.from_numpy(pts), 'view_dirs': torch.from_numpy(vd)})
                arrays = {'pts': pts, 'view_dirs': vd, 'seed': seed, 'sigma_gain': gain, 'sigma_shift': shift,
                          'depth': depth, 'width': width, 'views_width': vwidth}
                for k, v in out.items():
                    arrays[f'out_{k}'] = v
                save(f'mlp_{lname}_{depth}x{width}_{mode}.npz', **arrays)


# ------------------------------------------------------------------------------------------------
# G4 compositing, G5 resampling
# ------------------------------------------------------------------------------------------------
def make_composite(cams):
    arrays = {}
    rng = numpy.random.RandomState(9)
    n = 64
    for case, ndc, white, s in (('ndc_s64', True, False, 64), ('ndc_s192', True, False, 192),
                                ('ndc_s256', True, False, 256), ('world_s64', False, False, 64),
                                ('world_s192', False, False, 192), ('world_white_s64', False, True, 64),
                                ('ndc_white_s128', True, True, 128)):
        cfg = synth.with_overrides(synth.make_configs('config2'), white_bkgd=white)
        cfg['data_loader']['ndc'] = ndc
        m = SimpleNeRF(cfg, None).eval()
        if ndc:
            batch, _ = fern_batch(31, n, cams)
            z = numpy.sort(rng.uniform(0, 1, (n, s)).astype(numpy.float32), axis=1)
            z[: n // 2] = numpy.linspace(0, 1, s, dtype=numpy.float32)  # includes z_S == 1 (the c=1e-3 branch)
        else:
            wr = synth.random_world_rays(n, seed=2)
            wr['rays_d'] = (wr['rays_d'] * rng.uniform(0.5, 2.0, (n, 1))).astype(numpy.float32)
            batch = {k: torch.from_numpy(v) for k, v in wr.items()}
            z = numpy.sort(rng.uniform(2, 6, (n, s)).astype(numpy.float32), axis=1)
        sigma = (rng.gamma(0.5, 20.0, (n, s, 1))).astype(numpy.float32)
        sigma[rng.uniform(size=sigma.shape) < 0.3] = 0.0
        sigma[:4] = 0.0                 # fully empty rays
        sigma[4:8] = 1e4                # opaque at the first sample
        rgb = rng.uniform(0, 1, (n, s, 3)).astype(numpy.float32)
        net = {'rgb': torch.from_numpy(rgb), 'sigma': torch.from_numpy(sigma)}
        zt = torch.from_numpy(z)
        with torch.no_grad():
            if ndc:
                out = m.volume_rendering(net, z_vals_ndc=zt, rays_d_ndc=batch['rays_d_ndc'], rays_o=batch['rays_o'],
                                         rays_d=batch['rays_d'])
            else:
                out = m.volume_rendering(net, z_vals=zt, rays_d=batch['rays_d'])
        arrays[f'{case}_sigma'], arrays[f'{case}_rgb'], arrays[f'{case}_z'] = sigma[..., 0], rgb, z
        for k in ('rays_o', 'rays_d', 'rays_o_ndc', 'rays_d_ndc'):
            if k in batch:
                arrays[f'{case}_{k}'] = batch[k]
        for k, v in out.items():
            arrays[f'{case}_out_{k}'] = v
    save('composite.npz', **arrays)


def make_resample():
    arrays = {}
    rng = numpy.random.RandomState(13)
    n = 128
    for case, s_c, s_f in (('c64_f128', 64, 128), ('c128_f128', 128, 128), ('c64_f64', 64, 64)):
        cfg = synth.make_configs('config2')
        cfg['model']['coarse_mlp']['num_samples'] = s_c
        cfg['model']['fine_mlp']['num_samples'] = s_f
        m = SimpleNeRF(cfg, None).eval()
        z = torch.linspace(0., 1., s_c).expand(n, s_c).contiguous()
        zj = numpy.sort(rng.uniform(2, 6, (n, s_c)).astype(numpy.float32), axis=1)
        z = torch.cat([z[: n // 2], torch.from_numpy(zj[n // 2:])], 0)
        w = rng.gamma(0.3, 0.05, (n, s_c)).astype(numpy.float32)          # diffuse
        w[: n // 4] = 0
        peak = rng.randint(1, s_c - 1, n // 4)
        w[numpy.arange(n // 4), peak] = 0.97                                # opaque / peaky
        w[n // 4: n // 4 + 4] = 0                                           # all-empty rays
        wt = torch.from_numpy(w)
        arrays[f'{case}_z_coarse'], arrays[f'{case}_weights'] = z, w
        arrays[f'{case}_det'] = m.get_z_vals_fine(z, wt)
        m.train()
        torch.manual_seed(77)
        arrays[f'{case}_seed77'] = m.get_z_vals_fine(z, wt)
        torch.manual_seed(77)
        arrays[f'{case}_u_seed77'] = torch.rand((n, s_f))
    save('resample.npz', **arrays)


# ------------------------------------------------------------------------------------------------
# G6 end to end
# ------------------------------------------------------------------------------------------------
def sigma_heads(model):
    return {name: mod for name, mod in model.named_modules() if name.endswith('pts_output_linear')}


def calibrate_density(model, batch, train_mode):
    """Rescale every density head so the random field has empty space AND opaque regions.

    With the plain Linear init sigma*delta ~ 1e-3: every ray composites to ~0 and an absolute 1e-4 colour
    tolerance would be vacuous.  For each MLP the pre-activation density ``raw`` seen on this batch is
    re-centred on its median and scaled so that its 90th percentile maps to sigma = 30:
    ``row0 <- g*row0, bias0 <- g*(bias0 - median)``.  Coarse-level heads first, then the fine head (whose sample
    positions depend on the calibrated coarse weights).  The resulting head tensors are stored in the fixture
    (``ovr_*``) because they are not a function of the seed alone.
    """
    captured = {}
    hooks = [mod.register_forward_hook(lambda m, i, o, name=name: captured.setdefault(name, []).append(o[..., 0].detach()))
             for name, mod in sigma_heads(model).items()]
    overrides = {}

    def run():
        captured.clear()
        torch.manual_seed(0)
        with torch.no_grad():
            model(batch)

    def apply(names):
        for name in names:
            raw = torch.cat([c.reshape(-1) for c in captured[name]])
            med = raw.median()
            spread = torch.quantile(raw[:200000], 0.9) - med
            g = 30.0 / spread
            lin = dict(model.named_modules())[name]
            with torch.no_grad():
                lin.weight[0] *= g
                lin.bias[0] = g * (lin.bias[0] - med)
            overrides[f'ovr_{name}.weight'] = lin.weight.detach().clone()
            overrides[f'ovr_{name}.bias'] = lin.bias.detach().clone()

    was_training = model.training
    model.train(train_mode)
    run()
    apply([n for n in captured if not n.startswith('fine_model')])
    if any(n.startswith('fine_model') for n in sigma_heads(model)):
        run()
        apply([n for n in captured if n.startswith('fine_model')])
    for h in hooks:
        h.remove()
    model.train(was_training)
    return overrides


def tie_fine_to_coarse(model):
    """'consistent' profile: the fine MLP gets the coarse MLP's (calibrated) weights, so both passes see the same
    geometry -- as in a trained model, and unlike two independent random fields.  This matters because sample_pdf
    has a rounding-dependent discontinuity (denom < 1e-5 -> 1, :357) exactly on EMPTY coarse bins: which of those
    bins collapse to their left edge depends on the last bit of the running cumsum, so no other implementation (nor
    the reference on another BLAS) reproduces it sample-for-sample.  With consistent geometry the affected samples
    carry ~zero weight in the fine pass and the rendered outputs are well defined."""
    model.fine_model.load_state_dict(model.coarse_model.state_dict())
    return {'fine_equals_coarse': 1}


def make_e2e(cams):
    # eval-mode: config1 (world rays), config2 and headline (fern NDC rays)
    for kind, n, seed in (('config1', 512, 101), ('config2', 160, 102), ('headline', 128, 103),
                          ('headline_world', 64, 104)):
        for profile in ('plain', 'dense', 'consistent'):
            cfg = synth.make_configs(kind)
            if profile == 'consistent' and 'fine_mlp' not in cfg['model']:
                continue
            model = ref_model(cfg, seed, training=False)
            if cfg['data_loader']['ndc']:
                batch, pix = fern_batch(41, n, cams)
            else:
                batch = {k: torch.from_numpy(v) for k, v in synth.random_world_rays(n, seed=1).items()}
                pix = numpy.arange(n)
            overrides = calibrate_density(model, batch, train_mode=False) if profile != 'plain' else {}
            if profile == 'consistent':
                overrides = {k: v for k, v in overrides.items() if not k.startswith('ovr_fine_model')}
                overrides.update(tie_fine_to_coarse(model))
            with torch.no_grad():
                out = model(batch, retraw=True)
                out_plain = model(batch)
            assert all(torch.equal(out[k], out_plain[k]) for k in out_plain)
            arrays = {'seed': seed, 'pixel_indices': pix, 'eval_keys': numpy.array(sorted(out_plain.keys()))}
            arrays.update(overrides)
            arrays.update({f'in_{k}': v for k, v in batch.items()})
            arrays.update({f'out_{k}': v for k, v in out.items()})
            save(f'e2e_{kind}_{profile}.npz', **arrays)
            print('   acc_coarse mean %.3f  max weight mean %.3f' % (out['acc_coarse'].mean(), out['weights_coarse'].max(1)[0].mean()))

    # train-mode: config3 (augmented MLPs run only when training)
    n = 96
    for variant, perturb, noise_std, profile in (('det', False, 0.0, 'dense'), ('rand', True, 1.0, 'dense'),
                                                 ('rand', True, 1.0, 'plain'), ('det', False, 0.0, 'consistent'),
                                                 ('rand', True, 1.0, 'consistent')):
        cfg = synth.with_overrides(synth.make_configs('config3'), perturb=perturb, raw_noise_std=noise_std)
        model = ref_model(cfg, 105, training=True)
        batch, pix = fern_batch(43, n, cams)
        overrides = calibrate_density(model, batch, train_mode=True) if profile != 'plain' else {}
        if profile == 'consistent':
            overrides = {k: v for k, v in overrides.items() if not k.startswith('ovr_fine_model')}
            overrides.update(tie_fine_to_coarse(model))
        torch.manual_seed(2024)
        with torch.no_grad():
            out = model(batch)
        arrays = {'seed': 105, 'torch_seed': 2024, 'pixel_indices': pix, 'perturb': perturb, 'raw_noise_std': noise_std}
        arrays.update(overrides)
        arrays.update({f'in_{k}': v for k, v in batch.items()})
        arrays.update({f'out_{k}': v for k, v in out.items()})
        save(f'e2e_config3_train_{variant}_{profile}.npz', **arrays)


GRAD_SAMPLE_STRIDE = 61


def grad_loss(out):
    """Fixed scalar loss over exactly the outputs the shipped losses read: mean squares of every
    rgb_* / depth_* output (augmentation-prefixed ones included), depths scaled to O(1)."""
    loss = 0.
    for k in sorted(out):
        base = k.replace('points_augmentation_', '').replace('views_augmentation_', '')
        if base in ('rgb_coarse', 'rgb_fine'):
            loss = loss + (out[k] ** 2).mean()
        elif base in ('depth_coarse', 'depth_fine'):
            loss = loss + 0.01 * (out[k] ** 2).mean()
    return loss


def make_grads(cams):
    for kind, n, profile in (('config3', 64, 'consistent'), ('config2', 64, 'consistent'), ('headline_world', 48, 'dense'),
                             ('config1', 96, 'dense')):
        cfg = synth.with_overrides(synth.make_configs(kind), perturb=False, raw_noise_std=0.0)
        model = ref_model(cfg, 106, training=True)
        if cfg['data_loader']['ndc']:
            batch, pix = fern_batch(47, n, cams)
        else:
            batch = {k: torch.from_numpy(v) for k, v in synth.random_world_rays(n, seed=5).items()}
        overrides = calibrate_density(model, batch, train_mode=True)
        if profile == 'consistent':
            overrides = {k: v for k, v in overrides.items() if not k.startswith('ovr_fine_model')}
            overrides.update(tie_fine_to_coarse(model))
        model.zero_grad()
        out = model(batch)
        loss = grad_loss(out)
        loss.backward()
        arrays = {'seed': 106, 'loss': loss.detach()}
        arrays.update(overrides)
        arrays.update({f'in_{k}': v for k, v in batch.items()})
        for name, p in model.named_parameters():  # fixture stays small: norm + a strided sample of every gradient
            flat = p.grad.reshape(-1)
            arrays[f'gradnorm_{name}'] = flat.double().norm()
            arrays[f'gradsample_{name}'] = flat[::GRAD_SAMPLE_STRIDE].clone()
        for k in ('rgb_coarse', 'depth_coarse'):
            arrays[f'out_{k}'] = out[k].detach()
        save(f'grads_{kind}_{profile}.npz', **arrays)
        print('   loss %.6f  max|grad| %.3e' % (float(loss), max(float(p.grad.abs().max()) for p in model.parameters())))


if __name__ == '__main__':
    cams = make_cameras()
    make_raygen(cams)
    make_zvals()
    make_mlp()
    make_composite(cams)
    make_resample()
    make_e2e(cams)
    make_grads(cams)

"""Pin the oracle (oracle/*.py) against golden vectors produced by the reference itself
(tools/make_golden.py, run in the build container).  CPU only."""
import numpy
import pytest
import torch

from oracle import nerf_oracle as oracle
from oracle import raygen_oracle
from simplenerf_amd import synth
from tests import util

torch.set_num_threads(max(1, min(8, torch.get_num_threads())))


# ---------------------------------------------------------------- G1 ray generation
@pytest.mark.parametrize('scene', ['fern', 're10k'])
def test_raygen_matches_reference(scene):
    g = util.load(f'raygen_{scene}.npz')
    cams = synth.load_cameras()[scene]
    pix = g['pixel_indices']
    for pi in range(3):
        pose = raygen_oracle.process_pose(numpy.array(cams['raw_poses'][pi]), numpy.array(cams['average_pose']),
                                          cams['translation_scale'])
        assert util.linf(pose, numpy.array(cams['processed_poses'][pi], dtype=numpy.float32)) == 0.0
        batch = raygen_oracle.full_frame_batch(cams['resolution'], numpy.array(cams['intrinsic']), pose, cams['near'],
                                               cams['far'], True, cams['near_ndc'], cams['far_ndc'])
        for k in ('rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc', 'near', 'far', 'near_ndc', 'far_ndc'):
            assert batch[k].dtype == numpy.float32
            assert util.linf(batch[k][pix], g[f'pose{pi}_{k}']) == 0.0, k


# ---------------------------------------------------------------- G2 coarse depths
def test_coarse_depths_match_reference():
    g = util.load('zvals.npz')
    near_w, far_w = torch.from_numpy(g['near_world']), torch.from_numpy(g['far_world'])
    n = near_w.shape[0]
    for key, ref in g.items():
        if not (key.startswith('eval_') or key.startswith('train_')):
            continue
        parts = key.split('_')
        ndc = parts[-3] == 'ndc1'
        lindisp = parts[-2] == 'lindisp1'
        s = int(parts[-1][1:])
        near, far = (torch.zeros(n, 1), torch.ones(n, 1)) if ndc else (near_w, far_w)
        t_rand = None
        if key.startswith('train_'):
            gen = torch.Generator().manual_seed(1234)
            t_rand = torch.rand((n, s), generator=gen)
        z = oracle.coarse_depths(near, far, s, lindisp, t_rand)
        assert util.linf(z, ref) == 0.0, key


# ---------------------------------------------------------------- G3 MLP layouts
@pytest.mark.parametrize('layout', ['main', 'ptsaug', 'viewsaug'])
@pytest.mark.parametrize('size', ['8x256', '4x128'])
@pytest.mark.parametrize('mode', ['plain', 'dense'])
def test_mlp_matches_reference(layout, size, mode):
    g = util.load(f'mlp_{layout}_{size}_{mode}.npz')
    kw = {'main': {}, 'ptsaug': dict(sigma_pe_degree=3),
          'viewsaug': dict(use_view_dirs=False, view_dependent_rgb=False)}[layout]
    cfg = synth.mlp_config(64, depth=int(g['depth']), width=int(g['width']), views_width=int(g['views_width']), **kw)
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), int(g['seed']), float(g['sigma_gain']), float(g['sigma_shift']))
    params = {k: torch.from_numpy(v) for k, v in sd.items()}
    out = oracle.mlp_forward(params, '', cfg, torch.from_numpy(g['pts']), torch.from_numpy(g['view_dirs']))
    assert util.rel_linf(out['sigma'], g['out_sigma']) < 2e-6
    assert util.linf(out['rgb'], g['out_rgb']) < 1e-6
    for k in ('rgb_view_dependent', 'rgb_view_independent'):
        assert (k in out) == (f'out_{k}' in g)
        if k in out:
            assert util.linf(out[k], g[f'out_{k}']) < 1e-6


# ---------------------------------------------------------------- G4 compositing
CASES_G4 = [('ndc_s64', True, False), ('ndc_s192', True, False), ('ndc_s256', True, False), ('world_s64', False, False),
            ('world_s192', False, False), ('world_white_s64', False, True), ('ndc_white_s128', True, True)]


@pytest.mark.parametrize('case,ndc,white', CASES_G4)
def test_composite_matches_reference(case, ndc, white):
    g = util.load('composite.npz')
    t = lambda k: torch.from_numpy(g[f'{case}_{k}'])
    if ndc:
        out = oracle.composite(t('sigma'), t('rgb'), t('z'), t('rays_d_ndc'), True, white, t('rays_o'), t('rays_d'))
    else:
        out = oracle.composite(t('sigma'), t('rgb'), t('z'), t('rays_d'), False, white)
    ref_keys = sorted(k[len(case) + 5:] for k in g if k.startswith(f'{case}_out_'))
    assert sorted(out.keys()) == ref_keys
    for k in ref_keys:
        assert util.rel_linf(out[k], g[f'{case}_out_{k}']) < 1e-6, k


# ---------------------------------------------------------------- G5 hierarchical resampling
@pytest.mark.parametrize('case,s_f', [('c64_f128', 128), ('c128_f128', 128), ('c64_f64', 64)])
def test_resample_matches_reference(case, s_f):
    g = util.load('resample.npz')
    z, w = torch.from_numpy(g[f'{case}_z_coarse']), torch.from_numpy(g[f'{case}_weights'])
    assert util.linf(oracle.resample_depths(z, w, s_f), g[f'{case}_det']) == 0.0
    u = torch.from_numpy(g[f'{case}_u_seed77'])
    assert util.linf(oracle.resample_depths(z, w, s_f, u), g[f'{case}_seed77']) == 0.0
    # the replayed draw is the reference's draw
    gen = torch.Generator().manual_seed(77)
    assert torch.equal(torch.rand(u.shape, generator=gen), u)


# ---------------------------------------------------------------- G6 end to end
@pytest.mark.parametrize('kind', ['config1', 'config2', 'headline', 'headline_world'])
@pytest.mark.parametrize('profile', ['plain', 'dense', 'consistent'])
def test_render_eval_matches_reference(kind, profile):
    if kind == 'config1' and profile == 'consistent':
        pytest.skip('config1 has no fine pass')
    g = util.load(f'e2e_{kind}_{profile}.npz')
    cfg = synth.make_configs(kind)
    params = util.golden_params(cfg, g)
    batch = util.golden_batch(g)
    out = oracle.render(params, cfg, batch, training=False, retraw=True)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    assert sorted(out.keys()) == sorted(ref.keys())
    for k, v in ref.items():
        assert util.rel_linf(out[k], v) < 2e-5, k
    plain = oracle.render(params, cfg, batch, training=False, retraw=False)
    assert sorted(plain.keys()) == sorted(g['eval_keys'].tolist())


@pytest.mark.parametrize('variant,profile', [('det', 'dense'), ('rand', 'dense'), ('rand', 'plain'),
                                             ('det', 'consistent'), ('rand', 'consistent')])
def test_render_train_matches_reference(variant, profile):
    g = util.load(f'e2e_config3_train_{variant}_{profile}.npz')
    cfg = synth.with_overrides(synth.make_configs('config3'), perturb=bool(g['perturb']),
                               raw_noise_std=float(g['raw_noise_std']))
    params = util.golden_params(cfg, g)
    batch = util.golden_batch(g)
    draws = oracle.replay_reference_draws(cfg, batch['rays_o'].shape[0], int(g['torch_seed']))
    out = oracle.render(params, cfg, batch, training=True, rand_per_chunk=draws)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    assert sorted(out.keys()) == sorted(ref.keys())
    for k, v in ref.items():
        assert util.rel_linf(out[k], v) < 2e-5, k


def test_render_train_with_fine_augmentation_mlps_matches_reference():
    """config3f: fine-level points-/views-augmentation MLPs on the fine samples (reference :234-263), deterministic."""
    g = util.load('e2e_config3f_train_det_consistent.npz')
    cfg = synth.with_overrides(synth.make_configs('config3f'), perturb=False, raw_noise_std=0.0)
    out = oracle.render(util.golden_params(cfg, g), cfg, util.golden_batch(g), training=True)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    assert sorted(out.keys()) == sorted(ref.keys())
    assert 'points_augmentation_rgb_fine' in ref and 'views_augmentation_depth_fine' in ref
    for k, v in ref.items():
        assert util.rel_linf(out[k], v) < 2e-5, k


# ---------------------------------------------------------------- G7 parameter gradients
def oracle_grads(cfg, g):
    """Autograd through the oracle's training-mode forward with the fixed scalar loss of tools/make_golden.py."""
    params = {k: v.clone().requires_grad_(True) for k, v in util.golden_params(cfg, g).items()}
    out = oracle.render(params, cfg, util.golden_batch(g), training=True)
    loss = util.grad_loss(out)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in params.items()}, out


@pytest.mark.parametrize('kind,profile', [('config3', 'consistent'), ('config2', 'consistent'), ('headline_world', 'dense'),
                                          ('config1', 'dense')])
def test_gradients_match_reference(kind, profile):
    g = util.load(f'grads_{kind}_{profile}.npz')
    cfg = synth.with_overrides(synth.make_configs(kind), perturb=False, raw_noise_std=0.0)
    loss, grads, _ = oracle_grads(cfg, g)
    assert abs(float(loss) - float(g['loss'])) < 1e-5 * max(1.0, abs(float(g['loss'])))
    checked = 0
    for name, grad in grads.items():
        assert grad is not None, name
        ref_norm = float(g[f'gradnorm_{name}'])
        sample = grad.reshape(-1)[::util.GRAD_SAMPLE_STRIDE]
        scale = max(ref_norm / max(1.0, grad.numel() ** 0.5), 1e-12)  # rms magnitude of this tensor's gradient
        assert abs(float(grad.double().norm()) - ref_norm) <= 2e-3 * ref_norm + 1e-12, name
        assert util.linf(sample, g[f'gradsample_{name}']) <= 2e-3 * max(scale, float(numpy.abs(g[f'gradsample_{name}']).max())), name
        checked += 1
    assert checked == len(util.model_param_shapes(cfg))


# ---------------------------------------------------------------- G8 losses
from oracle import loss_oracle  # noqa: E402

LOSS_CASES = ['full', 'early', 'nosd', 'empty']


@pytest.mark.parametrize('case', LOSS_CASES)
def test_loss_oracle_matches_reference(case):
    g = util.load(f'losses_{case}.npz')
    configs, input_dict, output_dict = util.loss_case(g)
    values = loss_oracle.compute_losses(configs, input_dict, output_dict)
    assert float(torch.as_tensor(values['TotalLoss']).detach()) == pytest.approx(float(g['TotalLoss']), rel=2e-6, abs=1e-9)
    for cfg in configs['losses']:
        assert float(values[cfg['name']]) == pytest.approx(float(g[f"value_{cfg['name']}"]), rel=2e-6, abs=1e-9), cfg['name']
    if isinstance(values['TotalLoss'], torch.Tensor) and values['TotalLoss'].requires_grad:
        values['TotalLoss'].backward()
    for k in util.LOSS_OUTPUT_KEYS:
        grad = output_dict[k].grad
        grad = numpy.zeros_like(g[f'grad_{k}']) if grad is None else grad.numpy()
        scale = max(float(numpy.abs(g[f'grad_{k}']).max()), 1e-12)
        assert util.linf(grad, g[f'grad_{k}']) <= 2e-6 * scale, k


def test_reprojection_and_masks_match_reference():
    """Un-rounded reprojected positions are bit-identical, the nearest-view choice is identical, and the per-ray loss
    maps (which encode the patch decision masks) match on every ray."""
    g = util.load('losses_nosd.npz')
    configs, inp, out = util.loss_case(g)
    common = inp['common_data']
    closest = loss_oracle.closest_other_view(common['poses'], inp['pixel_id'][:, 0].long())
    assert numpy.array_equal(closest.numpy(), g['closest_view'])
    pts = inp['rays_o'] + inp['rays_d'] * out['depth_coarse'].detach()[:, None]
    pos = loss_oracle.reproject(pts, common['poses'][closest], common['intrinsics'][0])
    assert util.linf(pos.numpy(), g['reprojected_depth_coarse']) == 0.0
    pairs = {'PointsAugmentationDepthLoss02': ('depth_coarse', 'points_augmentation_depth_coarse', 'coarse_main', 'coarse_augmented'),
             'ViewsAugmentationDepthLoss02': ('depth_coarse', 'views_augmentation_depth_coarse', 'coarse_main', 'coarse_augmented'),
             'CoarseFineConsistencyLoss02': ('depth_coarse', 'depth_fine', 'coarse', 'fine')}
    for name, (k1, k2, m1, m2) in pairs.items():
        _, m = loss_oracle.consistency_loss(out[k1], out[k2], inp['indices_mask_nerf'], inp['rays_o'], inp['rays_d'],
                                            inp['pixel_id'], common['poses'], common['images'], common['intrinsics'],
                                            common['resolution'], [5, 5], 0.1)
        assert util.linf(m['map1'].detach().numpy(), g[f'map_{name}_{name}_{m1}']) <= 1e-6, name
        assert util.linf(m['map2'].detach().numpy(), g[f'map_{name}_{name}_{m2}']) <= 1e-6, name
        frac = float(m['mask1'].float().mean()), float(m['mask2'].float().mean())
        assert 0.05 < frac[0] < 0.95 and 0.05 < frac[1] < 0.95, (name, frac)   # the fixture exercises both outcomes


# ---------------------------------------------------------------- G10 optimiser + learning-rate schedules
from oracle import optim_oracle  # noqa: E402


def test_adam_oracle_is_bit_identical_to_torch_cpu_adam():
    g = util.load('optim_adam.npz')
    case = synth.optim_case(int(g['seed']))
    params = [p.copy() for p in case['params']]
    m = [numpy.zeros_like(p) for p in params]
    v = [numpy.zeros_like(p) for p in params]
    for step, iter_num in enumerate(case['iters'], 1):
        lr = optim_oracle.nerf_learning_rate(5e-4, 250, iter_num)
        assert lr == float(g['lrs'][step - 1])
        optim_oracle.adam_step(params, case['grads'][step - 1], m, v, step, lr)
        if step in case['record']:
            for i in range(len(params)):
                assert numpy.array_equal(params[i], g[f'step{step}_param{i}']), (step, i)
                assert numpy.array_equal(m[i], g[f'step{step}_exp_avg{i}']), (step, i)
                assert numpy.array_equal(v[i], g[f'step{step}_exp_avg_sq{i}']), (step, i)


def test_learning_rate_schedules_match_reference():
    g = util.load('optim_adam.npz')
    for it, nerf, mip in zip(g['probe_iters'], g['nerf_lr'], g['mip_lr']):
        assert optim_oracle.nerf_learning_rate(5e-4, 250, int(it)) == float(nerf)
        assert optim_oracle.mipnerf_learning_rate(5e-4, 5e-6, 500000, 2500, 0.01, int(it)) == float(mip)
    dense = util.load('lr_schedules.npz')           # every 37th iteration of the whole horizon: the same doubles
    for it, nerf, mip in zip(dense['iters'], dense['nerf_lr'], dense['mip_lr']):
        assert optim_oracle.nerf_learning_rate(5e-4, 250, int(it)) == float(nerf), it
        assert optim_oracle.mipnerf_learning_rate(5e-4, 5e-6, 500000, 2500, 0.01, int(it)) == float(mip), it


# ---------------------------------------------------------------- G9 batch assembly, index stream, draws
from oracle import batch_oracle  # noqa: E402

BATCH_KEYS = ('rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc', 'pixel_id', 'target_rgb', 'near', 'far',
              'near_ndc', 'far_ndc', 'sparse_depth_values', 'sparse_depth_errors', 'sparse_depth_values_ndc',
              'indices_mask_nerf', 'indices_mask_sparse_depth')


def test_batch_oracle_matches_reference_batches_bit_for_bit():
    g = util.load('batch_assembly.npz')
    res = tuple(int(v) for v in g['resolution'])
    cache = batch_oracle.build_ray_cache(g['poses'], g['intrinsics'], res, float(g['near']), True)
    for b in range(3):
        idx = g[f'batch{b}_indices']
        out = batch_oracle.assemble_batch(idx, 96, cache, g['images'], float(g['near']), float(g['far']), True,
                                          float(g['near_ndc']), float(g['far_ndc']), g['sparse_depths'], g['sparse_errors'],
                                          g['sparse_depths_ndc'])
        for k in BATCH_KEYS:
            assert out[k].dtype == g[f'batch{b}_{k}'].dtype, k
            assert numpy.array_equal(out[k], g[f'batch{b}_{k}']), (b, k)
    out = batch_oracle.assemble_batch(g['image1_indices'], g['image1_indices'].size, cache, g['images'], float(g['near']),
                                      float(g['far']), True)
    for k in ('rays_o', 'rays_d_ndc', 'target_rgb', 'pixel_id', 'indices_mask_nerf'):
        assert numpy.array_equal(out[k], g[f'image1_{k}']), k
    assert not bool(g['image1_has_sparse']) and numpy.array_equal(g['image1_indices'], numpy.arange(res[0] * res[1]) + res[0] * res[1])
    # candidate sets: sparse-depth pixels and the pre-crop window
    assert numpy.array_equal(numpy.where(g['sparse_depths'].reshape(-1) > 0)[0], g['sparse_candidates'])
    y0, y1, x0, x1 = batch_oracle.precrop_window(res[0], res[1], 0.5)
    domain = 3 * (y1 - y0) * (x1 - x0)
    window = batch_oracle.shuffled_indices(5, 0, 0, domain, domain, num_views=3, height=res[0], width=res[1], crop=(y0, y1, x0, x1))
    assert numpy.array_equal(numpy.sort(window), g['precrop_candidates'])
    assert int(g['after_precrop_count']) == g['precrop_candidates'].size    # the reference never leaves the crop


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for counter, key, expect in kat:
        got = batch_oracle.philox4x32_10(numpy.array(counter, dtype=numpy.uint32), key)
        assert tuple(int(v) for v in got) == expect


@pytest.mark.parametrize('domain', [1, 2, 3, 17, 1000, 9216, 2 ** 16, 2 ** 16 + 1])
def test_index_stream_is_a_permutation_per_epoch(domain):
    a = batch_oracle.shuffled_positions(11, 0, 0, domain, domain)
    assert numpy.array_equal(numpy.sort(a), numpy.arange(domain))
    pieces = numpy.concatenate([batch_oracle.shuffled_positions(11, 0, s, min(97, domain - s), domain) for s in range(0, domain, 97)])
    assert numpy.array_equal(pieces, a)                               # slicing an epoch does not change it
    if domain >= 1000:
        b = batch_oracle.shuffled_positions(11, 1, 0, domain, domain)
        assert (a == b).mean() < 0.01 and abs(float(numpy.corrcoef(a, numpy.arange(domain))[0, 1])) < 0.1


def test_draw_statistics():
    from scipy import stats
    u = batch_oracle.random_uniform(9, 4, 0, 4096, 63).reshape(-1)
    z = batch_oracle.random_normal(9, 5, 0, 4096, 63).reshape(-1)
    assert 0.0 <= u.min() and u.max() < 1.0
    assert stats.kstest(u, 'uniform').pvalue > 1e-3 and stats.kstest(z, 'norm').pvalue > 1e-3
    assert numpy.array_equal(batch_oracle.random_uniform(9, 4, 100, 50, 63), u.reshape(4096, 63)[100:150])   # row-keyed


# ---------------------------------------------------------------- config 4 (RealEstate-10K camera) end to end
@pytest.mark.parametrize('profile', ['dense', 'consistent'])
def test_render_config4_re10k_matches_reference(profile):
    """BASELINE config 4: the RE10K camera (1024x576, f = 493.9, near 1, far 133.3) through the 64+128 NDC renderer."""
    g = util.load(f'e2e_config4_{profile}.npz')
    cfg = synth.make_configs('config4')
    out = oracle.render(util.golden_params(cfg, g), cfg, util.golden_batch(g), training=False, retraw=True)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    assert sorted(out.keys()) == sorted(ref.keys())
    for k, v in ref.items():
        assert util.rel_linf(out[k], v) < 2e-5, k
    assert float(ref['depth_fine'].max()) > 20.0        # the far RE10K geometry is in the fixture


# ---------------------------------------------------------------- f3 display conversion + which outputs leave the device
def same_bits(a, b):
    a, b = numpy.ascontiguousarray(a), numpy.ascontiguousarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def test_display_oracle_matches_reference_post_processing():
    g = util.load('display.npz')
    image, depth = raygen_oracle.to_display(g['rgb'], g['depth'])
    assert same_bits(image, g['image']) and same_bits(depth, g['depth_out'])      # incl. NaN payloads and the sign of -0.0
    assert g['image'][520].tolist() == [255, 0, 0] and numpy.isnan(g['depth_out'][4]) and numpy.signbit(g['depth_out'][1])


@pytest.mark.parametrize('case,kind,ndc', [('fine_ndc', 'config2', True), ('coarse_world', 'config1', False)])
def test_inference_outputs_oracle_matches_reference(case, kind, ndc):
    g = util.load('inference_outputs.npz')
    cfg = synth.make_configs(kind)
    assert cfg['data_loader']['ndc'] == ndc
    net = {k[len(case) + 5:]: v for k, v in g.items() if k.startswith(f'{case}_net_')}
    out = raygen_oracle.retrieve_inference_outputs(cfg, (12, 20), net)
    assert list(out.keys()) == g[f'{case}_keys'].tolist()
    for k, v in out.items():
        assert same_bits(v, g[f'{case}_out_{k}']), k


# ---------------------------------------------------------------- predict_visibility (off in every shipped config)
def visibility_case(case):
    g = util.load(f'e2e_visibility_{case}.npz')
    if case == 'ndc_eval':
        cfg = synth.make_configs('config2')
        cfg['model']['coarse_mlp'] = synth.mlp_config(64, predict_visibility=True)
        cfg['model']['fine_mlp'] = synth.mlp_config(128, predict_visibility=True)
        batch = util.golden_batch(g)
    else:
        cfg = synth.with_overrides(synth.make_configs('config1'), perturb=False, raw_noise_std=0.0)
        cfg['model']['coarse_mlp'] = synth.mlp_config(64, depth=4, width=128, views_width=64, predict_visibility=True)
        batch = util.golden_batch(g)
        batch['num_frames'] = int(g['num_frames'])
        batch['common_data'] = {'poses': torch.from_numpy(g['poses'])}
    return g, cfg, batch


@pytest.mark.parametrize('case', ['ndc_eval', 'world_train'])
def test_render_with_predicted_visibility_matches_reference(case):
    """predict_visibility MLPs (views head with a 4th, visibility row; secondary view directions per sample; visibility2
    composited per ray): SimpleNeRF01.py:317-326, :646-649, :691-714, :479-482."""
    g, cfg, batch = visibility_case(case)
    params = util.golden_params(cfg, g)
    training = case == 'world_train'
    out = oracle.render(params, cfg, batch, training=training, retraw=True, sec_views_vis=True)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    assert sorted(out.keys()) == sorted(g['key_order'].tolist())
    for k, v in ref.items():
        assert util.rel_linf(out[k], v) < 2e-5, k
    assert float(ref['raw_visibility2_coarse'].std()) > 0.05
    if case == 'ndc_eval':
        blind = oracle.render(params, cfg, batch, training=False, retraw=True)
        assert sorted(blind.keys()) == sorted(g['blind_keys'].tolist())
        plain = oracle.render(params, cfg, batch, training=False, sec_views_vis=True)
        assert sorted(plain.keys()) == sorted(g['eval_keys'].tolist())

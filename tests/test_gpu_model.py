"""End-to-end parity of the drop-in model on a real MI355X against the reference's golden outputs (G6) at
north_star's tolerance -- colour-like quantities within 1e-4, depths within 1e-3 (L-infinity) -- through the
reference's own call signature ``model(input_batch[, retraw])``.

What is gated how:
  * coarse-pass outputs and both augmented models (which run on the coarse samples): every ray, every profile;
  * fine-pass outputs of the 'consistent' profile (fine MLP = coarse MLP, i.e. both passes see one geometry, as in
    a trained model): every ray in the fp32 mode.  In the f16x3 mode (whose coarse weights differ from the fp32 kernel's
    in the last bits: another accumulation order) a ray whose resampled fine depth ITSELF moved is exempt -- which rays
    sit on sample_pdf's threshold depends on the rounding, not on correctness -- and every other ray is gated;
  * fine-pass outputs when coarse and fine MLPs are two INDEPENDENT random fields: at most 2 % of rays may exceed the
    bound for the 'plain' fields and 5 % for the 'dense' ones (density head boosted until the frame is opaque -- the
    adversarial case: measured 4.5 % of 4096 fern rays against the oracle on the same host, 0.1 % on the RE10K camera), AND
    every such ray must be one whose resampled depths moved: rays whose fine depths agree with the reference's are gated
    at the full tolerance (<= 0.1 % of them may exceed it: a sample that moves by less than the 1e-5 "moved" threshold
    inside a density spike).  The observed fractions are printed in pytest's summary (util.observe).  Cause: the reference's sample_pdf replaces ``denom < 1e-5`` by 1
    (src/models/SimpleNeRF01.py:357) and the pdf of an EMPTY coarse bin is 0.9994e-5 -- within half an ulp of the
    running fp32 cumsum of that threshold -- so which empty bins collapse to their left edge is decided by the last
    bit of the reference's own sequential cumsum (SURVEY 8a row 8 measured 0.26 % of samples moving a full bin
    between two fp32 implementations).  With independent random fields a moved sample can land in dense fine
    geometry; with consistent geometry it carries ~zero weight.  ``test_fine_pass_on_reference_samples`` removes the
    resampling step and shows the fine MLP + compositing themselves match on every ray;
  * per-sample fine arrays (alpha_fine, weights_fine, raw_*_fine) are index-aligned with the reference only when the
    sorted fine depths agree to the last bit, which an independent fp32 evaluation of the coarse weights never gives
    (t = (u - cdf_b)/denom amplifies a 1e-7 cdf difference by 1/denom); they are gated, on every ray, by the
    intervention test, which feeds the reference's own fine depths to the kernels;
  * world-space depth/depth_var of NDC scenes multiply every weight by 1/(1 - z) (up to 1e3): one low-weight far
    sample that moves shifts them past 1e-3 on a few rays even with consistent geometry, so for the fine pass they
    are gated outlier-tolerantly (<= 5 % of rays; observed up to 4 rays of 128) while depth_ndc / depth_var_ndc are gated
    like the colour;
  * depth = sum(w z)/(acc + 1e-6) is only gated on rays with acc > 1e-2: for an almost-empty ray the reference's own
    alpha = 1 - exp(-1e-5) has ~1e-3 relative rounding noise.
"""
import numpy
import pytest
import torch

from oracle import nerf_oracle as oracle
from simplenerf_amd import ops, synth
from simplenerf_amd.models.ModelFactory import get_model
from tests import util
from tests.test_gpu_kernels import abi_param_list

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
RGB_TOL, DEPTH_TOL = 1e-4, 1e-3
MAX_OUTLIER_RAYS = 0.02          # fine colour / acc / NDC depth over tolerance; observed <= 0.8 % ('plain', 'consistent')
MAX_OUTLIER_RAYS_DENSE = 0.05    # ... for two independent OPAQUE random fields ('dense').  Pinned to the reference's own noise
#                                  (tests/golden/selfnoise.json, 4096 rays of config 2): the reference with relabelled hidden
#                                  units puts 2.8 % of these rays over the bounds, its fp32 run against its fp64 run 4.2 %
MAX_OUTLIER_RAYS_WORLD = 0.05    # world-space depth / depth_var of NDC scenes (weights x 1/(1-z) <= 1e3); observed <= 3.1 %
MAX_UNMOVED_OVER = 0.001         # ... of which on rays whose fine depths agree with the reference's to 1e-5; observed <= 1/4096


def build(configs, golden, precision='fp32'):
    configs = synth.with_overrides(configs, hip_precision=precision)
    model = get_model(configs, None)
    res = model.load_state_dict(util.golden_params(configs, golden), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.to(DEV)


def per_ray_violation(key, got, ref, acc_ref, ndc=True):
    """bool (N,): rays on which `key` exceeds its bound.  None -> key not gated.  ``ndc``: the scene is rendered in NDC, i.e.
    its world-space depths are the NDC ones pushed through 1/(1 - z) and reach ~1e3 -- those (and only those) are bounded
    relative to their magnitude; world depths of a non-NDC scene are gated at north_star's ABSOLUTE 1e-3 (round 4)."""
    base = key.replace('points_augmentation_', '').replace('views_augmentation_', '')
    a = got.detach().cpu().numpy().astype(numpy.float64)
    b = ref.astype(numpy.float64)
    n = b.shape[0]
    d = numpy.abs(a - b).reshape(n, -1)
    bm = numpy.abs(b).reshape(n, -1)
    if base.startswith(('rgb_', 'acc_', 'alpha_', 'raw_rgb', 'weights_', 'visibility_')):
        return d.max(1) > RGB_TOL
    if base.startswith('raw_sigma'):
        return (d / numpy.maximum(bm, 1.0)).max(1) > 1e-3
    if base.startswith(('depth_var_ndc', 'depth_ndc')):
        return (d.max(1) > DEPTH_TOL) & (acc_ref > 1e-2)
    if base.startswith('depth_var'):
        if not ndc:
            return (d.max(1) > DEPTH_TOL) & (acc_ref > 1e-2)
        # un-normalised second moment sum w (z - depth)^2 of world depths up to ~1e3: a difference of large numbers
        # that no reference loss reads; bounded relatively, 10x looser
        return ((d / numpy.maximum(bm, 1.0)).max(1) > 10 * DEPTH_TOL) & (acc_ref > 1e-2)
    if base.startswith('depth_'):
        if not ndc:
            return (d.max(1) > DEPTH_TOL) & (acc_ref > 1e-2)
        return ((d / numpy.maximum(bm, 1.0)).max(1) > DEPTH_TOL) & (acc_ref > 1e-2)
    return None  # z_vals_*: checked separately


def check_outputs(out, ref, strict_fine, tag='', precision='fp32', dense=False):
    assert sorted(out.keys()) == sorted(ref.keys()), sorted(set(out) ^ set(ref))
    for k, v in ref.items():
        assert tuple(out[k].shape) == tuple(v.shape), k
        assert out[k].dtype == torch.float32 and out[k].device.type == 'cuda', k
        assert torch.isfinite(out[k]).all(), k
    ndc = 'depth_ndc_coarse' in ref
    moved_rays = numpy.zeros(next(iter(ref.values())).shape[0], dtype=bool)
    if 'z_vals_fine' in ref:
        zr = ref['z_vals_fine']
        moved_rays = (numpy.abs(out['z_vals_fine'].cpu().numpy() - zr) > 1e-5 * float(numpy.abs(zr).max())).any(1)
    worst_key, worst_frac, strict_bad = '', 0.0, 0
    for k, v in ref.items():
        level = 'fine' if k.endswith('_fine') else 'coarse'
        acc_key = next(c for c in (f'{p}acc_{level}' for p in ('points_augmentation_', 'views_augmentation_', ''))
                       if k.startswith(c.split('acc_')[0]) and c in ref)
        bad = per_ray_violation(k, out[k], v, ref[acc_key].astype(numpy.float64), ndc)
        if bad is None:
            continue
        per_sample = v.ndim >= 2 and v.shape[1] > 3
        world_depth = k.replace('points_augmentation_', '').replace('views_augmentation_', '') in ('depth_fine', 'depth_var_fine')
        if level == 'coarse':
            assert not bad.any(), (tag, k, int(bad.sum()), util.linf(out[k], v))
        elif per_sample:
            continue  # not index-aligned unless the fine depths are bit-identical: see test_fine_pass_on_reference_samples
        elif strict_fine and not world_depth:
            if precision == 'f16x3':
                # every ray within tolerance, except rays on which a resampled fine depth itself moved: sample_pdf's
                # `denom < 1e-5` branch (src/models/SimpleNeRF01.py:357) flips on a last-bit difference of the coarse
                # weights, and the 16x16x32 kernel accumulates in another order than the fp32 one -- with the
                # reference's fine depths pinned these rays agree to 1e-7 (test_fine_pass_on_reference_samples)
                assert not (bad & ~moved_rays).any(), (tag, k, int((bad & ~moved_rays).sum()), util.linf(out[k], v))
                assert bad.mean() <= MAX_OUTLIER_RAYS, (tag, k, float(bad.mean()))
            else:
                assert not bad.any(), (tag, k, int(bad.sum()), util.linf(out[k], v))       # fp32: every ray
            strict_bad = max(strict_bad, int(bad.sum()))
        else:
            bound = MAX_OUTLIER_RAYS_WORLD if world_depth else (MAX_OUTLIER_RAYS_DENSE if dense else MAX_OUTLIER_RAYS)
            assert bad.mean() <= bound, (tag, k, float(bad.mean()))
            # ... and the outliers are rays whose resampled depths moved, not rays the kernels got wrong
            assert (bad & ~moved_rays).mean() <= MAX_UNMOVED_OVER, (tag, k, int((bad & ~moved_rays).sum()))
            if float(bad.mean()) > worst_frac:
                worst_key, worst_frac = k, float(bad.mean())
    assert util.linf(out['z_vals_coarse'], ref['z_vals_coarse']) == 0.0
    if 'z_vals_fine' in ref:
        zr = ref['z_vals_fine']
        moved = util.outlier_fraction(out['z_vals_fine'], zr, 1e-5 * float(numpy.abs(zr).max()))
        assert moved < 0.01
        z = out['z_vals_fine']
        assert torch.all(z[:, 1:] >= z[:, :-1])
        n = zr.shape[0]
        util.observe(tag, f'fine depths moved {moved:.5f} of samples [0.01], rays with a moved depth {int(moved_rays.sum())}/{n}; '
                          + (f'strict fine gate: {strict_bad} rays over tolerance [{"only moved rays" if precision == "f16x3" else "0"}]'
                             if strict_fine else f'rays over tolerance, worst fine key {worst_key or "-"}: {worst_frac:.4f} [{MAX_OUTLIER_RAYS_DENSE if dense else MAX_OUTLIER_RAYS}; world depth {MAX_OUTLIER_RAYS_WORLD}]'))


EVAL_CASES = [(k, p) for k in ('config1', 'config2', 'headline', 'headline_world') for p in ('plain', 'dense', 'consistent')
              if not (k == 'config1' and p == 'consistent')] + [('config4', 'dense'), ('config4', 'consistent')]


@pytest.mark.parametrize('kind,profile', EVAL_CASES)
@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
def test_eval_forward_matches_reference(kind, profile, precision):
    g = util.load(f'e2e_{kind}_{profile}.npz')
    cfg = synth.make_configs(kind)
    model = build(cfg, g, precision).eval()
    batch = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
    before = {k: v.clone() for k, v in batch.items()}
    with torch.no_grad():
        out = model(batch, retraw=True)
        plain = model(batch)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    check_outputs(out, ref, strict_fine=(profile == 'consistent'), tag=f'{kind}/{profile}/{precision}', precision=precision,
                  dense=(profile == 'dense'))
    assert sorted(plain.keys()) == sorted(g['eval_keys'].tolist())
    assert all(torch.equal(plain[k], out[k]) for k in plain)
    assert list(batch.keys()) == list(before.keys()) and all(torch.equal(batch[k], before[k]) for k in batch)


TRAIN_CASES = [('det', 'dense'), ('rand', 'dense'), ('rand', 'plain'), ('det', 'consistent'), ('rand', 'consistent')]


@pytest.mark.parametrize('variant,profile', TRAIN_CASES)
def test_train_forward_matches_reference(variant, profile):
    """Training-mode forward (both augmented MLPs active, config 3) with the reference's CPU-generator draws
    (stratified jitter, inverse-CDF u, density noise) replayed in its order and injected."""
    g = util.load(f'e2e_config3_train_{variant}_{profile}.npz')
    cfg = synth.with_overrides(synth.make_configs('config3'), perturb=bool(g['perturb']),
                               raw_noise_std=float(g['raw_noise_std']))
    model = build(cfg, g).train()
    batch = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
    draws = oracle.replay_reference_draws(cfg, batch['rays_o'].shape[0], int(g['torch_seed']))
    assert len(draws) == 1
    model.set_random_draws(draws[0])
    with torch.no_grad():
        out = model(batch)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    check_outputs(out, ref, strict_fine=(profile == 'consistent'), tag=f'train/{variant}/{profile}', dense=(profile == 'dense'))


@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
def test_train_forward_with_fine_augmentation_mlps_matches_reference(precision):
    """config3f: six MLP passes per training forward (main, points-aug, views-aug at the coarse AND the fine level)."""
    g = util.load('e2e_config3f_train_det_consistent.npz')
    cfg = synth.with_overrides(synth.make_configs('config3f'), perturb=False, raw_noise_std=0.0)
    model = build(cfg, g, precision).train()
    with torch.no_grad():
        out = model({k: v.to(DEV) for k, v in util.golden_batch(g).items()})
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    assert 'points_augmentation_rgb_fine' in out and 'views_augmentation_depth_fine' in out
    # the main fine model shares the coarse weights ('consistent'), but the fine AUGMENTATION fields are independent
    # random MLPs: a resampled depth that sits in another bin than the reference's (sample_pdf discontinuity, DESIGN.md
    # section 4) can land in their dense geometry, so fine-level outputs are gated per ray with the outlier allowance
    check_outputs(out, ref, strict_fine=False, tag=f'train/config3f/{precision}', precision=precision, dense=True)


@pytest.mark.parametrize('kind,profile', [('config2', 'dense'), ('headline', 'dense'), ('headline_world', 'dense'),
                                          ('config2', 'plain'), ('config4', 'dense')])
def test_fine_pass_on_reference_samples(kind, profile):
    """Intervention: give the fine MLP + compositing kernels the REFERENCE's fine depths (skipping only the
    rounding-sensitive resampling) -- every ray must then match at the full tolerance, for independent fields too."""
    g = util.load(f'e2e_{kind}_{profile}.npz')
    cfg = synth.make_configs(kind)
    params = {k: v.to(DEV) for k, v in util.golden_params(cfg, g).items()}
    mlp = ops.PackedMlp(cfg['model']['fine_mlp'], DEV)
    mlp.pack(abi_param_list(params, 'fine_model.'))
    b = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
    ndc = cfg['data_loader']['ndc']
    mo, md = (b['rays_o_ndc'], b['rays_d_ndc']) if ndc else (b['rays_o'], b['rays_d'])
    z = torch.from_numpy(g['out_z_vals_fine']).to(DEV)
    sigma, rgb = mlp.forward(mo, md, b['view_dirs'], z)
    comp = ops.composite(sigma, rgb, z, md, ndc, False, b['rays_o'], b['rays_d'])
    acc = g['out_acc_fine'].astype(numpy.float64)
    for k, v in comp.items():
        bad = per_ray_violation(f'{k}_fine', v, g[f'out_{k}_fine'], acc, ndc)
        assert bad is not None and not bad.any(), (k, util.linf(v, g[f'out_{k}_fine']))
    assert not per_ray_violation('raw_sigma_fine', sigma, g['out_raw_sigma_fine'], acc).any()
    assert not per_ray_violation('raw_rgb_fine', rgb, g['out_raw_rgb_fine'], acc).any()


def test_train_forward_with_device_rng_is_statistically_sane():
    cfg = synth.make_configs('config3')
    g = util.load('e2e_config3_train_rand_plain.npz')
    model = build(cfg, g).train()
    batch = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
    with torch.no_grad():
        a = model(batch)
        b = model(batch)
    assert not torch.equal(a['z_vals_coarse'], b['z_vals_coarse'])
    z = a['z_vals_coarse']
    assert torch.all(z[:, 1:] >= z[:, :-1]) and float(z.min()) >= 0 and float(z.max()) <= 1
    assert all(torch.isfinite(v).all() for v in a.values())
    # density noise ~ N(0, raw_noise_std): the two draws of the pre-ReLU perturbation differ on most samples
    assert float((a['raw_sigma_coarse'] != b['raw_sigma_coarse']).float().mean()) > 0.3


def test_training_draws_do_not_depend_on_how_rays_are_sharded():
    """Philox draws are keyed by (seed, training call, kind, row_offset + ray, sample): two ranks that each take half of
    a batch produce, row for row, what one process produces for the whole batch -- and another seed does not."""
    cfg = synth.make_configs('config3')
    g = util.load('e2e_config3_train_rand_plain.npz')
    batch = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
    n = batch['rays_o'].shape[0]
    half = n // 2
    keys = ('rgb_fine', 'depth_fine', 'z_vals_coarse', 'z_vals_fine', 'raw_sigma_coarse', 'points_augmentation_rgb_coarse',
            'views_augmentation_depth_coarse')
    with torch.no_grad():
        whole = build(cfg, g).train()(batch)
        parts = []
        for lo, hi in ((0, half), (half, n)):
            shard = {k: v[lo:hi] for k, v in batch.items()}
            shard['row_offset'] = lo
            parts.append(build(cfg, g).train()(shard))
        other = build({**cfg, 'seed': 1}, g).train()(batch)
    for k in keys:
        assert torch.equal(torch.cat([p[k] for p in parts]), whole[k]), k
    assert not torch.equal(other['z_vals_coarse'], whole['z_vals_coarse'])


def test_cpu_tensors_are_rejected_not_silently_computed():
    cfg = synth.make_configs('config1')
    model = get_model(cfg, None).eval()  # parameters left on the CPU
    batch = {k: torch.from_numpy(v) for k, v in synth.random_world_rays(8).items()}
    with pytest.raises(RuntimeError, match='GPU'):
        with torch.no_grad():
            model(batch)


# ---------------------------------------------------------------- configuration switches and edge cases vs the oracle
def _oracle_vs_model(cfg, batch_np, training=False, draws=None, retraw=True):
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 11, 150.0, 4.0).items()}
    model.load_state_dict(sd)
    model = model.to(DEV).train(training)
    batch = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    if draws is not None:
        model.set_random_draws(draws)
    with torch.no_grad():
        out = model({k: v.to(DEV) for k, v in batch.items()}, retraw=retraw)
    ref = oracle.render(sd, cfg, batch, training=training, retraw=retraw, rand_per_chunk=None if draws is None else [draws])
    return out, ref


@pytest.mark.parametrize('lindisp,white', [(True, False), (False, True), (True, True)])
def test_lindisp_and_white_background_match_oracle(lindisp, white):
    """model.lindisp (:286-289) and model.white_bkgd (:462-463) are off in the shipped configs; checked against the
    oracle on world rays (coarse-only config 1, so no resampling sensitivity)."""
    cfg = synth.with_overrides(synth.make_configs('config1'), lindisp=lindisp, white_bkgd=white)
    out, ref = _oracle_vs_model(cfg, synth.random_world_rays(257, seed=3))
    assert sorted(out) == sorted(ref)
    for k, v in ref.items():
        tol = RGB_TOL if not k.startswith('depth') else DEPTH_TOL * max(1.0, float(v.abs().max()))
        assert util.linf(out[k], v) <= tol, k


def test_zero_rays_and_single_ray():
    cfg = synth.make_configs('config2')
    model = get_model(cfg, None).to(DEV).eval()
    cam = synth.camera('fern', 0)
    from simplenerf_amd import harness
    with torch.no_grad():
        one = model(harness.frame_batch(cam, True, DEV, 1234, 1))
        none = model(harness.frame_batch(cam, True, DEV, 1234, 0))
    assert one['rgb_fine'].shape == (1, 3) and one['alpha_fine'].shape == (1, 192)
    assert none['rgb_fine'].shape == (0, 3) and none['alpha_coarse'].shape == (0, 64)
    assert all(torch.isfinite(v).all() for v in one.values())


def test_extra_batch_keys_are_ignored_and_common_data_tolerated():
    cfg = synth.make_configs('config1')
    out_a, _ = _oracle_vs_model(cfg, synth.random_world_rays(64, seed=9))
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 11, 150.0, 4.0).items()})
    model = model.to(DEV).eval()
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.random_world_rays(64, seed=9).items()}
    batch.update({'target_rgb': torch.rand(64, 3, device=DEV), 'iter_num': 7, 'indices': numpy.arange(64),
                  'common_data': {'poses': torch.eye(4, device=DEV)[None, None].repeat(1, 2, 1, 1)}})
    with torch.no_grad():
        out_b = model(batch, retraw=True)
    assert all(torch.equal(out_a[k], out_b[k]) for k in out_a)


def test_full_frame_render_equals_blockwise_and_sharded_ranges():
    """harness.render_frame (65 536-ray blocks, device-side ray generation) == one call on the same rays; and the union of
    the per-rank shard ranges (what each GPU renders before the gather) == the unsharded frame."""
    from simplenerf_amd import harness
    cfg = synth.make_configs('config1')
    cfg['data_loader']['ndc'] = False
    model = get_model(cfg, None).to(DEV).eval()
    cam = synth.camera('fern', 1, downscale=8)  # 94 x 126 = 11 844 rays
    h, w = cam['resolution']
    n = h * w
    keys = ('rgb_coarse', 'depth_coarse')
    whole = harness.render_frame(model, cam, False, DEV, keys=keys, ray_block=4096)
    with torch.no_grad():
        direct = model(harness.frame_batch(cam, False, DEV))
    parts = [harness.render_rays_blockwise(model, cam, False, DEV, *harness.shard_range(n, r, 3), keys=keys) for r in range(3)]
    for k in keys:
        assert whole[k].shape[0] == n and torch.equal(whole[k], direct[k])
        assert torch.equal(torch.cat([p[k] for p in parts], 0), direct[k])
    img, dep = harness.to_display(whole['rgb_coarse'], whole['depth_coarse'])
    assert img.shape == (n, 3) and img.dtype == torch.uint8 and float(dep.min()) >= 0


def test_full_size_frame_properties_and_eight_way_shards():
    """BASELINE config 2/4 at full size (fern 1008 x 756 = 762 048 rays, 64+128 samples, 8x256 coarse+fine) through
    size-independent properties: sorted depths inside [0,1] whose fine set contains every coarse depth, non-negative
    weights summing to acc <= 1, colours inside [0, acc], NDC depth inside [0,1]; the union of the eight per-rank ray
    ranges of the sharded render is bit-identical to the unsharded frame; the f16x3 kernels agree with the fp32 ones to
    the parity tolerance on all but the few rays whose resampling hit the sample_pdf discontinuity."""
    from simplenerf_amd import harness
    cfg = synth.make_configs('config2')
    g = util.load('e2e_config2_consistent.npz')
    model = build(cfg, g).eval()
    cam = synth.camera('fern', 0)
    h, w = cam['resolution']
    n = h * w
    assert n == 762048
    rgb_blocks = []
    with torch.no_grad():
        for start in range(0, n, 65536):
            count = min(65536, n - start)
            out = model(harness.frame_batch(cam, True, DEV, start, count), retraw=True)
            zc, zf, wf, acc, rgb = out['z_vals_coarse'], out['z_vals_fine'], out['weights_fine'], out['acc_fine'], out['rgb_fine']
            assert zc.shape == (count, 64) and zf.shape == (count, 192)
            assert bool((zf[:, 1:] >= zf[:, :-1]).all()) and float(zf.min()) >= 0.0 and float(zf.max()) <= 1.0
            pos = torch.searchsorted(zf, zc.contiguous()).clamp(max=191)
            assert torch.equal(torch.gather(zf, 1, pos), zc)                     # the merged set keeps every coarse depth
            assert float(wf.min()) >= 0.0 and float((wf.sum(1) - acc).abs().max()) <= 1e-5 and float(acc.max()) <= 1.0 + 1e-5
            assert float(rgb.min()) >= 0.0 and bool((rgb <= acc[:, None] + 1e-5).all())
            assert float(out['depth_ndc_fine'].min()) >= 0.0 and float(out['depth_ndc_fine'].max()) <= 1.0 + 1e-5
            assert all(torch.isfinite(v).all() for v in out.values())
            rgb_blocks.append(rgb)
    whole = torch.cat(rgb_blocks)
    shards = [harness.render_rays_blockwise(model, cam, True, DEV, *harness.shard_range(n, r, 8), keys=('rgb_fine',))['rgb_fine']
              for r in range(8)]
    assert [s.shape[0] for s in shards] == [95256] * 8 and torch.equal(torch.cat(shards), whole)
    fast = build(cfg, g, 'f16x3').eval()
    other = harness.render_frame(fast, cam, True, DEV, keys=('rgb_fine',))['rgb_fine']
    off = ((other - whole).abs().max(1)[0] > 1e-4).float().mean()
    assert float(off) < 0.01, float(off)


@pytest.mark.parametrize('resolution', [None, (756, 1008)], ids=['native1024x576', 'named1008x756'])
def test_config4_re10k_full_frame_properties_and_eight_way_shards(resolution):
    """BASELINE config 4 at full size: the RealEstate-10K camera (runs/training/train0011/00000/ModelConfigs.json: f = 493.9,
    near 1, far 133.3) at its own 1024 x 576 = 589 824 rays and at the 1008 x 756 = 762 048 rays BASELINE names, 64+128
    samples, 8x256 coarse+fine, with the weights of the config-4 golden.  Size-independent properties on every block;
    the golden's own pixels re-rendered inside the full frame reproduce the fixture (native size: same camera); the
    union of the eight per-rank ray ranges is bit-identical to the unsharded frame, and so is the display conversion
    of the gathered frame (harness.predict_frame's single-rank path)."""
    from simplenerf_amd import harness
    cfg = synth.make_configs('config4')
    g = util.load('e2e_config4_consistent.npz')
    model = build(cfg, g).eval()
    cam = synth.camera('re10k', 0, resolution=resolution)
    h, w = cam['resolution']
    n = h * w
    assert n == (589824 if resolution is None else 762048)
    keys = ('rgb_fine', 'depth_fine', 'depth_var_fine', 'depth_ndc_fine', 'depth_var_ndc_fine')
    blocks = {k: [] for k in keys}
    with torch.no_grad():
        for start in range(0, n, 65536):
            count = min(65536, n - start)
            out = model(harness.frame_batch(cam, True, DEV, start, count), retraw=True)
            zc, zf, wf, acc, rgb = out['z_vals_coarse'], out['z_vals_fine'], out['weights_fine'], out['acc_fine'], out['rgb_fine']
            assert zc.shape == (count, 64) and zf.shape == (count, 192)
            assert bool((zf[:, 1:] >= zf[:, :-1]).all()) and float(zf.min()) >= 0.0 and float(zf.max()) <= 1.0
            pos = torch.searchsorted(zf, zc.contiguous()).clamp(max=191)
            assert torch.equal(torch.gather(zf, 1, pos), zc)
            assert float(wf.min()) >= 0.0 and float((wf.sum(1) - acc).abs().max()) <= 1e-5 and float(acc.max()) <= 1.0 + 1e-5
            assert float(rgb.min()) >= 0.0 and bool((rgb <= acc[:, None] + 1e-5).all())
            assert float(out['depth_ndc_fine'].min()) >= 0.0 and float(out['depth_ndc_fine'].max()) <= 1.0 + 1e-5
            assert float(out['depth_fine'][acc > 0.5].min()) > 0.0           # world depth: in front of the camera
            assert all(torch.isfinite(v).all() for v in out.values())
            for k in keys:
                blocks[k].append(out[k])
    whole = {k: torch.cat(v) for k, v in blocks.items()}
    if resolution is None:
        pix = torch.from_numpy(g['pixel_indices']).to(DEV)
        assert util.linf(whole['rgb_fine'][pix], g['out_rgb_fine']) <= RGB_TOL
        assert util.linf(whole['depth_ndc_fine'][pix], g['out_depth_ndc_fine']) <= DEPTH_TOL
    shards = [harness.render_rays_blockwise(model, cam, True, DEV, *harness.shard_range(n, r, 8), keys=keys) for r in range(8)]
    assert [s['rgb_fine'].shape[0] for s in shards] == [n // 8] * 8
    for k in keys:
        assert torch.equal(torch.cat([s[k] for s in shards]), whole[k]), k
    frame = harness.predict_frame(model, cfg, cam, DEV)
    assert list(frame) == ['image', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc']
    assert frame['image'].shape == (h, w, 3) and frame['image'].dtype == numpy.uint8
    ref_img, ref_depth = oracle_display(whole['rgb_fine'].cpu().numpy(), whole['depth_fine'].cpu().numpy())
    assert numpy.array_equal(frame['image'].reshape(-1, 3), ref_img) and numpy.array_equal(frame['depth'].reshape(-1), ref_depth)


K_SELF = 1.25         # allowance vs the committed fp32 reference outputs = K_SELF x the reference's own evaluation-order noise (1.5 until round 4; VERDICT r4: observed <= 1.12 x)
K_EXACT = 1.25        # allowance vs the reference in double precision = K_EXACT x the reference's own fp32 error against it
SLACK_RAYS = 4        # + 4 rays of 4096 (0.1 %) on either gate: fractions of a few rays are counting noise


@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
@pytest.mark.parametrize('kind,profile', [('config2', 'dense'), ('config2', 'consistent'), ('config4', 'dense'), ('config4', 'consistent')])
def test_frame_slice_against_committed_reference_outputs(kind, profile, precision):
    """A 4096-ray slice out of the middle of the BASELINE config 2 (fern 1008x756) and config 4 (RE10K camera at 1008x756)
    frames, rays generated on the device, against the REFERENCE's outputs for exactly these rays
    (tests/golden/slice_<config>_<profile>.npz, written by tools/make_golden_selfnoise.py) -- no CPU arithmetic of the GPU box
    is involved (until round 3 the comparison was against the oracle evaluated on that box).  The allowances are pinned to the
    reference's own irreproducibility, measured by the same tool (tests/golden/selfnoise.json): its fp32 CPU path is
    bit-identical across thread counts, GEMM back ends, vector widths, chunk sizes and row order, so the figure that matters
    is what ANOTHER SUMMATION ORDER of the same function does to it -- the reference with its hidden units relabelled --
    and how far its fp32 run is from its own double-precision run.  Gates:
      * coarse outputs: every ray at the full tolerance; coarse depths bit-equal;
      * fine colour / opacity / NDC depth vs the committed fp32 outputs: rays over tolerance <= K_SELF x (relabelled
        reference vs reference) + 0.1 %, and (all but 0.1 % of) them on rays whose resampled depths moved  [round 4, first
        run: config 2 'dense' 2.9 % against the relabelled reference's 2.8 %; 4.5 % in round 3, before K5 summed in torch's
        orders];
      * the same outputs vs the reference in DOUBLE precision: rays over tolerance <= K_EXACT x the reference's own fp32 run
        against it + 0.1 % -- the kernels are as close to the exact value as the reference is."""
    import json
    import os
    from simplenerf_amd import harness
    g = util.load(f'e2e_{kind}_{profile}.npz')
    fixture = util.load(f'slice_{kind}_{profile}.npz')
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'selfnoise.json')) as f:
        noise = json.load(f)['cases'][f'{kind}/{profile}']
    cfg = synth.make_configs(kind)
    cam = synth.camera('fern', 0) if kind == 'config2' else synth.camera('re10k', 0, resolution=(756, 1008))
    first, count = int(fixture['first_ray']), int(fixture['count'])
    model = build(cfg, g, precision).eval()
    batch = harness.frame_batch(cam, True, DEV, first, count)
    with torch.no_grad():
        out = model(batch, retraw=True)
    ref = {k[4:]: v for k, v in fixture.items() if k.startswith('out_')}
    exact = {k[4:]: v for k, v in fixture.items() if k.startswith('f64_')}
    assert float(ref['acc_fine'].mean()) > 0.05, 'the slice must not be empty space'
    acc = {lvl: ref[f'acc_{lvl}'].astype(numpy.float64) for lvl in ('coarse', 'fine')}
    assert util.linf(out['z_vals_coarse'], ref['z_vals_coarse']) == 0.0       # same rays, same coarse depths, to the bit
    for k in ('rgb_coarse', 'acc_coarse', 'depth_ndc_coarse', 'weights_coarse'):
        bad = per_ray_violation(k, out[k], ref[k], acc['coarse'])
        assert not bad.any(), (k, int(bad.sum()), util.linf(out[k], ref[k]))
    zr = ref['z_vals_fine']
    moved = numpy.abs(out['z_vals_fine'].cpu().numpy() - zr) > 1e-5
    moved_rays = moved.any(1)

    def over_tolerance(target):
        flags = numpy.zeros(count, dtype=bool)
        for k in ('rgb_fine', 'acc_fine', 'depth_ndc_fine'):
            flags |= per_ray_violation(k, out[k], target[k], target['acc_fine'].astype(numpy.float64))
        return flags

    over, over_exact = over_tolerance(ref), over_tolerance(exact)
    self_noise = noise['t8_unitperm']['rays_over_1e-4_rgb_acc_or_1e-3_ndc_depth']
    bound = K_SELF * self_noise + SLACK_RAYS / count
    bound_exact = K_EXACT * noise['canonical_vs_f64']['rays_over_1e-4_rgb_acc_or_1e-3_ndc_depth'] + SLACK_RAYS / count
    unmoved_allowed = 0 if profile == 'consistent' else int(MAX_UNMOVED_OVER * count)
    util.observe(f'slice/{kind}/{profile}/{precision}',
                 f'vs committed reference fp32: rays over tol {int(over.sum())}/{count} = {over.mean():.5f} [{bound:.5f} = {K_SELF} x '
                 f'{self_noise:.5f} + {SLACK_RAYS} rays], of them with unmoved fine depths {int((over & ~moved_rays).sum())} '
                 f'[{unmoved_allowed}]; vs reference fp64: {over_exact.mean():.5f} [{bound_exact:.5f}; the reference\'s own fp32 run: '
                 f'{noise["canonical_vs_f64"]["rays_over_1e-4_rgb_acc_or_1e-3_ndc_depth"]:.5f}]; fine samples moved {moved.mean():.5f} '
                 f'(relabelled reference: {noise["t8_unitperm"]["samples_moved"]:.5f}), rays with a moved depth {moved_rays.mean():.4f} '
                 f'({noise["t8_unitperm"]["rays_with_a_moved_depth"]:.4f})')
    assert over.mean() <= bound, (float(over.mean()), bound)
    assert (over & ~moved_rays).sum() <= unmoved_allowed, int((over & ~moved_rays).sum())
    assert over_exact.mean() <= bound_exact, (float(over_exact.mean()), bound_exact)
    # the resampling itself moves no more samples than relabelling the reference's hidden units does (x K_SELF)
    assert moved.mean() <= K_SELF * noise['t8_unitperm']['samples_moved'] + 1e-4, float(moved.mean())


def oracle_display(rgb, depth):
    from oracle import raygen_oracle
    return raygen_oracle.to_display(rgb, depth)


# ---------------------------------------------------------------- predict_visibility (off in every shipped config)
@pytest.mark.parametrize('case', ['ndc_eval', 'world_train'])
def test_predicted_visibility_matches_reference(case):
    """predict_visibility MLPs through the drop-in model: raw_visibility_*, raw_visibility2_* and the composited
    visibility2_* (src/models/SimpleNeRF01.py:317-326, :646-649, :691-714, :479-482) against the reference's outputs;
    'ndc_eval' takes rays_o2 from the batch (Tester path), 'world_train' derives it from common_data poses / pixel_id /
    num_frames (Trainer path, training mode).  Every other output keeps its tolerance."""
    from tests.test_oracle_golden import visibility_case
    g, cfg, batch = visibility_case(case)
    training = case == 'world_train'
    model = build(cfg, g).train(training)

    def to_dev(v):
        if isinstance(v, torch.Tensor):
            return v.to(DEV)
        if isinstance(v, dict):
            return {k: to_dev(x) for k, x in v.items()}
        return v

    dev_batch = {k: to_dev(v) for k, v in batch.items()}
    if training:
        dev_batch['common_data'] = {'poses': dev_batch['common_data']['poses'][None]}     # the loader's replica axis
    with torch.no_grad():
        out = model(dev_batch, retraw=True, sec_views_vis=True)
    ref = {k[4:]: v for k, v in g.items() if k.startswith('out_')}
    vis_keys = [k for k in ref if 'visibility2' in k or k.startswith('raw_visibility_')]
    assert len(vis_keys) >= 3
    for k in vis_keys:
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        if k.endswith('_fine') and ref[k].ndim > 2:
            continue        # per-sample fine arrays are index-aligned only on identical fine depths (see the intervention below)
        assert util.linf(out[k], ref[k]) <= RGB_TOL, (k, util.linf(out[k], ref[k]))
    check_outputs({k: v for k, v in out.items() if k not in vis_keys}, {k: v for k, v in ref.items() if k not in vis_keys},
                  strict_fine=True, tag=f'visibility/{case}')
    if case == 'ndc_eval':
        # the fine pass on the reference's own fine depths: every per-sample visibility matches
        model.set_random_draws({'z_vals_fine': torch.from_numpy(ref['z_vals_fine'])})
        with torch.no_grad():
            pinned = model(dev_batch, retraw=True, sec_views_vis=True)
            blind = model(dev_batch, retraw=True)
            plain = model(dev_batch, sec_views_vis=True)
        for k in ('raw_visibility_fine', 'raw_visibility2_fine', 'visibility2_fine'):
            assert util.linf(pinned[k], ref[k]) <= RGB_TOL, (k, util.linf(pinned[k], ref[k]))
        # the injected fine depths come back as z_vals_fine (with secondary views the model returned rays_o2 here)
        assert tuple(pinned['z_vals_fine'].shape) == tuple(ref['z_vals_fine'].shape)
        assert util.linf(pinned['z_vals_fine'], ref['z_vals_fine']) == 0.0
        assert sorted(blind.keys()) == sorted(g['blind_keys'].tolist())
        assert sorted(plain.keys()) == sorted(g['eval_keys'].tolist())
    else:
        # training mode with gradients: the visibility outputs carry none, the others still train every parameter, and
        # the visibility row of the views head gets exact zeros
        model.zero_grad(set_to_none=True)
        out = model(dev_batch)
        assert not out['visibility2_coarse'].requires_grad and not out['raw_visibility_coarse'].requires_grad
        (out['rgb_coarse'] ** 2).mean().backward()
        w = model.coarse_model.views_output_linear
        assert float(w.weight.grad[3].abs().max()) == 0.0 and float(w.bias.grad[3]) == 0.0
        assert float(w.weight.grad[:3].abs().max()) > 0.0


def test_predict_visibility_is_refused_in_the_fp16_modes():
    cfg = synth.with_overrides(synth.make_configs('config1'), hip_precision='f16x3')
    cfg['model']['coarse_mlp'] = synth.mlp_config(64, depth=4, width=128, views_width=64, predict_visibility=True)
    model = get_model(cfg, None).to(DEV).eval()
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.random_world_rays(8).items()}
    with pytest.raises(NotImplementedError, match='fp32'):
        with torch.no_grad():
            model(batch)

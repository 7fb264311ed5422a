"""The 16-bit mode (SNERF_PRECISION_F16, configs['model']['hip_precision'] = 'f16') on a real MI355X.

This is the throughput variant BASELINE config 5 names ("bf16" training step; SURVEY 8d: "parity ... at a stated bf16
tolerance plus an fp32 run at 1e-4/1e-3"): one fp16 MFMA per product (11 significand bits per operand, fp32 accumulate,
fp32 master weights / biases / heads / outputs), activations saved as fp16 and layer gradients as bf16.  It is NOT inside
north_star's 1e-4 / 1e-3 bar -- that is what 'fp32' and 'f16x3' are for, with their own parity tests -- so its tolerances
are stated here, each a few times what is observed:

    MLP outputs vs the fp32 oracle        sigma 5e-3 relative to max, rgb 2e-4 absolute    (observed 9e-4 / 2e-5)
    parameter gradients vs autograd        15 % relative L2 per tensor on the 315-sample spiky-gradient case, where a
                                           handful of flipped ReLU masks dominate           (observed 1-8 %)
    rendered colour / NDC depth vs fp32    1e-3 / 5e-3 on 2048 headline rays                 (observed 1.4e-4 / 6e-4)
    training batch vs fp32 (9 losses)      every loss value 5e-3 relative, every accumulated parameter gradient
                                           5 % relative L2                                  (observed 9e-4 / 1.6 %)
    short training run                     same PSNR as the fp32 run to 1 dB

Properties that hold exactly are tested exactly: eval and training forward give the same bits, results are
bit-reproducible, and the backward is exactly linear in a power-of-two loss scale."""
import math
import os

import numpy
import pytest
import torch

from oracle import nerf_oracle as oracle
from simplenerf_amd import harness, ops, optim, synth
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer
from simplenerf_amd.models.ModelFactory import get_model
from tests import util
from tests.test_gpu_grads import rel_l2, rel_to_max
from tests.test_gpu_kernels import LAYOUTS, abi_param_list

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
F16 = ops.PRECISIONS['f16']


def mlp_case(layout, size, n=7, s=45):
    depth, width, vwidth = size
    cfg = synth.mlp_config(64, depth=depth, width=width, views_width=vwidth, **LAYOUTS[layout])
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 31, 50.0, 1.0)
    rng = numpy.random.RandomState(depth)
    o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32))
    dd = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32))
    v = dd / dd.norm(dim=1, keepdim=True)
    z = torch.from_numpy(numpy.sort(rng.uniform(0, 1, (n, s)).astype(numpy.float32), axis=1))
    noise = torch.from_numpy(rng.standard_normal((n, s, 1)).astype(numpy.float32))
    g_sigma = torch.from_numpy(rng.standard_normal((n, s, 1)).astype(numpy.float32))
    g_rgb = torch.from_numpy(rng.standard_normal((n, s, 3)).astype(numpy.float32))
    return cfg, sd, (o, dd, v, z, noise), (g_sigma, g_rgb)


@pytest.mark.parametrize('layout', ['main', 'ptsaug', 'viewsaug'])
@pytest.mark.parametrize('size', [(8, 256, 128), (4, 128, 64)])
def test_f16_mlp_against_oracle(layout, size):
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case(layout, size)
    o, dd, v, z, noise = inputs
    params = {k: torch.from_numpy(v_).clone().requires_grad_(True) for k, v_ in sd.items()}
    ref = oracle.run_mlp(params, '', cfg, oracle.ray_points(o, dd, z), v, None, noise)
    ((ref['sigma'] * g_sigma).sum() + (ref['rgb'] * g_rgb).sum()).backward()
    plist = abi_param_list({k: torch.from_numpy(v_).to(DEV) for k, v_ in sd.items()})
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    dev = [t.to(DEV) for t in inputs]
    sigma_eval, rgb_eval = mlp.forward(*dev, F16)
    sigma, rgb, saved = mlp.forward_train(*dev, F16)
    # the storing forward and the inference forward are the same arithmetic; for the main 8x256 layout inference runs on the
    # 16x16x32 MFMA (mlp_forward_m16.hip), whose fp32 accumulation order differs: equal within this mode's own tolerance there
    if layout == 'main' and size == (8, 256, 128):
        assert util.rel_linf(sigma, sigma_eval) < 5e-3 and util.linf(rgb, rgb_eval) < 2e-4
        assert util.rel_linf(sigma_eval, ref['sigma']) < 5e-3 and util.linf(rgb_eval, ref['rgb']) < 2e-4
    else:
        assert torch.equal(sigma, sigma_eval) and torch.equal(rgb, rgb_eval)
    assert util.rel_linf(sigma, ref['sigma']) < 5e-3 and util.linf(rgb, ref['rgb']) < 2e-4
    shapes = [tuple(p.shape) for p in plist]
    grads = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, F16)
    again = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, F16)
    names = [k for k in abi_param_list({k: k for k in sd})]
    bad = {}
    for name, got, twice in zip(names, grads, again):
        assert torch.equal(got, twice), name                          # fixed-order reductions
        assert got.shape == params[name].grad.shape
        if rel_l2(got, params[name].grad) > 0.15:
            bad[name] = rel_l2(got, params[name].grad)
    assert not bad, bad


@pytest.mark.parametrize('precision', ['f16x3', 'f16'])
def test_backward_is_linear_in_the_loss_scale(precision):
    """Loss gradients reach 1e-10 in real training (means over thousands of rays); every product of the fp16 backward is
    renormalised by powers of two -- per sample in the chain, per dY region in the weight gradients, the head rows
    included -- so scaling the upstream gradient by 2^-30 must scale every parameter gradient by exactly 2^-30."""
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case('main', (8, 256, 128))
    plist = abi_param_list({k: torch.from_numpy(v_).to(DEV) for k, v_ in sd.items()})
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    prec = ops.PRECISIONS[precision]
    sigma, rgb, saved = mlp.forward_train(*[t.to(DEV) for t in inputs], prec)
    shapes = [tuple(p.shape) for p in plist]
    ref = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, prec)
    k = 2.0 ** -30
    small = mlp.backward(saved, sigma, rgb, (g_sigma * k).to(DEV), (g_rgb * k).to(DEV), shapes, prec)
    for a, b in zip(ref, small):
        assert torch.equal(a * k, b)


def synthetic_model(cfg, precision):
    model = get_model(synth.with_overrides(cfg, hip_precision=precision), None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    return model.to(DEV)


def test_f16_render_close_to_fp32():
    cfg = synth.make_configs('headline')
    cam = synth.camera('fern', 0)
    batch = harness.frame_batch(cam, True, DEV, 95000, 2048)
    with torch.no_grad():
        ref = synthetic_model(cfg, 'fp32').eval()(batch)
        got = synthetic_model(cfg, 'f16').eval()(batch)
    for k in ('rgb_coarse', 'rgb_fine'):
        assert util.linf(got[k], ref[k]) < 1e-3, k
    for k in ('depth_ndc_coarse', 'depth_ndc_fine'):
        assert util.linf(got[k], ref[k]) < 5e-3, k     # NDC depth range is [0, 1]


def test_f16_training_batch_close_to_fp32():
    """One reference-shaped training batch (four MLPs, nine losses): every loss value within 0.5 % of the fp32 path's, the
    accumulated parameter gradients within 5 % relative L2 per tensor, bit-reproducible."""
    def run(precision):
        cfg = synth.training_configs(precision, num_rays=1024, num_sparse=256)
        cfg['sub_batch_size'] = 1280
        cfg['losses'] = synth.loss_configs(iter_weighted=False)
        model = synthetic_model(cfg, precision).train()
        batch = BatchAssembler(cfg, synth.training_scene(0, 3, 96, 128, sparse_fraction=0.02), DEV).get_next_batch(0)
        losses = LossComputer(cfg)
        out = model(batch)
        terms = losses.compute_losses(batch, out)
        terms['TotalLoss'].backward()
        values = {k: float((v['loss_value'] if isinstance(v, dict) else v).detach()) for k, v in terms.items()}
        return values, {n: p.grad.clone() for n, p in model.named_parameters()}

    ref_loss, ref_grads = run('fp32')
    got_loss, got_grads = run('f16')
    again_loss, again_grads = run('f16')
    assert got_loss == again_loss and all(torch.equal(got_grads[k], again_grads[k]) for k in got_grads)
    for k, v in ref_loss.items():
        assert abs(got_loss[k] - v) <= 5e-3 * max(abs(v), 1e-6), (k, got_loss[k], v)
    bad = {k: rel_l2(got_grads[k], ref_grads[k]) for k in ref_grads if rel_l2(got_grads[k], ref_grads[k]) > 0.05}
    assert not bad, bad


def test_f16_training_run_tracks_fp32():
    """150 iterations of the whole training step on the synthetic plane scene in both precisions: same convergence."""
    def run(precision):
        cfg = synth.training_configs(precision, num_rays=1024, num_sparse=256)
        cfg['sub_batch_size'] = 1280
        cfg['losses'] = synth.loss_configs(iter_weighted=False)
        scene = synth.training_scene(0, 3, 96, 128, sparse_fraction=0.02)
        torch.manual_seed(0)
        model = get_model(cfg, None).to(DEV).train()
        batcher, losses = BatchAssembler(cfg, scene, DEV), LossComputer(cfg)
        opt = optim.Adam(list(model.parameters()), lr=cfg['optimizer']['lr_initial'], betas=(0.9, 0.999))
        decayer = get_lr_decayer(cfg)
        for it in range(150):
            for group in opt.param_groups:
                group['lr'] = decayer.get_updated_learning_rate(it)
            totals = harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it), cfg['sub_batch_size'])
        assert math.isfinite(float(totals['TotalLoss']))
        cam = {'resolution': scene['resolution'], 'intrinsic': scene['intrinsics'][0], 'pose': scene['poses'][0],
               'near': scene['near'], 'far': scene['far'], 'near_ndc': 0.0, 'far_ndc': 1.0}
        model.eval()
        rgb = harness.render_frame(model, cam, True, torch.device(DEV), keys=('rgb_fine',))['rgb_fine']
        target = torch.as_tensor(scene['images'][0]).reshape(-1, 3).to(DEV)
        return -10 * math.log10(max(float(torch.mean((rgb - target) ** 2)), 1e-12))

    ref, got = run('fp32'), run('f16')
    assert ref > 10.0 and abs(got - ref) < 1.0, (ref, got)


@pytest.mark.parametrize('case', list(range(10)))
def test_f16_random_shapes(case):
    """Ragged shapes (single rays / samples, counts off the 32- and 128-sample tiles) and depths 1, 2, 4, 6, 8 in all three
    weight layouts: forward against the oracle, gradients against the fp32 kernels on the same inputs."""
    rng = numpy.random.RandomState(3000 + case)
    layout = ['main', 'ptsaug', 'viewsaug'][case % 3]
    depth, width, vwidth = [(8, 256, 128), (4, 128, 64), (2, 128, 64), (6, 256, 128), (1, 256, 128)][case % 5]
    cfg = synth.mlp_config(64, depth=depth, width=width, views_width=vwidth, **LAYOUTS[layout])
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 300 + case, float(rng.choice([1.0, 30.0])), float(rng.uniform(-2, 2)))
    n, s = int(rng.randint(1, 90)), int(rng.choice([1, 2, 31, 33, 64, 127, 129, 192]))
    o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32))
    d = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32))
    v = d / d.norm(dim=1, keepdim=True)
    z = torch.from_numpy(numpy.sort(rng.uniform(0, 1, (n, s)).astype(numpy.float32), axis=1))
    params = {k: torch.from_numpy(a) for k, a in sd.items()}
    ref = oracle.run_mlp(params, '', cfg, oracle.ray_points(o, d, z), v, None, None)
    plist = abi_param_list({k: a.to(DEV) for k, a in params.items()})
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    dev = (o.to(DEV), d.to(DEV), v.to(DEV), z.to(DEV), None)
    g_sigma = torch.from_numpy(rng.standard_normal((n, s, 1)).astype(numpy.float32)).to(DEV)
    g_rgb = torch.from_numpy(rng.standard_normal((n, s, 3)).astype(numpy.float32)).to(DEV)
    shapes = [tuple(p.shape) for p in plist]
    sigma, rgb, saved = mlp.forward_train(*dev, F16)
    scale = max(1.0, float(ref['sigma'].abs().max()))
    assert util.linf(sigma, ref['sigma']) <= 5e-3 * scale and util.linf(rgb, ref['rgb']) <= 1e-3, (layout, depth, width, n, s)
    got = mlp.backward(saved, sigma, rgb, g_sigma, g_rgb, shapes, F16)
    sigma32, rgb32, saved32 = mlp.forward_train(*dev, ops.PRECISION_FP32)
    want = mlp.backward(saved32, sigma32, rgb32, g_sigma, g_rgb, shapes, ops.PRECISION_FP32)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.isfinite(a).all()
        assert rel_l2(a, b) < 0.2 or float(b.abs().max()) == 0.0, (i, layout, depth, width, n, s, rel_l2(a, b))


def test_both_inference_kernels_agree():
    """Rendering with the fp16 modes runs on the 16x16x32 MFMA layout (mlp_forward_m16.hip); training's storing forward is
    built on the 32x32x16 layout (mlp_forward_f16.hip).  Same weights, same arithmetic per product, different fp32
    accumulation order: f16x3 agrees to 5e-5 on the density (gain-300 head) and 5e-6 on the colour, the 16-bit mode within
    its own tolerance.  (Which kernel renders is decided by the call -- snerf_mlp_forward vs snerf_mlp_forward_train -- not by
    an environment switch.)"""
    cfg = synth.mlp_config(64)
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 5, 300.0, -5.0)
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(abi_param_list({k: torch.from_numpy(v).to(DEV) for k, v in sd.items()}))
    rng = numpy.random.RandomState(3)
    for n, s in ((1, 1), (7, 37), (64, 192)):
        o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32)).to(DEV)
        d = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32)).to(DEV)
        v = d / d.norm(dim=1, keepdim=True)
        z = torch.from_numpy(numpy.sort(rng.uniform(0, 1, (n, s)).astype(numpy.float32), axis=1)).to(DEV)
        for prec in (ops.PRECISION_F16X3, ops.PRECISION_F16):
            sigma, rgb = mlp.forward(o, d, v, z, None, precision=prec)
            sigma_t, rgb_t, _ = mlp.forward_train(o, d, v, z, None, precision=prec)
            f16_mode = prec == ops.PRECISION_F16
            assert n * s <= 4 or not torch.equal(sigma, sigma_t), (n, s, prec)       # (two kernels, not one)
            # (the synthetic density head has gain 300: few-ulp differences of the 256-term dot products are amplified)
            assert util.rel_linf(sigma, sigma_t) < (5e-3 if f16_mode else 5e-5), (n, s, prec)
            assert util.linf(rgb, rgb_t) < (2e-4 if f16_mode else 5e-6), (n, s, prec)


@pytest.mark.parametrize('precision', ['f16x3', 'f16'])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_leaving_the_fp16_range_is_detected_not_silent(precision, mode):
    """both fp16 modes hold hidden activations as fp16 numbers; a hidden unit past 65504 used to come out as a
    non-finite -- or, masked by a later ReLU, a finite but wrong -- sample without any signal.  Now every fp16-mode forward
    kernel watches the operands it converts and raises the device's range flag: visible through ops.range_status() after a
    synchronisation, and as Fp16RangeError from the NEXT fp16-mode call (launches are asynchronous).  The fp32 mode renders the
    same model without complaint."""
    cfg = synth.with_overrides(synth.make_configs('config2'), hip_precision=precision)
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()}
    model.load_state_dict(sd)
    model = model.to(DEV).train(mode == 'train')
    cam = synth.camera('fern', 0)
    batch = harness.frame_batch(cam, True, DEV, 200000, 64)
    ops.range_status(clear=True)
    with torch.no_grad():
        good = model(batch)
    torch.cuda.synchronize()
    assert ops.range_status() == 0 and torch.isfinite(good['rgb_fine']).all()
    # one hidden unit of the third trunk layer driven past the fp16 maximum (its bias: 1e5 > 65504)
    with torch.no_grad():
        model.coarse_model.pts_linears[2].bias[17] = 1.0e5
        out = model(batch)
    torch.cuda.synchronize()
    assert ops.range_status() & ops.RANGE_ACTIVATION
    with pytest.raises(ops.Fp16RangeError, match='fp16 range'):
        with torch.no_grad():
            model(batch)
    assert ops.range_status() == 0                      # reported once; the refused call enqueued nothing
    # the same weights in the fp32 mode: finite, no flag
    ref_model = get_model(synth.make_configs('config2'), None)
    ref_model.load_state_dict(model.state_dict())
    ref_model = ref_model.to(DEV).train(mode == 'train')
    with torch.no_grad():
        ref = ref_model(batch)
    torch.cuda.synchronize()
    assert ops.range_status() == 0 and all(torch.isfinite(v).all() for v in ref.values())
    # the flagged launch's outputs are wrong where they are finite at all -- which is why they must not be used
    assert not torch.isfinite(out['rgb_coarse']).all() or float((out['rgb_coarse'] - ref['rgb_coarse']).abs().max()) > 1e-3
    # a weight outside the range is caught when the stream is packed
    # (by the pack kernel of the same call: depending on how quickly it finishes, the flag is reported by that very call's
    # forward or stays set for the next one)
    with torch.no_grad():
        model.coarse_model.pts_linears[2].bias[17] = 0.0
        model.coarse_model.pts_linears[3].weight[5, 9] = -7.0e4
        try:
            model(batch)
            torch.cuda.synchronize()
            caught = bool(ops.range_status(clear=True) & ops.RANGE_WEIGHT)
        except ops.Fp16RangeError as error:
            caught = 'a weight' in str(error)
            torch.cuda.synchronize()
            ops.range_status(clear=True)
    assert caught
    with torch.no_grad():
        model.coarse_model.pts_linears[3].weight[5, 9] = 0.01
        again = model(batch)
    torch.cuda.synchronize()
    assert ops.range_status() == 0 and torch.isfinite(again['rgb_fine']).all()


def test_an_fp32_mode_model_with_large_weights_does_not_trip_another_models_fp16_call():
    """the weight-range bit used to be set by ``snerf_mlp_pack`` in the per-device flag for ANY model, so an fp32-mode
    model holding |w| > 65504 (legitimate there) made the next fp16-mode call of a DIFFERENT model fail with a spurious
    Fp16RangeError.  The bit now lives in the packed buffer and is reported only by an fp16-mode kernel that consumes it."""
    cam = synth.camera('fern', 0)
    batch = harness.frame_batch(cam, True, DEV, 200000, 64)

    def make(precision):
        cfg = synth.with_overrides(synth.make_configs('config2'), hip_precision=precision)
        model = get_model(cfg, None)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
        return model.to(DEV).eval()

    big, small = make('fp32'), make('f16')
    ops.range_status(clear=True)
    with torch.no_grad():
        big.coarse_model.pts_linears[3].weight[5, 9] = -7.0e4          # fine in fp32
        a = big(batch)
        torch.cuda.synchronize()
        assert ops.range_status() == 0 and torch.isfinite(a['rgb_fine']).all()
        b = small(batch)                                                # another model, fp16 mode: must not be refused
        torch.cuda.synchronize()
        assert ops.range_status() == 0 and torch.isfinite(b['rgb_fine']).all()


def test_overflow_of_a_view_independent_mlps_last_activations_is_detected():
    """in the 16-bit training forward the last trunk activations of a view-independent MLP (the views-augmentation
    model: no views layer) are converted to fp16 and saved for the head weight gradients only -- no later product would turn
    non-finite -- so they are watched directly."""
    cfg = synth.training_configs('f16', num_rays=96, num_sparse=32)
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 9, 200.0, 8.0).items()})
    model = model.to(DEV).train()
    assert not cfg['model']['views_augmentation']['coarse_mlp']['use_view_dirs']
    batch = harness.frame_batch(synth.camera('fern', 0), True, DEV, 200000, 64)
    ops.range_status(clear=True)
    model(batch)                  # (with autograd on: the STORING forward is the one that converts these activations)
    torch.cuda.synchronize()
    assert ops.range_status() == 0
    last = cfg['model']['views_augmentation']['coarse_mlp']['points_net_depth'] - 1
    with torch.no_grad():
        model.views_aug_coarse_model.pts_linears[last].bias[3] = 1.0e5
    model(batch)
    torch.cuda.synchronize()
    assert ops.range_status(clear=True) & ops.RANGE_ACTIVATION


def test_graphed_training_step_reports_a_range_violation():
    """A graph replay bypasses the entry points that report the flag: GraphedTrainStep asks after every replay."""
    cfg = synth.training_configs('f16', num_rays=192, num_sparse=64)
    cfg['losses'] = synth.loss_configs(iter_weighted=False)
    scene = synth.training_scene(0, 3, 48, 64, sparse_fraction=0.05)
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 9, 200.0, 8.0).items()})
    model = model.to(DEV).train()
    batcher = BatchAssembler(cfg, scene, DEV)
    step = harness.GraphedTrainStep(model, LossComputer(cfg), batcher.get_next_batch(0))
    ops.range_status(clear=True)
    step(batcher.get_next_batch(0))
    torch.cuda.synchronize()
    assert ops.range_status() == 0
    with torch.no_grad():
        model.fine_model.pts_linears[1].bias[3] = 2.0e5
    step(batcher.get_next_batch(1))
    torch.cuda.synchronize()
    with pytest.raises(ops.Fp16RangeError):
        step(batcher.get_next_batch(2))
    ops.range_status(clear=True)


# ---------------------------------------------------------------- 'f16s8': the 16-bit mode with fp8 saved trunk activations
S8 = ops.PRECISIONS['f16s8']
FP8_MODES = {'f16s8': (ops.PRECISIONS['f16s8'], ops.PRECISIONS['f16']), 'bf16s8': (ops.PRECISIONS['bf16s8'], ops.PRECISIONS['bf16'])}


@pytest.mark.parametrize('mode', ['f16s8', 'bf16s8'])
@pytest.mark.parametrize('layout', ['main', 'ptsaug', 'viewsaug'])
@pytest.mark.parametrize('size', [(8, 256, 128), (4, 128, 64), (6, 256, 128), (2, 128, 64), (1, 256, 128)])
def test_fp8_saved_activations_change_only_the_weight_gradients(layout, size, mode):
    """SNERF_PRECISION_F16S8 keeps h_1 .. h_D-1 as fp8 e4m3 tiles: rendering and the training forward give the fp16 mode's bits,
    and of the backward only the weight gradients that contract over those tensors (trunk layers 1 .. D-1 of a 256-wide MLP) may
    differ -- by the rounding of a 4-bit significand (3.6 % rms per element), which averages out over the samples of the
    contraction: on this 315-sample case a handful of samples dominate every gradient, so it barely averages (<= 8 % relative
    L2, observed 2.3-4.8 %); on a 1280-row training batch the worst tensor is 1 % from the fp16 mode's (below).  Every other
    gradient tensor is bit-identical to the fp16 mode's, and a 128-wide MLP is the fp16 mode altogether.  ('bf16s8': the same
    statements about the bf16 mode.)"""
    S8, F16 = FP8_MODES[mode]
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case(layout, size)
    plist = abi_param_list({k: torch.from_numpy(v_).to(DEV) for k, v_ in sd.items()})
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    dev = [t.to(DEV) for t in inputs]
    shapes = [tuple(p.shape) for p in plist]
    sigma, rgb, saved = mlp.forward_train(*dev, F16)
    ref = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, F16)
    sigma8, rgb8, saved8 = mlp.forward_train(*dev, S8)
    assert torch.equal(sigma8, sigma) and torch.equal(rgb8, rgb)
    sigma_eval, rgb_eval = mlp.forward(*dev, S8)
    sigma_f16, rgb_f16 = mlp.forward(*dev, F16)
    assert torch.equal(sigma_eval, sigma_f16) and torch.equal(rgb_eval, rgb_f16)
    got = mlp.backward(saved8, sigma8, rgb8, g_sigma.to(DEV), g_rgb.to(DEV), shapes, S8)
    again = mlp.backward(saved8, sigma8, rgb8, g_sigma.to(DEV), g_rgb.to(DEV), shapes, S8)
    names = [k for k in abi_param_list({k: k for k in sd})]
    depth = size[0]
    worst = 0.0
    for name, a, b, c in zip(names, got, ref, again):
        assert torch.equal(a, c), name
        through_fp8 = size[1] == 256 and any(name == f'pts_linears.{l}.weight' for l in range(1, depth))
        if through_fp8:
            worst = max(worst, rel_l2(a, b))
        else:
            assert torch.equal(a, b), name
    util.observe(f'{mode}/{layout}/{depth}x{size[1]}', f'weight gradients through fp8 activations vs the 16-bit mode: rel L2 {worst:.4f} [0.08]')
    assert worst < 0.08


@pytest.mark.parametrize('mode', ['f16s8', 'bf16s8'])
def test_fp8_saved_activations_clamp_instead_of_overflowing(mode):
    """e4m3 ends at 448 and the hardware conversion returns NaN above it: activations beyond are clamped in the weight-gradient
    operand (the forward itself is the fp16 mode's, range 65504) -- finite gradients, equal to the fp16 mode's wherever the
    clamped unit is not the operand."""
    S8, F16 = FP8_MODES[mode]
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case('main', (8, 256, 128))
    sd = dict(sd)
    sd['pts_linears.2.bias'] = sd['pts_linears.2.bias'].copy()
    sd['pts_linears.2.bias'][17] = 3000.0                 # h_3[:, 17] ~ 3000 > 448
    plist = abi_param_list({k: torch.from_numpy(v_).to(DEV) for k, v_ in sd.items()})
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    dev = [t.to(DEV) for t in inputs]
    shapes = [tuple(p.shape) for p in plist]
    sigma, rgb, saved = mlp.forward_train(*dev, S8)
    grads = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, S8)
    torch.cuda.synchronize()
    assert ops.range_status(clear=True) == 0 and all(torch.isfinite(g).all() for g in grads)
    ref = mlp.backward(*mlp.forward_train(*dev, F16)[2:3], sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, F16)
    names = [k for k in abi_param_list({k: k for k in sd})]
    w3 = names.index('pts_linears.3.weight')
    big = ref[w3][:, 17].abs() > 1e-3 * ref[w3][:, 17].abs().max()      # dW_3[:, 17] = sum dY_3 . h_3[:, 17]: clamped 3000 -> 448
    ratio = grads[w3][:, 17][big] / ref[w3][:, 17][big]
    assert big.any() and float((ratio - 448.0 / 3000.0).abs().max()) < 0.02, ratio
    others = torch.ones(256, dtype=torch.bool, device=DEV)
    others[17] = False
    assert rel_l2(grads[w3][:, others], ref[w3][:, others]) < 0.08


@pytest.mark.parametrize('mode', ['f16s8', 'bf16s8'])
def test_f16s8_training_batch_close_to_fp32(mode):
    def run(precision):
        cfg = synth.training_configs(precision, num_rays=1024, num_sparse=256)
        cfg['sub_batch_size'] = 1280
        cfg['losses'] = synth.loss_configs(iter_weighted=False)
        model = synthetic_model(cfg, precision).train()
        batch = BatchAssembler(cfg, synth.training_scene(0, 3, 96, 128, sparse_fraction=0.02), DEV).get_next_batch(0)
        losses = LossComputer(cfg)
        terms = losses.compute_losses(batch, model(batch))
        terms['TotalLoss'].backward()
        values = {k: float((v['loss_value'] if isinstance(v, dict) else v).detach()) for k, v in terms.items()}
        return values, {n: p.grad.clone() for n, p in model.named_parameters()}

    ref_loss, ref_grads = run('fp32')
    f16_loss, f16_grads = run(mode[:-2])
    got_loss, got_grads = run(mode)
    assert got_loss == f16_loss                                            # the forward is the 16-bit mode's
    worst32 = max(rel_l2(got_grads[k], ref_grads[k]) for k in ref_grads)
    worst16 = max(rel_l2(got_grads[k], f16_grads[k]) for k in ref_grads)
    bar32 = 0.05 if mode == 'f16s8' else 0.20          # (the bf16 mode's own distance from fp32: tests/test_gpu_bf16.py)
    util.observe(f'{mode}/training_batch', f'parameter gradients vs fp32: worst rel L2 {worst32:.4f} [{bar32}]; vs the 16-bit mode: {worst16:.4f} [0.02]')
    assert worst32 <= bar32 and worst16 <= 0.02


# ---------------------------------------------------------------- snerf_mlp_pack_for: only the operand formats one precision reads
@pytest.mark.parametrize('layout', ['main', 'ptsaug', 'viewsaug'])
@pytest.mark.parametrize('size', [(8, 256, 128), (4, 128, 64)])
@pytest.mark.parametrize('precision', ['fp32', 'f16x3', 'f16', 'bf16', 'f16s8', 'bf16s8'])
def test_selective_pack_gives_the_same_bits_as_the_full_pack(precision, size, layout):
    """snerf_mlp_pack writes every operand format of the weights, snerf_mlp_pack_for(precision, training) only what that
    precision reads in that mode: rendering (training = 0) and the storing forward + backward (training = 1) from the selective
    streams are bit-identical to the same calls on the full stream."""
    prec = ops.PRECISIONS[precision]
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case(layout, size)
    plist = abi_param_list({k: torch.from_numpy(v_).to(DEV) for k, v_ in sd.items()})
    shapes = [tuple(p.shape) for p in plist]
    dev = [t.to(DEV) for t in inputs]
    full = ops.PackedMlp(cfg, DEV)
    full.pack(plist)
    ref_eval = full.forward(*dev, prec)
    sigma, rgb, saved = full.forward_train(*dev, prec)
    ref_grads = full.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, prec)

    lean = ops.PackedMlp(cfg, DEV)
    lean.pack(plist, prec, training=False)
    got_eval = lean.forward(*dev, prec)
    assert all(torch.equal(a, b) for a, b in zip(got_eval, ref_eval))
    lean.pack(plist, prec, training=True)
    sigma2, rgb2, saved2 = lean.forward_train(*dev, prec)
    assert torch.equal(sigma2, sigma) and torch.equal(rgb2, rgb)
    got_grads = lean.backward(saved2, sigma2, rgb2, g_sigma.to(DEV), g_rgb.to(DEV), shapes, prec)
    assert all(torch.equal(a, b) for a, b in zip(got_grads, ref_grads))
    torch.cuda.synchronize()
    assert ops.range_status(clear=True) == 0


@pytest.mark.parametrize('packed_for,used_at,training_call', [
    (('f16', True), ('f16', False), False),          # packed for training, rendered from (the m16 stream is zero-filled)
    (('bf16', True), ('bf16', False), False),
    (('fp32', False), ('f16', False), False),        # packed for fp32, used at f16
    (('f16', False), ('fp32', False), False),        # ... and the reverse
    (('f16', False), ('bf16', False), False),
    (('fp32', False), ('f16x3', True), True),        # the storing forward of another precision
])
def test_a_stream_packed_for_another_precision_or_mode_is_refused(packed_for, used_at, training_call):
    """snerf_mlp_pack_for zero-fills the operand formats it does not write, and a consumer of such a format
    used to return bias-only output with SNERF_OK.  The library now remembers what every packed buffer holds and each entry
    point that reads a weight stream fails with SNERF_E_INVALID -- before enqueuing anything -- when the buffer's last pack did
    not write it; after a matching re-pack (or a full snerf_mlp_pack) the same call succeeds."""
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case('main', (8, 256, 128))
    plist = abi_param_list({k: torch.from_numpy(v_).to(DEV) for k, v_ in sd.items()})
    dev = [t.to(DEV) for t in inputs]
    stream = ops.PackedMlp(cfg, DEV)
    stream.pack(plist, ops.PRECISIONS[packed_for[0]], training=packed_for[1])
    prec = ops.PRECISIONS[used_at[0]]
    call = (lambda: stream.forward_train(*dev, prec)) if training_call else (lambda: stream.forward(*dev, prec))
    with pytest.raises(RuntimeError, match='holds the operand formats'):
        call()
    # the buffer's own (precision, mode) still works, and so does the refused call after a matching pack / a full pack
    own = ops.PRECISIONS[packed_for[0]]
    (stream.forward_train if packed_for[1] else stream.forward)(*dev, own)
    stream.pack(plist, prec, training=used_at[1])
    first = call()
    stream.pack(plist)
    second = call()
    assert torch.equal(first[0], second[0]) and torch.equal(first[1], second[1]) and float(first[0].abs().max()) > 0
    if packed_for == ('fp32', False):     # the backward chain reads the training layout's transposed stream
        stream.pack(plist, ops.PRECISIONS['f16'], training=True)
        sigma, rgb, saved = stream.forward_train(*dev, ops.PRECISIONS['f16'])
        stream.pack(plist, ops.PRECISIONS['fp32'], training=True)
        with pytest.raises(RuntimeError, match='holds the operand formats'):
            stream.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), [tuple(p.shape) for p in plist], ops.PRECISIONS['f16'])
    torch.cuda.synchronize()
    ops.range_status(clear=True)


def test_model_repacks_when_a_render_follows_training_steps():
    """The model keys its packed streams by (precision, keeps activations): an evaluation render between training iterations
    reads the rendering layout, which the training pack does not write -- it must trigger a re-pack, not read zeros."""
    cfg = synth.training_configs('f16', num_rays=192, num_sparse=64)
    cfg['losses'] = synth.loss_configs(iter_weighted=False)
    model = synthetic_model(cfg, 'f16').train()
    scene = synth.training_scene(0, 3, 48, 64, sparse_fraction=0.05)
    batcher = BatchAssembler(cfg, scene, DEV)
    losses = LossComputer(cfg)
    batch = batcher.get_next_batch(0)
    with torch.no_grad():
        before = {k: v.clone() for k, v in model(batch).items() if isinstance(v, torch.Tensor)}
    terms = losses.compute_losses(batch, model(batch))          # a training pass: packs the training layout only
    terms['TotalLoss'].backward()
    with torch.no_grad():
        after = {k: v for k, v in model(batch).items() if isinstance(v, torch.Tensor)}
    assert before and all(torch.equal(before[k], after[k]) for k in before), 'a render after a training pass changed'
    assert any(float(v.abs().max()) > 0 for v in after.values())

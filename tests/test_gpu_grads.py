"""Backward-pass parity on a real MI355X: K6 (compositing backward) and K7 (MLP parameter gradients) against autograd
through the oracle, and the whole training-mode model against the reference's own gradients (golden G7).

Bound: every gradient tensor within 1e-3 of its own largest magnitude (L-infinity relative to max|ref|); these are
fp32 sums over 1e4..1e5 samples evaluated in a different order than the CPU BLAS, so ~1e-5 is what is observed."""
import numpy
import pytest
import torch

from oracle import nerf_oracle as oracle
from simplenerf_amd import ops, synth
from simplenerf_amd.models.ModelFactory import get_model
from tests import util
from tests.test_gpu_kernels import LAYOUTS, abi_param_list

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GRAD_TOL = 1e-3


def rel_l2(got, ref):
    ref = ref.detach().cpu().double()
    got = got.detach().cpu().double()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


# The f16x3 kernels reproduce every activation to ~1e-6 relative instead of ~1e-7.  d relu/dx is discontinuous, so the few
# (sample, unit) pairs whose pre-activation lies within that distance of 0 get a different mask than the CPU evaluation.
# With the spiky gradients of a dense field (a handful of surface samples per ray carry the loss) one flipped pair on
# such a sample moves entries of the early layers' gradients by several percent of the tensor's largest entry.  The
# arithmetic itself is gated separately and tightly: test_f16x3_chain_arithmetic_with_identical_masks (<= 1e-4) and the
# forward's activations (<= 1e-5 of the fp32 kernel's).  Hence the loose end-to-end bounds for f16x3:
F16_TOL_MAX, F16_TOL_L2 = 1e-1, 6e-2


def rel_to_max(got, ref):
    ref = ref.detach().cpu().double()
    got = got.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


# ---------------------------------------------------------------- K6
@pytest.mark.parametrize('ndc,white,s', [(False, False, 64), (True, False, 64), (True, False, 192), (False, True, 130),
                                         (True, True, 256), (False, False, 7)])
def test_composite_backward_matches_autograd(ndc, white, s):
    rng = numpy.random.RandomState(s + 17 * ndc)
    n = 23
    if ndc:
        g = util.load('composite.npz')
        rays_o, rays_d = torch.from_numpy(g['ndc_s64_rays_o'][:n]), torch.from_numpy(g['ndc_s64_rays_d'][:n])
        march = torch.from_numpy(g['ndc_s64_rays_d_ndc'][:n])
        z = torch.from_numpy(numpy.sort(rng.uniform(0, 1, (n, s)).astype(numpy.float32), axis=1))
        z[: n // 2] = torch.linspace(0, 1, s)
    else:
        rays_o = rays_d = None
        march = torch.from_numpy(rng.standard_normal((n, 3)).astype(numpy.float32))
        z = torch.from_numpy(numpy.sort(rng.uniform(2, 6, (n, s)).astype(numpy.float32), axis=1))
    sigma = torch.from_numpy(rng.gamma(0.5, 8.0, (n, s)).astype(numpy.float32))
    sigma[rng.uniform(size=sigma.shape) < 0.3] = 0
    sigma[:2] = 0
    sigma[2:4, 5 % s] = 1e4  # an opaque sample: 1 - alpha + 1e-10 = 1e-10
    rgb = torch.from_numpy(rng.uniform(0, 1, (n, s, 3)).astype(numpy.float32))
    grads = {k: torch.from_numpy(rng.standard_normal(shape).astype(numpy.float32))
             for k, shape in (('rgb', (n, 3)), ('acc', (n,)), ('depth', (n,)), ('depth_ndc', (n,)))}
    sg, cg = sigma.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    out = oracle.composite(sg, cg, z, march, ndc, white, rays_o, rays_d)
    loss = sum((out[k] * grads[k]).sum() for k in grads if k in out)
    loss.backward()
    d = lambda t: None if t is None else t.to(DEV)
    d_sigma, d_rgb = ops.composite_backward(d(sigma), d(rgb), d(z), d(march), ndc, white, d(rays_o), d(rays_d),
                                            d(grads['rgb']), d(grads['acc']), d(grads['depth']),
                                            d(grads['depth_ndc']) if ndc else None)
    assert rel_to_max(d_rgb, cg.grad) < 1e-5
    assert rel_to_max(d_sigma, sg.grad) < 1e-4


# ---------------------------------------------------------------- K7
@pytest.mark.parametrize('layout', ['main', 'ptsaug', 'viewsaug'])
@pytest.mark.parametrize('size', [(8, 256, 128), (4, 128, 64)])
@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
def test_mlp_backward_matches_autograd(layout, size, precision):
    depth, width, vwidth = size
    cfg = synth.mlp_config(64, depth=depth, width=width, views_width=vwidth, **LAYOUTS[layout])
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 31, 50.0, 1.0)
    rng = numpy.random.RandomState(depth)
    n, s = 7, 45  # 315 samples: not a multiple of the 128-sample workgroup tile
    o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32))
    dd = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32))
    v = dd / dd.norm(dim=1, keepdim=True)
    z = torch.from_numpy(numpy.sort(rng.uniform(0, 1, (n, s)).astype(numpy.float32), axis=1))
    noise = torch.from_numpy(rng.standard_normal((n, s, 1)).astype(numpy.float32))
    g_sigma = torch.from_numpy(rng.standard_normal((n, s, 1)).astype(numpy.float32))
    g_rgb = torch.from_numpy(rng.standard_normal((n, s, 3)).astype(numpy.float32))
    params = {k: torch.from_numpy(v_).clone().requires_grad_(True) for k, v_ in sd.items()}
    ref = oracle.run_mlp(params, '', cfg, oracle.ray_points(o, dd, z), v, None, noise)
    ((ref['sigma'] * g_sigma).sum() + (ref['rgb'] * g_rgb).sum()).backward()

    dev_params = {k: torch.from_numpy(v_).to(DEV) for k, v_ in sd.items()}
    plist = abi_param_list(dev_params)
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    prec = ops.PRECISIONS[precision]
    sigma, rgb, saved = mlp.forward_train(o.to(DEV), dd.to(DEV), v.to(DEV), z.to(DEV), noise.to(DEV), prec)
    assert util.rel_linf(sigma, ref['sigma']) < 1e-5 and util.linf(rgb, ref['rgb']) < 1e-5
    grads = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), [tuple(p.shape) for p in plist], prec)
    names = [k for k in abi_param_list({k: k for k in sd})]
    bad = {}
    for name, got in zip(names, grads):
        e_max, e_l2 = rel_to_max(got, params[name].grad), rel_l2(got, params[name].grad)
        ok = e_max < GRAD_TOL if precision == 'fp32' else (e_max < F16_TOL_MAX and e_l2 < F16_TOL_L2)
        if not ok:
            bad[name] = (e_max, e_l2)
    assert not bad, bad


def test_f16x3_chain_arithmetic_with_identical_masks():
    """The f16x3 backward chain against the fp32 chain on the SAME saved activations (identical ReLU masks), with
    gradients spanning many orders of magnitude between samples and scaled down to 1e-9: what remains is pure
    arithmetic (fp16 hi/lo split with per-sample power-of-two scaling) and must be fp32-grade.  Also: the f16x3
    training forward saves the same activations as the fp32 one."""
    cfg = synth.mlp_config(64)
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 31, 50.0, 1.0)
    plist = abi_param_list({k: torch.from_numpy(v).to(DEV) for k, v in sd.items()})
    shapes = [tuple(p.shape) for p in plist]
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    rng = numpy.random.RandomState(0)
    n, s = 64, 192
    o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32)).to(DEV)
    d = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32)).to(DEV)
    v = d / d.norm(dim=1, keepdim=True)
    z = torch.sort(torch.from_numpy(rng.uniform(0, 1, (n, s)).astype(numpy.float32)).to(DEV), 1)[0]
    sigma, rgb, saved = mlp.forward_train(o, d, v, z, None, ops.PRECISION_FP32)
    sigma16, rgb16, saved16 = mlp.forward_train(o, d, v, z, None, ops.PRECISION_F16X3)
    rows = saved.numel() // (((n * s + 127) // 128) * 4 * 32)
    a, b = saved.reshape(-1, rows, 32), saved16.reshape(-1, rows, 32)
    mask_rows = 2 * ((8 * 8 + 4 + 1) // 2)   # ReLU sign-bit words of the 64 trunk + 4 views tiles close every block's tile
    for r0, r1 in ((0, 63), (64, 91), (96, rows - mask_rows)):  # encodings, view encodings, every layer's activations
        assert float((a[:, r0:r1] - b[:, r0:r1]).abs().max()) <= 1e-5 * max(1.0, float(a[:, r0:r1].abs().max()))
    # the sign bits agree except where a pre-activation is within rounding of zero, and they are exactly the signs of
    # the saved (post-ReLU) activations: bit r of tile t, lane (j, half) <-> feature 32t' + (r&3) + 8(r>>2) + 4 half
    wa = a[:, rows - mask_rows:].contiguous().view(torch.int32).reshape(a.shape[0], -1, 64)
    wb = b[:, rows - mask_rows:].contiguous().view(torch.int32).reshape(a.shape[0], -1, 64)
    differing = (wa ^ wb).cpu().numpy().view(numpy.uint32)
    assert numpy.unpackbits(differing.view(numpy.uint8)).mean() < 1e-4
    h1 = a[:, 96:96 + 256].reshape(-1, 8, 32, 32)            # layer 0's output: (block, tile, feature in tile, sample)
    lanes = torch.arange(64, device=DEV)
    for t in (0, 5):
        word = wa[:, t // 2, :] >> (16 * (t % 2))
        for r in (0, 7, 15):
            f = (r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5)
            expect = h1[:, t, f, lanes & 31] > 0
            assert torch.equal(((word >> r) & 1).bool(), expect), (t, r)
    assert util.rel_linf(sigma16, sigma) < 1e-5 and util.linf(rgb16, rgb) < 1e-5
    for scale in (1.0, 1e-6, 1e-9):
        spread = numpy.exp(4 * rng.standard_normal((n, s, 1)))
        gs = torch.from_numpy((rng.standard_normal((n, s, 1)) * spread * scale).astype(numpy.float32)).to(DEV)
        gr = torch.from_numpy((rng.standard_normal((n, s, 3)) * spread * scale).astype(numpy.float32)).to(DEV)
        ref = mlp.backward(saved, sigma, rgb, gs, gr, shapes, ops.PRECISION_FP32)
        got = mlp.backward(saved, sigma, rgb, gs, gr, shapes, ops.PRECISION_F16X3)
        for g_, r_ in zip(got, ref):
            assert rel_to_max(g_, r_) < 1e-4 and rel_l2(g_, r_) < 1e-4, scale


# ---------------------------------------------------------------- whole model
# (config 1 has no fine pass: its fine depths cannot be the oracle's -- not a case, rather than a skipped one)
@pytest.mark.parametrize('kind,profile,fine_depths', [(k, p, f) for k, p in (('config3', 'consistent'), ('config2', 'consistent'),
                                                                             ('headline_world', 'dense'), ('config1', 'dense'))
                                                      for f in ('own', 'oracle') if not (k == 'config1' and f == 'oracle')])
@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
def test_model_gradients_match_reference(kind, profile, fine_depths, precision):
    """loss.backward() through the drop-in model vs (a) autograd through the oracle, every element, and (b) the
    reference's own gradients (strided sample in fixture G7).

    'oracle': the fine pass runs on the oracle's fine depths, so every parameter gradient must agree to GRAD_TOL.
    'own':    the model resamples itself; a fraction of a percent of its fine samples sit in other bins than the
              reference's, which perturbs the FINE model's gradients at the percent level, so those
              are bounded at 5e-2 while coarse and augmented models keep GRAD_TOL."""
    g = util.load(f'grads_{kind}_{profile}.npz')
    cfg = synth.with_overrides(synth.make_configs(kind), perturb=False, raw_noise_std=0.0)
    params = {k: v.clone().requires_grad_(True) for k, v in util.golden_params(cfg, g).items()}
    ref_out = oracle.render(params, cfg, util.golden_batch(g), training=True)
    util.grad_loss(ref_out).backward()

    model = get_model(synth.with_overrides(cfg, hip_precision=precision), None)
    model.load_state_dict(util.golden_params(cfg, g))
    model = model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
    if fine_depths == 'oracle':
        model.set_random_draws({'z_vals_fine': ref_out['z_vals_fine'].detach()})
    out = model(batch)
    loss = util.grad_loss(out)
    loss.backward()
    assert abs(float(loss.detach()) - float(g['loss'])) < 1e-4 * max(1.0, abs(float(g['loss'])))
    bad = {}
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        fine = name.startswith('fine_model.')
        tol = 5e-2 if (fine_depths == 'own' and fine) else (2 * GRAD_TOL if precision == 'fp32' else F16_TOL_MAX)
        # the fixture was produced on the build container's CPU; the reference's fine depths are not reproducible
        # across CPUs/BLAS builds either (same discontinuity), so its FINE-model gradients are only loosely comparable
        tol_ref = 5e-2 if fine else (2 * GRAD_TOL if precision == 'fp32' else F16_TOL_MAX)
        err = rel_to_max(p.grad, params[name].grad)
        sample = p.grad.reshape(-1)[::util.GRAD_SAMPLE_STRIDE].cpu().double().numpy()
        ref = g[f'gradsample_{name}'].astype(numpy.float64)
        err_ref = float(numpy.abs(sample - ref).max() / max(float(params[name].grad.abs().max()), 1e-30))
        l2_ok = precision == 'fp32' or (fine and fine_depths == 'own') or rel_l2(p.grad, params[name].grad) < F16_TOL_L2
        if not (err < tol and err_ref < tol_ref and l2_ok):
            bad[name] = (err, err_ref, tol, tol_ref, rel_l2(p.grad, params[name].grad))
    assert not bad, bad


def test_losses_on_outputs_without_gradient_path_fail_loudly():
    cfg = synth.make_configs('config1')
    model = get_model(cfg, None).to(DEV).train()
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.random_world_rays(16).items()}
    out = model(batch)
    assert out['rgb_coarse'].requires_grad and out['depth_coarse'].requires_grad
    assert not out['weights_coarse'].requires_grad and not out['depth_var_coarse'].requires_grad
    with pytest.raises(RuntimeError):
        out['depth_var_coarse'].sum().backward()


@pytest.mark.parametrize('precision', ['fp32', 'f16x3', 'f16'])
def test_full_size_training_batch_properties(precision):
    """BASELINE config 5 at full size (4096 rays, four MLPs, 64 + 192 samples) through size-independent properties:
    the gradient of a sum-type loss over the whole batch equals the accumulated gradients of its two 2048-ray halves
    (the reference's sub-batching), doubles when the loss doubles, and is bit-reproducible run to run.  The 16-bit mode
    ('f16', the mode BASELINE config 5 names) quantises layer gradients with one power-of-two scale per region of the
    launch, so halves and whole agree to its own tolerance (2 %) instead of 1e-4; doubling is exact in every mode."""
    from simplenerf_amd import harness
    cfg = synth.with_overrides(synth.make_configs('config3'), perturb=False, raw_noise_std=0.0, hip_precision=precision)
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    model = model.to(DEV).train()
    batch = harness.frame_batch(synth.camera('fern', 0), True, DEV, 300000, 4096)
    gen = torch.Generator(device=DEV).manual_seed(0)
    target = torch.rand((4096, 3), device=DEV, generator=gen)

    def grads(rows, scale):
        model.zero_grad(set_to_none=True)
        for lo, hi in rows:
            out = model({k: v[lo:hi] for k, v in batch.items()})
            loss = sum(((out[k] - target[lo:hi]) ** 2).sum() for k in out if k.endswith(('rgb_coarse', 'rgb_fine')) and 'raw' not in k)
            loss = loss + 0.1 * sum((out[k] ** 2).sum() for k in out if k.endswith(('depth_ndc_coarse', 'depth_ndc_fine')))
            (scale * loss).backward()
        return [p.grad.clone() for p in model.parameters()]

    whole = grads([(0, 4096)], 1.0)
    again = grads([(0, 4096)], 1.0)
    halves = grads([(0, 2048), (2048, 4096)], 1.0)
    double = grads([(0, 4096)], 2.0)
    assert all(torch.isfinite(g).all() and float(g.abs().max()) > 0 for g in whole)
    for a, b, c, d in zip(whole, again, halves, double):
        assert torch.equal(a, b)                                              # fixed-order reductions
        assert rel_to_max(c, a) < (2e-2 if precision == 'f16' else 1e-4) and rel_to_max(d, 2 * a) < 1e-5

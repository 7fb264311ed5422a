"""The two host bindings of the one-call render ops -- torch.ops.snerf.render (TORCH_LIBRARY extension with a C++ autograd
node, csrc_torch/snerf_torch.cpp) and ops.RenderCall (ctypes) -- drive the same two C-ABI entry points: outputs, parameter
gradients, accumulation into existing gradients and the returned-gradient mode must agree BIT FOR BIT."""
import gc

import pytest
import torch

from simplenerf_amd import harness, ops, synth
from simplenerf_amd.models.ModelFactory import get_model
from tests import util

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make(kind, binding, precision='fp32', **overrides):
    cfg = synth.with_overrides(synth.make_configs(kind), hip_precision=precision, hip_host_binding=binding, **overrides)
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    return model.to(DEV)


@pytest.mark.parametrize('kind', ['config1', 'config2', 'headline_world'])
@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
def test_eval_outputs_are_identical(kind, precision):
    models = {b: make(kind, b, precision).eval() for b in ('torch_ext', 'ctypes')}
    if kind == 'config1':
        batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.random_world_rays(301, seed=4).items()}
    else:
        cam = synth.camera('fern', 0)
        batch = harness.frame_batch(cam, kind == 'config2', DEV, 300000, 301)
    with torch.no_grad():
        for retraw in (False, True):
            a, b = models['torch_ext'](batch, retraw=retraw), models['ctypes'](batch, retraw=retraw)
            assert list(a.keys()) == list(b.keys())
            for k in a:
                assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), (k, retraw)
        # the fine-depth override comes back as z_vals_fine through both
        if kind != 'config1':
            z = b['z_vals_fine'].clone()
            for m in models.values():
                m.set_random_draws({'z_vals_fine': z})
            a, b = models['torch_ext'](batch, retraw=True), models['ctypes'](batch, retraw=True)
            assert torch.equal(a['z_vals_fine'], z) and all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize('precision', ['fp32', 'f16'])
def test_training_outputs_and_gradients_are_identical(precision):
    """config 3 (four MLPs, device draws keyed by iteration and row): two backward passes (overwrite, then accumulate into the
    existing .grad), every parameter."""
    models = {b: make('config3', b, precision).train() for b in ('torch_ext', 'ctypes')}
    cam = synth.camera('fern', 0)
    grads = {}
    for name, model in models.items():
        for it in (3, 4):
            batch = harness.frame_batch(cam, True, DEV, 250000 + 100 * it, 130)
            batch['iter_num'] = it
            out = model(batch)
            assert out['rgb_fine'].requires_grad and not out['alpha_fine'].requires_grad and not out['z_vals_fine'].requires_grad
            util.grad_loss(out).backward()
        grads[name] = {k: p.grad.clone() for k, p in model.named_parameters()}
        grads[name + '/out'] = {k: v.detach().clone() for k, v in out.items()}
    for k in grads['ctypes']:
        assert torch.equal(grads['torch_ext'][k], grads['ctypes'][k]), k
    for k in grads['ctypes/out']:
        assert torch.equal(grads['torch_ext/out'][k], grads['ctypes/out'][k]), k


def test_returned_gradient_mode_and_autograd_grad():
    """hip_return_param_grads: the node hands the parameter gradients back to autograd (torch.autograd.grad works)."""
    cam = synth.camera('fern', 0)
    batch = harness.frame_batch(cam, True, DEV, 123456, 64)
    batch['iter_num'] = 0
    results = {}
    for binding in ('torch_ext', 'ctypes'):
        model = make('config3', binding, hip_return_param_grads=True).train()
        out = model(batch)
        params = list(model.parameters())
        got = torch.autograd.grad(util.grad_loss(out), params, allow_unused=True)
        assert all(p.grad is None for p in params)
        results[binding] = got
    for a, b in zip(results['torch_ext'], results['ctypes']):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))


def test_extension_frees_its_memory_without_the_cycle_collector():
    """The C++ node owns the call object (pools, saved activations) and must die with the outputs -- no reference cycle."""
    model = make('config3', 'torch_ext').train()
    cam = synth.camera('fern', 0)
    gc.disable()
    try:
        torch.cuda.synchronize()
        base = None
        for it in range(12):
            batch = harness.frame_batch(cam, True, DEV, 200000, 512)
            batch['iter_num'] = it
            out = model(batch)
            if it % 2 == 0:
                util.grad_loss(out).backward()      # (odd iterations drop the graph without a backward)
            del out
            torch.cuda.synchronize()
            used = torch.cuda.memory_allocated()
            if it == 3:
                base = used
            if it > 3:
                assert used <= base * 1.05 + (1 << 20), (it, used, base)
    finally:
        gc.enable()


def test_extension_errors_are_runtime_errors():
    model = make('config2', 'torch_ext').eval()
    cam = synth.camera('fern', 0)
    batch = harness.frame_batch(cam, True, DEV, 1000, 16)
    bad = dict(batch, rays_d=batch['rays_d'][:8])
    with pytest.raises(RuntimeError, match='rays_d'):
        with torch.no_grad():
            model(bad)
    bad = dict(batch, near_ndc=batch['near_ndc'].double())
    with pytest.raises(RuntimeError, match='float32'):
        with torch.no_grad():
            model(bad)
    del batch['view_dirs']
    with pytest.raises(RuntimeError, match='view_dirs'):
        with torch.no_grad():
            model(batch)

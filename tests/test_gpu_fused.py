"""The fused per-ray-tile render kernel (csrc/render_fused.hip; configs['model']['hip_fused_render']): an eval-mode render of
a plain coarse + fine model as ONE launch -- coarse depths, coarse MLP, compositing + inverse-CDF resampling, fine MLP,
compositing -- with each ray group's sample tile resident in LDS.  It calls the same device functions as the six-launch
path in the same order, so every output must be BIT-IDENTICAL; calls outside its scope silently take the six-launch path."""
import numpy
import pytest
import torch

from simplenerf_amd import harness, ops, synth
from simplenerf_amd.models.ModelFactory import get_model
from tests import util

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def pair(kind, binding='torch_ext', coarse_samples=None, **overrides):
    """(six-launch model, fused model) with the same synthetic weights"""
    models = []
    for fused in (False, True):
        cfg = synth.with_overrides(synth.make_configs(kind), hip_fused_render=fused, hip_host_binding=binding, **overrides)
        if coarse_samples:
            cfg['model']['coarse_mlp']['num_samples'] = coarse_samples
        model = get_model(cfg, None)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
        models.append(model.to(DEV).eval())
    return models


def launches_of(model, batch):
    """MLP-forward launches the library times for one call of the model (1 = the fused kernel, 2 = coarse + fine)"""
    ops.profile_enable(16)
    with torch.no_grad():
        model(batch)
    torch.cuda.synchronize()
    ms, samples = ops.profile_collect(ops.PROFILE_MLP_FORWARD)
    ops.profile_enable(0)
    return len(ms), sum(samples)


@pytest.mark.parametrize('binding', ['torch_ext', 'ctypes'])
@pytest.mark.parametrize('kind,count', [('config2', 1000), ('headline', 1024), ('headline', 1), ('config2', 3), ('headline', 1027)])
def test_fused_render_is_bit_identical_to_the_six_launch_path(kind, count, binding):
    plain, fused = pair(kind, binding)
    batch = harness.frame_batch(synth.camera('fern', 0), True, DEV, 190000, count)
    with torch.no_grad():
        want = plain(batch, retraw=True)
        got = fused(batch, retraw=True)
    assert sorted(got) == sorted(want)
    for k, v in want.items():
        assert torch.equal(got[k], v), (k, util.linf(got[k], v))
    assert float(want['acc_fine'].mean()) > 0.05
    s = {'config2': 64 + 192, 'headline': 128 + 256}[kind]
    assert launches_of(plain, batch) == (2, count * s) and launches_of(fused, batch) == (1, count * s)


def test_fused_render_of_world_rays_and_white_background():
    """non-NDC rays (headline_world) and model.white_bkgd"""
    plain, fused = pair('headline_world', white_bkgd=True)
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.random_world_rays(300, seed=5).items()}
    with torch.no_grad():
        want, got = plain(batch), fused(batch)
    assert sorted(got) == sorted(want) and all(torch.equal(got[k], want[k]) for k in want)


def test_calls_outside_the_fused_kernels_scope_take_the_six_launch_path():
    """Training-mode forwards (augmentation MLPs, saved activations), the fp16 modes and sample counts the kernel is not built
    for run stage by stage -- same results as with the flag off, no error."""
    cam = synth.camera('fern', 0)
    batch = harness.frame_batch(cam, True, DEV, 190000, 64)
    for overrides in ({'hip_precision': 'f16x3'}, {'coarse_samples': 48}):
        plain, fused = pair('config2', **overrides)
        with torch.no_grad():
            want, got = plain(batch), fused(batch)
        assert all(torch.equal(got[k], want[k]) for k in want), overrides
        assert launches_of(fused, batch)[0] == 2, overrides
    cfg = synth.with_overrides(synth.training_configs('fp32', num_rays=64, num_sparse=0), hip_fused_render=True)
    model = get_model(cfg, None).to(DEV).train()
    out = model(batch)
    out['rgb_fine'].sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.fine_model.parameters())


def test_fused_full_frame_equals_the_six_launch_frame():
    """harness.predict_frame (65 536-ray blocks) of the 504 x 378 fern frame: the five display outputs are identical."""
    plain, fused = pair('config2')
    cam = synth.camera('fern', 0, downscale=2)
    cfg = synth.make_configs('config2')
    want = harness.predict_frame(plain, cfg, cam, torch.device(DEV))
    got = harness.predict_frame(fused, cfg, cam, torch.device(DEV))
    assert sorted(got) == sorted(want) and all(numpy.array_equal(got[k], want[k]) for k in want)

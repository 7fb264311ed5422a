"""Loss row on the GPU: the fused HIP loss evaluation behind ``LossComputer`` against the reference's
own loss classes (G8 fixtures) and against the CPU oracle."""
import numpy
import pytest
import torch

from oracle import loss_oracle
from simplenerf_amd import ops, synth
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
from tests import util

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
REL = 5e-6   # fp32 sums in a different (fixed) order than torch's


@pytest.mark.parametrize('case', ['full', 'early', 'nosd', 'empty'])
def test_fused_losses_match_reference(case):
    g = util.load(f'losses_{case}.npz')
    configs, inp, out = util.loss_case(g, DEV)
    inp['common_data'] = {k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in inp['common_data'].items()}
    losses = LossComputer(configs).compute_losses(inp, out)
    assert inp['common_data']['poses'].dim() == 3          # un-replicated in place, like the reference
    assert float(losses["TotalLoss"].detach()) == pytest.approx(float(g['TotalLoss']), rel=REL, abs=1e-9)
    for cfg in configs['losses']:
        assert float(losses[cfg['name']]['loss_value'].detach()) == pytest.approx(float(g[f"value_{cfg['name']}"]), rel=REL, abs=1e-9), cfg['name']
    losses['TotalLoss'].backward()
    for k in util.LOSS_OUTPUT_KEYS:
        grad = out[k].grad
        grad = numpy.zeros_like(g[f'grad_{k}']) if grad is None else grad.cpu().numpy()
        scale = max(float(numpy.abs(g[f'grad_{k}']).max()), 1e-12)
        assert util.linf(grad, g[f'grad_{k}']) <= REL * scale, k


def test_loss_maps_match_reference():
    g = util.load('losses_nosd.npz')
    configs, inp, out = util.loss_case(g, DEV)
    inp['common_data'] = {k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in inp['common_data'].items()}
    losses = LossComputer(configs).compute_losses(inp, out, return_loss_maps=True)
    seen = 0
    for name, entry in losses.items():
        if name == 'TotalLoss':
            continue
        for map_name, loss_map in entry['loss_maps'].items():
            ref = g[f'map_{name}_{map_name}']
            assert util.linf(loss_map.detach().cpu().numpy(), ref) <= 1e-6 * max(1.0, float(numpy.abs(ref).max())), map_name
            seen += 1
    assert seen == 10


@pytest.mark.parametrize('case', ['full', 'nosd'])
def test_patch_masks_match_oracle_on_every_ray(case):
    g = util.load(f'losses_{case}.npz')
    configs, inp, out = util.loss_case(g, 'cpu')
    common = inp['common_data']
    mask = inp['indices_mask_nerf']
    for k1, k2 in (('depth_coarse', 'points_augmentation_depth_coarse'), ('depth_coarse', 'views_augmentation_depth_coarse'),
                   ('depth_coarse', 'depth_fine')):
        ref = loss_oracle.consistency_masks(out[k1].detach()[mask], out[k2].detach()[mask], inp['rays_o'][mask],
                                            inp['rays_d'][mask], inp['pixel_id'][mask], common['poses'], common['images'],
                                            common['intrinsics'], common['resolution'], [5, 5], 0.1)
        d = lambda t: t.to(DEV)
        m1, m2, r1, r2 = ops.patch_consistency_masks(
            d(inp['rays_o']), d(inp['rays_d']), d(out[k1].detach()), d(out[k2].detach()), d(mask), d(inp['pixel_id']),
            d(common['poses']), d(common['intrinsics'][0]), d(common['images']), [5, 5], 0.1, with_rmse=True)
        sel = mask.numpy()
        assert numpy.array_equal(m1.cpu().numpy()[sel], ref['mask1'].numpy()), (k1, k2)
        assert numpy.array_equal(m2.cpu().numpy()[sel], ref['mask2'].numpy()), (k1, k2)
        assert not m1.cpu().numpy()[~sel].any() and not m2.cpu().numpy()[~sel].any()
        assert util.linf(r1.cpu().numpy()[sel], ref['rmse1'].numpy()) <= 1e-6
        assert util.linf(r2.cpu().numpy()[sel], ref['rmse2'].numpy()) <= 1e-6
        assert 0.05 < ref['mask2'].float().mean() < 0.95


def test_large_batch_matches_float64_and_is_deterministic():
    """Size-independent properties at a full-frame ray count: value/gradient against a float64 evaluation, gradients
    linear in the upstream gradient, shared buffers summed, bitwise repeatable."""
    n = 762048
    gen = torch.Generator(device=DEV).manual_seed(5)
    rgb = torch.rand((n, 3), device=DEV, generator=gen).requires_grad_(True)
    tgt = torch.rand((n, 3), device=DEV, generator=gen)
    depth = (4 + torch.randn((n,), device=DEV, generator=gen)).requires_grad_(True)
    other = 4 + torch.randn((n,), device=DEV, generator=gen)
    keep = torch.rand((n,), device=DEV, generator=gen) < 0.7
    part = (torch.rand((n,), device=DEV, generator=gen) < 0.3) & keep

    def run():
        terms = [ops.LossTermSpec(rgb, tgt, keep, keep, 0, 1.0), ops.LossTermSpec(depth, other, part, keep, 1, 0.1),
                 ops.LossTermSpec(depth, other, None, None, 1, 0.1)]
        values, scales = ops.loss_forward(terms, 2)
        up = torch.zeros(6, device=DEV)
        up[5] = 1.0
        grads = ops.loss_backward(terms, 2, scales, up, [True, True, True])
        grads2 = ops.loss_backward(terms, 2, scales, 3 * up, [True, True, True])
        return values, grads, grads2

    values, grads, grads2 = run()
    cnt = keep.sum().double()
    e_rgb, e_d = (rgb.detach().double() - tgt.double()), (depth.detach().double() - other.double())
    v0 = (e_rgb[keep] ** 2).sum() / (3 * cnt)
    v1 = (e_d[part] ** 2).sum() / cnt
    v2 = (e_d ** 2).mean()
    expect = torch.stack([v0, v1, v2, v0, v1 + v2, v0 + 0.1 * v1 + 0.1 * v2])
    assert torch.allclose(values.double(), expect, rtol=2e-6, atol=0)
    g_rgb = 2 * e_rgb * keep[:, None] / (3 * cnt)
    g_d = 0.1 * 2 * e_d * part / cnt + 0.1 * 2 * e_d / n
    assert grads[1] is grads[2]
    assert float((grads[0].double() - g_rgb).abs().max()) <= 2e-6 * float(g_rgb.abs().max())
    assert float((grads[1].double() - g_d).abs().max()) <= 2e-6 * float(g_d.abs().max())
    assert float((grads2[1] - 3 * grads[1]).abs().max()) <= 1e-6 * float(grads[1].abs().max())
    values_b, grads_b, _ = run()
    assert torch.equal(values, values_b) and torch.equal(grads[0], grads_b[0]) and torch.equal(grads[1], grads_b[1])


def test_losses_train_the_renderer_end_to_end():
    """LossComputer on the HIP model's outputs: TotalLoss.backward() reaches every MLP's parameters."""
    from simplenerf_amd.models.ModelFactory import get_model
    scene = synth.synth_scene(0)
    batch = synth.loss_batch(scene, 192, 64, 1)
    configs = synth.make_configs('config3')
    configs['data_loader']['ndc'] = False
    configs['data_loader']['sparse_depth'] = {}
    configs['losses'] = synth.loss_configs(iter_weighted=False)
    model = get_model(configs, None).to(DEV).train()
    t = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).to(DEV)
    n = 256
    rays_d = t(batch['rays_d'])
    inp = {'iter_num': 0, 'rays_o': t(batch['rays_o']), 'rays_d': rays_d,
           'view_dirs': rays_d / rays_d.norm(dim=1, keepdim=True), 'near': torch.full((n, 1), 2.0, device=DEV),
           'far': torch.full((n, 1), 6.0, device=DEV), 'pixel_id': t(batch['pixel_id']), 'target_rgb': t(batch['target_rgb']),
           'indices_mask_nerf': t(batch['indices_mask_nerf']), 'indices_mask_sparse_depth': t(batch['indices_mask_sparse_depth']),
           'sparse_depth_values': t(batch['sparse_depth_values']),
           'common_data': {'poses': t(scene['poses'])[None], 'images': t(scene['images'])[None],
                           'intrinsics': t(scene['intrinsics'])[None], 'resolution': scene['resolution']}}
    out = model(inp)
    losses = LossComputer(configs).compute_losses(inp, out)
    assert torch.isfinite(losses['TotalLoss'])
    losses['TotalLoss'].backward()
    norms = {k: float(p.grad.norm()) for k, p in model.named_parameters() if p.grad is not None}
    for prefix in ('coarse_model', 'fine_model', 'pts_aug_coarse_model', 'views_aug_coarse_model'):
        assert any(k.startswith(prefix) and v > 0 for k, v in norms.items()), prefix


def test_patch_masks_ignore_rows_without_a_view():
    """A row whose view index is the loader's -1 fill (or past the last view) must not be dereferenced."""
    scene = synth.synth_scene(0)
    d = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).to(DEV)
    pixel_id = torch.tensor([[0, 10, 10], [-1, -1, -1], [3, 5, 5], [1, 20, 20]], dtype=torch.int32, device=DEV)
    rays = torch.zeros(4, 3, device=DEV)
    depth = torch.full((4,), 4.0, device=DEV)
    m1, m2 = ops.patch_consistency_masks(rays, rays - torch.tensor([0.0, 0.0, 1.0], device=DEV), depth, depth, None, pixel_id,
                                         d(scene['poses']), d(scene['intrinsics'][0]), d(scene['images']), [5, 5], 0.1)
    assert not bool(m1[1]) and not bool(m2[1]) and not bool(m1[2]) and not bool(m2[2])


def test_loss_errors():
    with pytest.raises(RuntimeError, match='expected a tensor on the GPU'):
        ops.LossTermSpec(torch.zeros(4), torch.zeros(4), None, None, 0, 1.0)
    z = torch.zeros(4, device=DEV)
    with pytest.raises(RuntimeError, match='terms'):
        ops.loss_forward([ops.LossTermSpec(z, z, None, None, 0, 1.0)] * 17, 1)
    with pytest.raises(RuntimeError, match='group'):
        ops.loss_forward([ops.LossTermSpec(z, z, None, None, 3, 1.0)], 2)
    with pytest.raises(RuntimeError, match='at least 2 views'):
        ops.patch_consistency_masks(torch.zeros(4, 3, device=DEV), torch.zeros(4, 3, device=DEV), z, z, None,
                                    torch.zeros(4, 3, dtype=torch.int32, device=DEV), torch.zeros(1, 4, 4, device=DEV),
                                    torch.eye(3, device=DEV), torch.zeros(1, 8, 8, 3, device=DEV), [5, 5], 0.1)


def test_loss_variants_match_oracle():
    """Configurations outside the shipped one, against the CPU oracle: augmentations that also have FINE MLPs (MSE02/03
    then read two colours each, SparseDepthMSE02/03 switch to the MAIN model's depth_fine, the augmentation depth losses
    add a fine pair) and a model without a fine MLP (coarse-only losses; CoarseFineConsistencyLoss02 contributes 0)."""
    import copy
    g = util.load('losses_full.npz')
    base_cfg, inp_cpu, out_cpu = util.loss_case(g, 'cpu')
    rng = numpy.random.RandomState(4)
    n = inp_cpu['rays_o'].shape[0]
    extra = {}
    for aug in ('points_augmentation', 'views_augmentation'):
        extra[f'{aug}_rgb_fine'] = torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(numpy.float32))
        extra[f'{aug}_depth_fine'] = out_cpu['depth_fine'].detach() + torch.from_numpy((0.3 * rng.standard_normal(n)).astype(numpy.float32))

    def run(cfg, keys):
        ref_out = {k: (out_cpu[k] if k in out_cpu else extra[k]).detach().clone().requires_grad_(True) for k in keys}
        ref = loss_oracle.compute_losses(cfg, copy.copy(inp_cpu), ref_out)
        ref['TotalLoss'].backward()
        dev_inp = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in inp_cpu.items() if k != 'common_data'}
        dev_inp['common_data'] = {k: (v.to(DEV)[None] if isinstance(v, torch.Tensor) else v) for k, v in inp_cpu['common_data'].items()}
        dev_out = {k: ref_out[k].detach().to(DEV).requires_grad_(True) for k in keys}
        got = LossComputer(cfg).compute_losses(dev_inp, dev_out)
        got['TotalLoss'].backward()
        assert float(got['TotalLoss'].detach()) == pytest.approx(float(ref['TotalLoss'].detach()), rel=REL)
        for c in cfg['losses']:
            assert float(got[c['name']]['loss_value'].detach()) == pytest.approx(float(torch.as_tensor(ref[c['name']]).detach()), rel=REL, abs=1e-9), c['name']
        for k in keys:
            a = dev_out[k].grad
            b = ref_out[k].grad
            a = torch.zeros_like(dev_out[k]) if a is None else a
            b = torch.zeros_like(ref_out[k]) if b is None else b
            assert float((a.cpu() - b).abs().max()) <= REL * max(float(b.abs().max()), 1e-12), k

    cfg = copy.deepcopy(base_cfg)
    for aug in ('points_augmentation', 'views_augmentation'):
        cfg['model'][aug]['fine_mlp'] = copy.deepcopy(cfg['model'][aug]['coarse_mlp'])
    run(cfg, list(util.LOSS_OUTPUT_KEYS) + list(extra))
    cfg = copy.deepcopy(base_cfg)
    del cfg['model']['fine_mlp']
    run(cfg, [k for k in util.LOSS_OUTPUT_KEYS if not k.endswith('_fine')])

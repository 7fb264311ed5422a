"""Optimiser row on the GPU: the HIP Adam against torch.optim.Adam's CPU path (fixture and live)."""
import numpy
import pytest
import torch

from oracle import optim_oracle
from simplenerf_amd import optim, synth
from tests import util

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_adam_matches_reference_fixture_bit_for_bit():
    g = util.load('optim_adam.npz')
    case = synth.optim_case(int(g['seed']))
    params = [torch.nn.Parameter(torch.from_numpy(p.copy()).to(DEV)) for p in case['params']]
    opt = optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
    for step, iter_num in enumerate(case['iters'], 1):
        for group in opt.param_groups:
            group['lr'] = float(g['lrs'][step - 1])
        opt.zero_grad(set_to_none=True)
        for p, grad in zip(params, case['grads'][step - 1]):
            p.grad = torch.from_numpy(grad.copy()).to(DEV)
        opt.step()
        if step in case['record']:
            for i, p in enumerate(params):
                assert numpy.array_equal(p.detach().cpu().numpy(), g[f'step{step}_param{i}']), (step, i)
                assert numpy.array_equal(opt.state[p]['exp_avg'].cpu().numpy(), g[f'step{step}_exp_avg{i}']), (step, i)
                assert numpy.array_equal(opt.state[p]['exp_avg_sq'].cpu().numpy(), g[f'step{step}_exp_avg_sq{i}']), (step, i)
    assert float(opt.state_dict()['state'][0]['step']) == len(case['iters'])      # step tensors are refreshed on read


def test_adam_on_the_full_model_matches_the_oracle_and_skips_gradless_tensors():
    """All 90 tensors of the 4-MLP model (2.27 M parameters, two launches), three steps, one tensor without gradient."""
    configs = synth.make_configs('config3')
    shapes = util.model_param_shapes(configs)
    sd = synth.synth_state_dict(shapes, 3)
    names = list(sd)
    assert len(names) == 90 and sum(v.size for v in sd.values()) == 2265488
    params = [torch.nn.Parameter(torch.from_numpy(sd[k].copy()).to(DEV)) for k in names]
    opt = optim.Adam(params, lr=5e-4, betas=(0.9, 0.999))
    ref_p = [sd[k].copy() for k in names]
    ref_m = [numpy.zeros_like(p) for p in ref_p]
    ref_v = [numpy.zeros_like(p) for p in ref_p]
    rng = numpy.random.RandomState(0)
    for step in (1, 2, 3):
        grads = [(rng.standard_normal(p.shape) * 10.0 ** rng.uniform(-5, 0)).astype(numpy.float32) for p in ref_p]
        grads[7] = None
        for p, gr in zip(params, grads):
            p.grad = None if gr is None else torch.from_numpy(gr).to(DEV)
        lr = optim_oracle.nerf_learning_rate(5e-4, 250, step * 1000)
        opt.param_groups[0]['lr'] = lr
        opt.step()
        optim_oracle.adam_step(ref_p, grads, ref_m, ref_v, step, lr)
    for i, k in enumerate(names):
        assert numpy.array_equal(params[i].detach().cpu().numpy(), ref_p[i]), k
    assert numpy.array_equal(params[7].detach().cpu().numpy(), sd[names[7]]) and len(opt.state[params[7]]) == 0


def test_step_invalidates_the_models_packed_weights():
    """The update happens outside autograd; the optimiser must still mark the parameters as changed so the renderer
    re-packs them -- two training iterations must not render with the same weights."""
    from simplenerf_amd import harness
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.make_configs('config1')
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    # a dense field (density head scaled up): with PyTorch's default initialisation most rays composite to exactly 0
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 5, 200.0, 8.0).items()})
    model = model.to(DEV).train()
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.random_world_rays(64).items()}
    opt = optim.Adam(list(model.parameters()), lr=1e-2)
    versions = [p._version for p in model.parameters()]
    with torch.no_grad():
        before = model.eval()(batch)['rgb_coarse'].clone()
    model.train()
    (model(batch)['rgb_coarse'] ** 2).mean().backward()
    opt.step()
    assert all(p._version > v for p, v in zip(model.parameters(), versions))
    with torch.no_grad():
        after = model.eval()(batch)['rgb_coarse']
    assert float(before.abs().max()) > 0.1 and float((after - before).abs().max()) > 1e-4


@pytest.mark.parametrize('precision', ['fp32', 'f16x3', 'f16'])
def test_graphed_training_pass_equals_eager(precision):
    """harness.GraphedTrainStep (one captured HIP graph: re-pack, forwards, losses, backward) against the same pass run
    eagerly: identical loss values and bit-identical parameter gradients, on the capture call and on a later replay
    with another batch; and the optimiser step between replays is seen by the next one (re-pack inside the graph)."""
    from simplenerf_amd import harness
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.training_configs(precision, num_rays=192, num_sparse=64)
    cfg['losses'] = synth.loss_configs(iter_weighted=False)
    scene = synth.training_scene(0, 3, 48, 64, sparse_fraction=0.05)
    models = []
    for _ in range(2):
        m = get_model(cfg, None)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 9, 200.0, 8.0).items()})
        models.append(m.to(DEV).train())
    eager, graphed = models
    batcher = BatchAssembler(cfg, scene, DEV)
    losses = LossComputer(cfg)
    step = harness.GraphedTrainStep(graphed, losses, batcher.get_next_batch(0))
    opt_e, opt_g = optim.Adam(list(eager.parameters()), lr=1e-3), optim.Adam(list(graphed.parameters()), lr=1e-3)
    for it in range(3):
        batch = batcher.get_next_batch(it)
        eager.set_random_draws(eager.draw_training_randomness(256, 0, DEV))
        piece = dict(batch)
        piece['common_data'] = dict(batch['common_data'])
        opt_e.zero_grad(set_to_none=True)
        ref = losses.compute_losses(piece, eager(piece))
        ref['TotalLoss'].backward()
        totals = step(batch)
        assert float(totals['TotalLoss']) == float(ref['TotalLoss'].detach()), it
        for (name, a), b in zip(eager.named_parameters(), graphed.parameters()):
            assert torch.equal(a.grad, b.grad), (it, name)
        opt_e.step()
        opt_g.step()
    for a, b in zip(eager.parameters(), graphed.parameters()):
        assert torch.equal(a, b)
    # a short batch (end of an epoch) goes through the same pass without the graph, into the same gradient buffers
    batch = {k: (v[:200] if isinstance(v, torch.Tensor) else v) for k, v in batcher.get_next_batch(3).items()}
    grad_ptrs = [p.grad.data_ptr() for p in graphed.parameters()]
    eager.set_random_draws(eager.draw_training_randomness(200, 0, DEV))
    piece = dict(batch)
    piece['common_data'] = dict(batch['common_data'])
    opt_e.zero_grad(set_to_none=True)
    ref = losses.compute_losses(piece, eager(piece))
    ref['TotalLoss'].backward()
    totals = step(batch)
    assert float(totals['TotalLoss']) == float(ref['TotalLoss'].detach())
    assert [p.grad.data_ptr() for p in graphed.parameters()] == grad_ptrs
    for a, b in zip(eager.parameters(), graphed.parameters()):
        assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize('precision', ['fp32', 'f16'])
def test_graphed_sub_batched_iteration_equals_the_eager_trainer_iteration(precision):
    """GraphedTrainStep(sub_batch_size=...) replays the reference's iteration exactly (two sub-batches, in-kernel gradient
    accumulation, per-sub-batch draws): parameters after three optimiser steps are bit-identical to
    harness.train_one_iter's, device draws included."""
    from simplenerf_amd import harness
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.training_configs(precision, num_rays=192, num_sparse=64)
    cfg['sub_batch_size'] = 128
    scene = synth.training_scene(0, 3, 48, 64, sparse_fraction=0.05)
    models = []
    for _ in range(2):
        m = get_model(cfg, None)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 9, 200.0, 8.0).items()})
        models.append(m.to(DEV).train())
    eager, graphed = models
    batch_e, batch_g = BatchAssembler(cfg, scene, DEV), BatchAssembler(cfg, scene, DEV)
    losses = LossComputer(cfg)
    opt_e, opt_g = optim.Adam(list(eager.parameters()), lr=1e-3), optim.Adam(list(graphed.parameters()), lr=1e-3)
    step = harness.GraphedTrainStep(graphed, losses, batch_g.get_next_batch(20000), warmup=1, sub_batch_size=128)
    batch_g = BatchAssembler(cfg, scene, DEV)
    for it in range(20000, 20003):
        ref = harness.train_one_iter(eager, losses, opt_e, batch_e.get_next_batch(it), 128)
        got = step(batch_g.get_next_batch(it))
        opt_g.step()
        assert float(got['TotalLoss']) == float(ref['TotalLoss']), it
    for (name, a), b in zip(eager.named_parameters(), graphed.parameters()):
        assert torch.equal(a, b), name
    # a SHORT batch (end of an epoch; 200 of 256 rows = sub-batches of 128 + 72) takes the same sub-batched pass without the
    # graph: two model -> losses -> backward rounds whose totals are summed, one draw set per sub-batch (it used
    # to run as one whole-batch pass -- another objective scale and other draws from then on)
    cut = lambda b: {k: (v[:200] if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    ref = harness.train_one_iter(eager, losses, opt_e, cut(batch_e.get_next_batch(20003)), 128)
    got = step(cut(batch_g.get_next_batch(20003)))
    opt_g.step()
    assert float(got['TotalLoss']) == float(ref['TotalLoss'])
    assert sorted(got) == sorted(ref) and all(float(got[k]) == float(ref[k]) for k in ref)
    ref = harness.train_one_iter(eager, losses, opt_e, batch_e.get_next_batch(20004), 128)      # and the replays go on exactly
    got = step(batch_g.get_next_batch(20004))
    opt_g.step()
    assert float(got['TotalLoss']) == float(ref['TotalLoss'])
    for (name, a), b in zip(eager.named_parameters(), graphed.parameters()):
        assert torch.equal(a, b), name


@pytest.mark.parametrize('precision', ['fp32', 'f16', 'f16s8'])
def test_graphed_whole_iteration_equals_the_eager_trainer_iteration(precision):
    """harness.GraphedIteration: batch assembly, draws, the sub-batched pass AND the Adam update replayed from ONE HIP graph,
    the per-iteration scalars (epoch positions, iteration number, Adam's factors with the decayed learning rate) read from a
    device-resident record that the graph's first node refreshes from a pinned host ring.  Twelve iterations across an epoch
    boundary of the sparse-depth stream (a short batch, run eagerly into the static gradient buffers) against the eager
    trainer iteration: loss values and every parameter bit-identical.  The ring has 4 slots here, so the slot guard is used."""
    from simplenerf_amd import harness
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.training_configs(precision, num_rays=192, num_sparse=64)
    cfg['sub_batch_size'] = 128
    cfg['losses'] = synth.loss_configs(iter_weighted=False)
    scene = synth.training_scene(0, 3, 48, 64, sparse_fraction=0.05)
    models = []
    for _ in range(2):
        m = get_model(cfg, None)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 9, 200.0, 8.0).items()})
        models.append(m.to(DEV).train())
    eager, graphed = models
    batch_e, batch_g = BatchAssembler(cfg, scene, DEV), BatchAssembler(cfg, scene, DEV)
    assert batch_e.sparse_candidates.shape[0] // 64 < 11          # the sparse-depth epoch ends inside the run
    losses = LossComputer(cfg)
    decayer = get_lr_decayer(cfg)
    opt_e, opt_g = optim.Adam(list(eager.parameters()), lr=5e-4), optim.Adam(list(graphed.parameters()), lr=5e-4)
    step = harness.GraphedIteration(graphed, losses, opt_g, batch_g, decayer, sub_batch_size=128, slots=4)
    short = 0
    for it in range(20000, 20012):
        for group in opt_e.param_groups:
            group['lr'] = decayer.get_updated_learning_rate(it)
        batch = batch_e.get_next_batch(it)
        short += batch['rays_o'].shape[0] < 256
        ref = harness.train_one_iter(eager, losses, opt_e, batch, 128)
        got = step(it)
        assert float(got['TotalLoss']) == float(ref['TotalLoss']), it
        assert sorted(got) == sorted(ref) and all(float(got[k]) == float(ref[k]) for k in ref), it
    assert short >= 1
    for (name, a), b in zip(eager.named_parameters(), graphed.parameters()):
        assert torch.equal(a, b), name
    # the optimiser's bookkeeping followed the replays: same step counts, same state -- a checkpoint taken now is the eager one
    sd_e, sd_g = opt_e.state_dict(), opt_g.state_dict()
    for k in sd_e['state']:
        assert float(sd_e['state'][k]['step']) == float(sd_g['state'][k]['step']) == 12.0
        assert torch.equal(sd_e['state'][k]['exp_avg'], sd_g['state'][k]['exp_avg'])
        assert torch.equal(sd_e['state'][k]['exp_avg_sq'], sd_g['state'][k]['exp_avg_sq'])


@pytest.mark.parametrize('precision', ['f16x3', 'f16'])
def test_graph_replays_stay_exact_while_the_gradients_shrink(precision):
    """Regression (r02): with the fp16 modes' region-maximum table cleared by a captured hipMemsetAsync node, graph
    replays kept the maxima of earlier iterations; once the gradients had shrunk by a few powers of two (third iteration
    of a fresh model at 1280 rows) the weight gradients were 30 % off -- a graphed fp16 training run stalled at 24 dB where
    the eager one reached 34 dB.  The table is now cleared by a kernel: six iterations, parameters bit-identical."""
    from simplenerf_amd import harness
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.training_configs(precision, num_rays=1024, num_sparse=256)
    cfg['sub_batch_size'] = 1280
    cfg['losses'] = synth.loss_configs(iter_weighted=False)
    scene = synth.training_scene(0, 3, 96, 128, sparse_fraction=0.5)      # enough sparse pixels: no short batches
    models = []
    for _ in range(2):
        torch.manual_seed(0)
        models.append(get_model(cfg, None).to(DEV).train())
    eager, graphed = models
    batch_e, batch_g = BatchAssembler(cfg, scene, DEV), BatchAssembler(cfg, scene, DEV)
    losses = LossComputer(cfg)
    opt_e, opt_g = optim.Adam(list(eager.parameters()), lr=5e-4), optim.Adam(list(graphed.parameters()), lr=5e-4)
    step = harness.GraphedTrainStep(graphed, losses, batch_g.get_next_batch(0), sub_batch_size=1280)
    batch_g = BatchAssembler(cfg, scene, DEV)
    first = None
    for it in range(6):
        ref = harness.train_one_iter(eager, losses, opt_e, batch_e.get_next_batch(it), 1280)
        got = step(batch_g.get_next_batch(it))
        opt_g.step()
        first = float(ref['TotalLoss']) if first is None else first
        assert float(got['TotalLoss']) == float(ref['TotalLoss']), it
        for (name, a), b in zip(eager.named_parameters(), graphed.parameters()):
            assert torch.equal(a, b), (it, name)
    assert float(ref['TotalLoss']) < 0.7 * first        # the run did move (the gradients did shrink)


def test_readme_quick_start_runs():
    """The snippet in README.md (training iteration + frame render through the reference's interfaces)."""
    from simplenerf_amd import harness
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.training_configs('f16x3', num_rays=256, num_sparse=64)
    model = get_model(cfg, None).cuda().train()
    batches = BatchAssembler(cfg, synth.training_scene(0, 3, 48, 64, 0.05), 'cuda:0')
    losses, opt = LossComputer(cfg), optim.Adam(list(model.parameters()), lr=5e-4)
    totals = harness.train_one_iter(model, losses, opt, batches.get_next_batch(0), 160)
    assert torch.isfinite(totals['TotalLoss'])
    cam = synth.camera('fern', 0, downscale=16)
    frame = harness.render_frame(model.eval(), cam, True, 'cuda:0')
    h, w = cam['resolution']
    assert frame['rgb_fine'].shape == (h * w, 3) and torch.isfinite(frame['rgb_fine']).all()


def test_single_pass_iteration_matches_sub_batched():
    """train_one_iter(single_pass=True): one model forward/backward over the whole batch with every loss still normalised
    per sub-batch -- same loss values and parameter gradients as the reference's sub-batched iteration (jitter and noise
    off, so that the draws, which are keyed by the training call, do not enter)."""
    from simplenerf_amd import harness, optim as snerf_optim, synth
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.models.ModelFactory import get_model

    def run(single_pass):
        cfg = synth.with_overrides(synth.training_configs('fp32', num_rays=512, num_sparse=512), perturb=False, raw_noise_std=0.0)
        cfg['sub_batch_size'] = 512
        cfg['losses'] = synth.loss_configs(iter_weighted=False)
        model = get_model(cfg, None)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
        model = model.to('cuda:0').train()
        batch = BatchAssembler(cfg, synth.training_scene(0, 3, 96, 128, sparse_fraction=0.05), 'cuda:0').get_next_batch(0)
        opt = snerf_optim.Adam(list(model.parameters()), lr=0.0)
        totals = harness.train_one_iter(model, LossComputer(cfg), opt, batch, cfg['sub_batch_size'], single_pass=single_pass)
        return {k: float(v) for k, v in totals.items()}, {n: p.grad.clone() for n, p in model.named_parameters()}

    ref_loss, ref_grads = run(False)
    got_loss, got_grads = run(True)
    for k, v in ref_loss.items():
        assert abs(got_loss[k] - v) <= 1e-5 * max(abs(v), 1e-6), (k, got_loss[k], v)
    for k, g in ref_grads.items():
        err = float((got_grads[k] - g).abs().max() / max(float(g.abs().max()), 1e-30))
        assert err < 1e-4, (k, err)


def test_training_passes_free_their_memory_without_the_cycle_collector():
    """The autograd node of a training pass owns the saved activations (GBs at full size); nothing may hold it in a
    reference cycle -- with Python's cycle collector switched off, memory after the 6th iteration equals memory after
    the 3rd."""
    import gc
    from simplenerf_amd import harness
    from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
    from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
    from simplenerf_amd.models.ModelFactory import get_model
    cfg = synth.training_configs('fp32', num_rays=192, num_sparse=64)
    cfg['sub_batch_size'] = 128
    cfg['losses'] = synth.loss_configs(iter_weighted=False)
    model = get_model(cfg, None).to(DEV).train()
    batcher = BatchAssembler(cfg, synth.training_scene(0, 3, 48, 64, sparse_fraction=0.05), DEV)
    losses, opt = LossComputer(cfg), optim.Adam(list(model.parameters()), lr=1e-3)
    gc.collect()
    gc.disable()
    try:
        seen = []
        for it in range(6):
            harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it), cfg['sub_batch_size'])
            with torch.no_grad():
                model.eval()
                model(harness.frame_batch(synth.camera('fern', 0), True, DEV, 0, 64))
                model.train()
            torch.cuda.synchronize()
            seen.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert seen[5] == seen[2], seen

#!/usr/bin/env python3
"""Config 5 on one MI355X: the reference's training iteration (Trainer.train_one_iter, src/Trainer01.py:60-107, with the
shipped LLFF settings: 2048 pixel rays + 2048 sparse-depth rays, sub-batches of 2048) with every stage on the device:
batch assembly -> 4-MLP forward -> nine losses -> backward -> Adam with the decayed learning rate.
    python tools/measure_train.py > gpurun_out/train.json          (SNERF_PREC=f16x3 for the split-precision kernels, f16 for the 16-bit mode)
Also times the stages around the renderer separately and, for comparison, the same loss set and optimiser step done
with stock torch ops on the GPU (what the reference's loss / optimiser code would launch)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplenerf_amd import harness, optim, synth  # noqa: E402
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler  # noqa: E402
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer  # noqa: E402
from simplenerf_amd.lr_decayers.LearningRateDecayerFactory import get_lr_decayer  # noqa: E402
from simplenerf_amd.models.ModelFactory import get_model  # noqa: E402

DEV = torch.device('cuda', 0)
FLOP = {'main': 2 * 593408, 'ptsaug': 2 * 577280, 'viewsaug': 2 * 492032}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    cfg = synth.training_configs(os.environ.get('SNERF_PREC', 'fp32'))
    model = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    model = model.to(DEV).train()
    batcher = BatchAssembler(cfg, synth.training_scene(), DEV)
    losses = LossComputer(cfg)
    opt = optim.Adam(list(model.parameters()), lr=cfg['optimizer']['lr_initial'],
                     betas=(cfg['optimizer']['beta1'], cfg['optimizer']['beta2']))
    decayer = get_lr_decayer(cfg)
    state = {'iter': 20000}            # past 10000: the three patch-consistency losses are switched on

    def step():
        it = state['iter']
        state['iter'] += 1
        lr = decayer.get_updated_learning_rate(it)
        for group in opt.param_groups:
            group['lr'] = lr
        return harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it), cfg['sub_batch_size'],
                                      single_pass=os.environ.get('SNERF_SINGLE_PASS') == '1')

    n = cfg['data_loader']['num_rays'] + cfg['data_loader']['sparse_depth']['num_rays']
    res = {}
    dt = timed(step, 5)
    fwd_flop = n * (64 * (FLOP['main'] + FLOP['ptsaug'] + FLOP['viewsaug']) + 192 * FLOP['main'])
    res['train_iteration'] = {'ms': dt * 1e3, 'rays_per_s': n / dt, 'algorithmic_tflops': 3 * fwd_flop / dt / 1e12,
                              'precision': os.environ.get('SNERF_PREC', 'fp32'), 'rows': n,
                              'single_pass': os.environ.get('SNERF_SINGLE_PASS') == '1'}
    totals = step()
    res['loss_values'] = {k: float(v) for k, v in totals.items()}

    # the stages either side of the renderer, alone
    res['batch_assembly_ms'] = timed(lambda: batcher.get_next_batch(0), 20) * 1e3
    batch = batcher.get_next_batch(0)
    first = {k: (v[:2048] if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    with torch.no_grad():
        out = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model(first).items()}

    def loss_pass():
        piece = dict(first)
        piece['common_data'] = dict(batch['common_data'])
        losses.compute_losses(piece, out)['TotalLoss'].backward()

    res['losses_forward_backward_ms'] = timed(loss_pass, 20) * 1e3
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    res['adam_step_ms'] = timed(opt.step, 20) * 1e3
    stock = torch.optim.Adam(list(model.parameters()), lr=5e-4)
    res['adam_step_torch_ms'] = timed(stock.step, 20) * 1e3
    res['peak_memory_gb'] = torch.cuda.max_memory_allocated() / 2 ** 30
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

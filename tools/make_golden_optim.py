#!/usr/bin/env python3
"""G10: golden vectors for the optimiser row.

    PYTHONDONTWRITEBYTECODE=1 python3 tools/make_golden_optim.py

Runs what the reference's trainer runs (src/Trainer01.py:293-295, :102, :516-517): ``torch.optim.Adam`` on the CPU with
the learning rate of the reference's ``NeRFLearningRateDecayer`` written into ``param_groups`` before every step.
Parameters and per-step gradients come from ``synth.optim_case`` (seeded); the fixture stores the learning rates and
the parameters / first / second moments after selected steps, plus learning rates of both reference decayers at a
spread of iterations.
"""
import os
import sys

import numpy
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference/src')

from lr_decayers.LearningRateDecayerFactory import get_lr_decayer  # noqa: E402  (the reference)

from simplenerf_amd import synth  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def main():
    configs = {'optimizer': {'lr_decayer_name': 'NeRFLearningRateDecayer01', 'lr_initial': 5e-4, 'lr_decay': 250,
                             'beta1': 0.9, 'beta2': 0.999}}
    decayer = get_lr_decayer(configs)
    case = synth.optim_case(seed=0)
    params = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in case['params']]
    opt = torch.optim.Adam(params, lr=configs['optimizer']['lr_initial'],
                           betas=(configs['optimizer']['beta1'], configs['optimizer']['beta2']))
    arrays = {'seed': 0, 'iters': numpy.array(case['iters'])}
    lrs = []
    for step, iter_num in enumerate(case['iters']):
        lr = decayer.get_updated_learning_rate(iter_num)
        lrs.append(lr)
        for group in opt.param_groups:
            group['lr'] = lr
        opt.zero_grad(set_to_none=True)
        for p, g in zip(params, case['grads'][step]):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
        if step + 1 in case['record']:
            for i, p in enumerate(params):
                st = opt.state[p]
                arrays[f'step{step + 1}_param{i}'] = p.detach().numpy().copy()
                arrays[f'step{step + 1}_exp_avg{i}'] = st['exp_avg'].numpy().copy()
                arrays[f'step{step + 1}_exp_avg_sq{i}'] = st['exp_avg_sq'].numpy().copy()
    arrays['lrs'] = numpy.array(lrs, dtype=numpy.float64)
    probe = numpy.array([0, 1, 999, 1000, 12345, 250000, 499999], dtype=numpy.int64)
    arrays['probe_iters'] = probe
    arrays['nerf_lr'] = numpy.array([decayer.get_updated_learning_rate(int(i)) for i in probe], dtype=numpy.float64)
    mip = get_lr_decayer({'num_iterations': 500000,
                          'optimizer': {'lr_decayer_name': 'MipNeRFLearningRateDecayer01', 'lr_initial': 5e-4,
                                        'lr_final': 5e-6, 'lr_decay_steps': 2500, 'lr_decay_mult': 0.01}})
    arrays['mip_lr'] = numpy.array([mip.get_updated_learning_rate(int(i)) for i in probe], dtype=numpy.float64)
    sd = opt.state_dict()
    arrays['state_dict_group_keys'] = numpy.array(sorted(sd['param_groups'][0].keys()))
    arrays['state_dict_state_keys'] = numpy.array(sorted(sd['state'][0].keys()))
    arrays['state_dict_step'] = numpy.asarray(sd['state'][0]['step'])
    path = os.path.join(OUT, 'optim_adam.npz')
    numpy.savez_compressed(path, **arrays)
    print(f'optim_adam.npz: {os.path.getsize(path) / 1024:.0f} KiB', lrs, arrays['state_dict_group_keys'], arrays['state_dict_step'])


if __name__ == '__main__':
    main()

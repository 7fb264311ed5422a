#!/usr/bin/env python3
"""Print per-output L-infinity differences between the HIP renderer (cuda:0) and the reference's golden outputs.
Run on a GPU box:  python tools/parity_report.py > gpurun_out/parity.txt"""
import os
import sys

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nerf_oracle as oracle  # noqa: E402
from simplenerf_amd import synth  # noqa: E402
from simplenerf_amd.models.ModelFactory import get_model  # noqa: E402
from tests import util  # noqa: E402

DEV = 'cuda:0'


def report(tag, out, ref):
    print(f'== {tag}')
    for k in sorted(ref):
        a = out[k].detach().cpu().numpy().astype(numpy.float64)
        b = ref[k].astype(numpy.float64)
        d = numpy.abs(a - b)
        print(f'   {k:45s} max|ref| {numpy.abs(b).max():10.4g}  Linf {d.max():10.3e}  rel {(d / numpy.maximum(numpy.abs(b), 1)).max():10.3e}'
              f'  frac>1e-5 {float((d > 1e-5 * max(1, numpy.abs(b).max())).mean()):.4f}')


def main():
    for kind in ('config1', 'config2', 'headline', 'headline_world'):
        for profile in ('plain', 'dense'):
            g = util.load(f'e2e_{kind}_{profile}.npz')
            cfg = synth.make_configs(kind)
            model = get_model(cfg, None)
            model.load_state_dict(util.golden_params(cfg, g))
            model = model.to(DEV).eval()
            batch = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
            with torch.no_grad():
                out = model(batch, retraw=True)
            report(f'eval {kind} {profile}', out, {k[4:]: v for k, v in g.items() if k.startswith('out_')})
    for variant, profile in (('det', 'dense'), ('rand', 'dense'), ('rand', 'plain')):
        g = util.load(f'e2e_config3_train_{variant}_{profile}.npz')
        cfg = synth.with_overrides(synth.make_configs('config3'), perturb=bool(g['perturb']),
                                   raw_noise_std=float(g['raw_noise_std']))
        model = get_model(cfg, None)
        model.load_state_dict(util.golden_params(cfg, g))
        model = model.to(DEV).train()
        batch = {k: v.to(DEV) for k, v in util.golden_batch(g).items()}
        model.set_random_draws(oracle.replay_reference_draws(cfg, batch['rays_o'].shape[0], int(g['torch_seed']))[0])
        with torch.no_grad():
            out = model(batch)
        report(f'train config3 {variant} {profile}', out, {k[4:]: v for k, v in g.items() if k.startswith('out_')})


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Probe: error of the 16-bit mode (and f16x3 as a sanity check) against the oracle for every MLP layout, forward outputs
and parameter gradients.  python tools/probes/f16_check.py"""
import os
import sys

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nerf_oracle as oracle  # noqa: E402
from simplenerf_amd import ops, synth  # noqa: E402
from tests import util  # noqa: E402
from tests.test_gpu_kernels import LAYOUTS, abi_param_list  # noqa: E402

DEV = 'cuda:0'


def rel_l2(got, ref):
    ref = ref.detach().cpu().double(); got = got.detach().cpu().double()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


def rel_max(got, ref):
    ref = ref.detach().cpu().double(); got = got.detach().cpu().double()
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


for precision in ('f16x3', 'f16'):
    for layout in ('main', 'ptsaug', 'viewsaug'):
        for depth, width, vwidth in ((8, 256, 128), (4, 128, 64)):
            cfg = synth.mlp_config(64, depth=depth, width=width, views_width=vwidth, **LAYOUTS[layout])
            sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 31, 50.0, 1.0)
            rng = numpy.random.RandomState(depth)
            n, s = 7, 45
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
            sig0, rgb0 = mlp.forward(o.to(DEV), dd.to(DEV), v.to(DEV), z.to(DEV), noise.to(DEV), prec)
            sigma, rgb, saved = mlp.forward_train(o.to(DEV), dd.to(DEV), v.to(DEV), z.to(DEV), noise.to(DEV), prec)
            torch.cuda.synchronize()
            same = bool((sig0 == sigma).all() and (rgb0 == rgb).all())
            print(f'{precision} {layout} {depth}x{width}: sigma rel_linf {util.rel_linf(sigma, ref["sigma"]):.2e} '
                  f'rgb linf {util.linf(rgb, ref["rgb"]):.2e} eval==train {same}', flush=True)
            grads = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), [tuple(p.shape) for p in plist], prec)
            torch.cuda.synchronize()
            names = [k for k in abi_param_list({k: k for k in sd})]
            worst = max(((rel_l2(g, params[nm].grad), rel_max(g, params[nm].grad), nm) for nm, g in zip(names, grads)))
            print(f'    grads: worst rel_l2 {worst[0]:.2e} (rel_max {worst[1]:.2e}) at {worst[2]}', flush=True)
            if worst[0] > 0.2:
                for nm, g in zip(names, grads):
                    print(f'      {nm:34s} l2 {rel_l2(g, params[nm].grad):.2e} max {rel_max(g, params[nm].grad):.2e}')

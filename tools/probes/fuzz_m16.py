"""Randomised agreement sweep of the fp16-mode inference kernels (16x16x32 layout) against the fp32 kernel of the same
library: random ray / sample counts (ragged tails, single samples, several workgroups), random weights and gains.
f16x3 must agree to 2e-5 (colour) / 1e-4 relative (density, gain-amplified); the 16-bit mode within its own tolerance.
usage: fuzz_m16.py [cases, default 60]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from simplenerf_amd import ops, synth
from simplenerf_amd.synth import abi_param_list
from tests import util

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = numpy.random.RandomState(11)
cfg = synth.mlp_config(64)
worst = {1: [0.0, 0.0], 2: [0.0, 0.0]}
for case in range(cases):
    seed = int(rng.randint(1, 10000)); gain = float(rng.choice([1.0, 30.0, 300.0])); shift = float(rng.uniform(-8, 8))
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), seed, gain, shift)
    mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()}))
    n = int(rng.choice([1, 2, 3, 17, 64, 257, 1024])); s = int(rng.choice([1, 5, 31, 32, 33, 64, 100, 192, 256]))
    o = torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(numpy.float32)).cuda()
    d = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32)).cuda()
    v = d / d.norm(dim=1, keepdim=True)
    z = torch.from_numpy(numpy.sort(rng.uniform(0, 4, (n, s)).astype(numpy.float32), axis=1)).cuda()
    noise = torch.from_numpy(rng.standard_normal((n, s, 1)).astype(numpy.float32)).cuda() if case % 3 == 0 else None
    ref_sigma, ref_rgb = mlp.forward(o, d, v, z, noise, precision=0)
    for prec in (1, 2):
        sigma, rgb = mlp.forward(o, d, v, z, noise, precision=prec)
        assert torch.isfinite(sigma).all() and torch.isfinite(rgb).all(), (case, prec)
        es = float((sigma - ref_sigma).abs().max() / ref_sigma.abs().max().clamp(min=1.0)); er = float((rgb - ref_rgb).abs().max())
        worst[prec][0] = max(worst[prec][0], es); worst[prec][1] = max(worst[prec][1], er)
        tol_s, tol_r = (1e-4, 2e-5) if prec == 1 else (2e-2, 2e-3)
        assert es < tol_s and er < tol_r, (case, prec, n, s, seed, gain, es, er)
print(f'{cases} cases ok; worst f16x3: density rel {worst[1][0]:.2e}, colour {worst[1][1]:.2e}; worst 16-bit: density rel {worst[2][0]:.2e}, colour {worst[2][1]:.2e}')

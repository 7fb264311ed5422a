#!/usr/bin/env python3
"""Throughput of the layered fp32 MLP path (csrc/mlp_generic.hip) on shapes the fused kernels do not cover, beside the fused
fp32 kernel on the 8 x 256 main MLP at the same sample count: ms per call and algorithmic TFLOP/s (2 x MACs of the Linear
layers) of the inference forward, the activation-keeping forward and the backward.
    python tools/probes/time_layered.py [samples, default 262144]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplenerf_amd import _lib  # noqa: E402

if os.environ.get('SNERF_LIB'):           # A/B builds (tools/probes/build_variant.py)
    _lib.LIB_PATH = os.path.abspath(os.environ['SNERF_LIB'])
from simplenerf_amd import ops, synth  # noqa: E402
from tests import util  # noqa: E402

DEV = 'cuda:0'


def macs(cfg):
    shapes = util.mlp_param_shapes(cfg)
    return sum(s[0] * s[1] for k, s in shapes.items() if k.endswith('.weight'))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    s = 256
    n = total // s
    gen = torch.Generator().manual_seed(0)
    o = torch.rand(n, 3, generator=gen).to(DEV)
    d = torch.rand(n, 3, generator=gen).to(DEV)
    v = d / d.norm(dim=1, keepdim=True)
    z = torch.sort(torch.rand(n, s, generator=gen), 1)[0].to(DEV)
    rows = []
    for label, kw in (('fused 8x256 / views 128 (reference point)', dict(depth=8, width=256, views_width=128)),
                      ('layered 8x512 / views 256', dict(depth=8, width=512, views_width=256)),
                      ('layered 8x256 / views 2x128', dict(depth=8, width=256, views_width=128, views_depth=2)),
                      ('layered 8x64 / views 32', dict(depth=8, width=64, views_width=32))):
        cfg = synth.mlp_config(64, **kw)
        sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 3, 30.0, 0.5)
        plist = synth.abi_param_list({k: torch.from_numpy(a).to(DEV) for k, a in sd.items()})
        mlp = ops.PackedMlp(cfg, DEV)
        mlp.pack(plist)
        flop = 2.0 * macs(cfg) * n * s
        fwd = timed(lambda: mlp.forward(o, d, v, z, None))
        sigma, rgb, saved = mlp.forward_train(o, d, v, z, None)
        fwd_train = timed(lambda: mlp.forward_train(o, d, v, z, None))
        gs, gc = torch.ones_like(sigma), torch.ones_like(rgb)
        shapes = [tuple(p.shape) for p in plist]
        bwd = timed(lambda: mlp.backward(saved, sigma, rgb, gs, gc, shapes))
        rows.append({'mlp': label, 'samples': n * s, 'forward_ms': fwd, 'forward_tflops': flop / fwd / 1e9,
                     'forward_keeping_ms': fwd_train, 'backward_ms': bwd, 'backward_tflops': 2 * flop / bwd / 1e9,
                     'fraction_of_fp32_mfma_peak_forward': flop / fwd / 1e9 / 157.3})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Two torch streams: the MLP backward (K7) of one level on stream A while stream B runs the compositing backward (K6) of another
level and feeds it to that level's MLP backward -- what render.hip's side-by-side backward does, without render.hip.
Reports which of the three results (A's gradients, K6's outputs, B's gradients) ever differ between repetitions.
usage: concurrent_k6_mlp.py [precision] [repeats] [lib.so]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy  # noqa: E402
import torch  # noqa: E402
from simplenerf_amd import _lib, ops  # noqa: E402

if len(sys.argv) > 3:
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
from tests.test_gpu_f16 import abi_param_list, mlp_case  # noqa: E402

DEV = 'cuda:0'
precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 100
prec = ops.PRECISIONS[precision]
N, S = 512, 192


def level(layout):
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case(layout, (8, 256, 128), N, S)
    plist = abi_param_list({k: torch.from_numpy(v).to(DEV) for k, v in sd.items()})
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    dev = [t.to(DEV) for t in inputs]
    sigma, rgb, saved = mlp.forward_train(*dev, prec)
    return dict(mlp=mlp, sigma=sigma, rgb=rgb, saved=saved, g_sigma=g_sigma.to(DEV), g_rgb=g_rgb.to(DEV),
                shapes=[tuple(p.shape) for p in plist], depths=dev[3], dirs=dev[1],
                work=torch.empty(mlp.backward_workspace_floats(N, S), dtype=torch.float32, device=DEV))


a, b = level('main'), level(sys.argv[4] if len(sys.argv) > 4 else 'ptsaug')
rng = numpy.random.RandomState(3)
g_ray = torch.from_numpy(rng.standard_normal((N, 3)).astype(numpy.float32)).to(DEV)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def once():
    with torch.cuda.stream(sa):
        ga = a['mlp'].backward(a['saved'], a['sigma'], a['rgb'], a['g_sigma'], a['g_rgb'], a['shapes'], prec, work=a['work'])
    with torch.cuda.stream(sb):
        ds, dr = ops.composite_backward(b['sigma'], torch.sigmoid(b['rgb']) if False else b['rgb'], b['depths'], b['dirs'], False, False,
                                        None, None, g_ray, None, None, None)
        gb = b['mlp'].backward(b['saved'], b['sigma'], b['rgb'], ds, dr, b['shapes'], prec, work=b['work'])
    torch.cuda.synchronize()
    return ga, (ds, dr), gb


ref = once()
bad = {'A gradients': 0, 'K6 outputs': 0, 'B gradients': 0}
for i in range(repeats):
    got = once()
    for key, x, y in zip(bad, ref, got):
        if not all(torch.equal(p, q) for p, q in zip(x, y)):
            bad[key] += 1
print(precision, 'repeats', repeats, bad)

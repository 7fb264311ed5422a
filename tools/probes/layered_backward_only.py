#!/usr/bin/env python3
"""The layered path's backward of one MLP shape at 262 144 samples, a few times and nothing else: a target for rocprofv3.
usage: layered_backward_only.py [width] [views_width] [repetitions]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplenerf_amd import _lib  # noqa: E402

if os.environ.get('SNERF_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SNERF_LIB'])
from simplenerf_amd import ops, synth  # noqa: E402
from tests import util  # noqa: E402

DEV = 'cuda:0'
width = int(sys.argv[1]) if len(sys.argv) > 1 else 512
views = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n, s = 1024, 256
gen = torch.Generator().manual_seed(0)
o = torch.rand(n, 3, generator=gen).to(DEV)
d = torch.rand(n, 3, generator=gen).to(DEV)
v = d / d.norm(dim=1, keepdim=True)
z = torch.sort(torch.rand(n, s, generator=gen), 1)[0].to(DEV)
cfg = synth.mlp_config(64, depth=8, width=width, views_width=views)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 3, 30.0, 0.5)
plist = synth.abi_param_list({k: torch.from_numpy(a).to(DEV) for k, a in sd.items()})
mlp = ops.PackedMlp(cfg, DEV)
mlp.pack(plist)
sigma, rgb, saved = mlp.forward_train(o, d, v, z, None)
gs, gc = torch.ones_like(sigma), torch.ones_like(rgb)
shapes = [tuple(p.shape) for p in plist]
torch.cuda.synchronize()
for _ in range(reps):
    mlp.backward(saved, sigma, rgb, gs, gc, shapes)
torch.cuda.synchronize()
print('done')

#!/usr/bin/env python3
"""The layered path's activation-keeping forward of the 8 x 512 / views 256 MLP at 262 144 samples, a few times and nothing
else: a target for rocprofv3 (--kernel-trace --stats, --pmc ...) whose gemm_kernel<128,128> rows are then that one GEMM shape
(262 144 x 512 x 512) for nine launches in eleven.   usage: layered_forward_only.py [repetitions]"""
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
n, s = 1024, 256
gen = torch.Generator().manual_seed(0)
o = torch.rand(n, 3, generator=gen).to(DEV)
d = torch.rand(n, 3, generator=gen).to(DEV)
v = d / d.norm(dim=1, keepdim=True)
z = torch.sort(torch.rand(n, s, generator=gen), 1)[0].to(DEV)
cfg = synth.mlp_config(64, depth=8, width=512, views_width=256)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 3, 30.0, 0.5)
mlp = ops.PackedMlp(cfg, DEV)
mlp.pack(synth.abi_param_list({k: torch.from_numpy(a).to(DEV) for k, a in sd.items()}))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    mlp.forward_train(o, d, v, z, None)
torch.cuda.synchronize()
print('done')

#!/usr/bin/env python3
"""Soak of the side-by-side levels on the SIX-level model (augmentation MLPs at the fine level too: the case of DESIGN 10.6): N
plain gradient-descent steps over 512-ray passes (every level side by side), each on another block of rays, on a given build of
the library; prints a SHA-256 of every parameter.  The shipped library and gpurun_abl_noside.so (levels in order;
tools/probes/build_variant.py noside --only render -DSNERF_PROBE_NO_SIDE_BY_SIDE) must end with the SAME hash.
    usage: soak_six_levels.py <lib.so> [iterations] [precision]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simplenerf_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
precision = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
from tests import test_gpu_side_by_side as t  # noqa: E402

model = t._model(precision, 'ctypes', 'config3f')
first = last = None
for it in range(iters):
    batch = t._batch(512, first=(it * 512) % 65536)
    batch['iter_num'] = it
    out = model(batch)
    model.zero_grad(set_to_none=True)
    loss = t._loss(out)
    loss.backward()
    with torch.no_grad():
        for p in model.parameters():
            if p.grad is not None:
                p.add_(p.grad, alpha=-1e-7)
    if it == 0:
        first = float(loss)
    last = float(loss)
torch.cuda.synchronize()
digest = hashlib.sha256()
for p in model.parameters():
    digest.update(p.detach().cpu().numpy().tobytes())
print(json.dumps({'lib': os.path.basename(sys.argv[1]), 'precision': precision, 'iterations': iters, 'first_loss': first, 'last_loss': last,
                  'parameters_sha256': digest.hexdigest()}))

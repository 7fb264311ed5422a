#!/usr/bin/env python3
"""Repeat one small training pass (512 rays: every level side by side) and report which parameter gradients differ between
repetitions, on a given build of the library.   usage: side_by_side_determinism.py <lib.so> [precision] [kind] [binding] [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simplenerf_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16s8'
kind = sys.argv[3] if len(sys.argv) > 3 else 'config3f'
binding = sys.argv[4] if len(sys.argv) > 4 else 'ctypes'
repeats = int(sys.argv[5]) if len(sys.argv) > 5 else 40
from tests import test_gpu_side_by_side as t  # noqa: E402

model = t._model(precision, binding, kind)
small = t._batch(512)
ref = None
bad = {}
for i in range(repeats):
    out = model(small)
    model.zero_grad(set_to_none=True)
    t._loss(out).backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    if ref is None:
        ref = grads
        continue
    for n, g in grads.items():
        if not torch.equal(g, ref[n]):
            d = float((g - ref[n]).abs().max()) / (float(ref[n].abs().max()) + 1e-30)
            bad.setdefault(n, []).append((i, d))
            if os.environ.get('SNERF_PROBE_WHERE'):
                w = (g != ref[n]).nonzero()
                lo, hi = w.min(0).values.tolist(), w.max(0).values.tolist()
                print('rep', i, n, tuple(g.shape), 'elements', int(w.shape[0]), 'box', lo, hi, 'rel %.1e' % d, flush=True)
print(os.path.basename(sys.argv[1]), precision, kind, binding, 'repeats', repeats, 'parameters that differed:', len(bad))
for n, v in sorted(bad.items()):
    print('  ', n, 'in', len(v), 'repetitions, worst relative', max(d for _, d in v))

#!/usr/bin/env python3
"""Which words of the backward's workspace differ between repetitions of one small training pass (levels side by side)?
Keeps the workspace of every snerf_render_backward call and compares each level's region -- [d sigma | d rgb | MLP-backward
scratch] -- with the first repetition's.   usage: side_by_side_capture.py <lib.so> [precision] [kind] [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simplenerf_amd import _lib, ops  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
kind = sys.argv[3] if len(sys.argv) > 3 else 'config3f'
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 60
from tests import test_gpu_side_by_side as t  # noqa: E402

seen = {}


class TorchShim:
    def __getattr__(self, name):
        return getattr(torch, name)

    def empty(self, *args, **kwargs):
        out = torch.empty(*args, **kwargs)
        if seen.get('recording') and out.dim() == 1 and out.dtype == torch.float32:
            if 'work' not in seen or out.numel() > seen['work'].numel():
                seen['work'] = out
        return out


ops.torch = TorchShim()
original = ops.RenderCall.backward


def recording_backward(self, *args):
    seen['call'] = self
    seen['recording'] = True
    seen.pop('work', None)
    try:
        return original(self, *args)
    finally:
        seen['recording'] = False


ops.RenderCall.backward = recording_backward
model = t._model(precision, 'ctypes', kind)
small = t._batch(512)
ref_work, ref_grads = None, None
for i in range(repeats):
    out = model(small)
    model.zero_grad(set_to_none=True)
    t._loss(out).backward()
    torch.cuda.synchronize()
    call, work = seen['call'], seen['work']
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    if ref_work is None:
        ref_work, ref_grads = work.clone(), grads
        regions, at = {}, 0
        for l in sorted(call.levels):
            s = call.samples(l)
            inner = call.mlps[l].backward_workspace_floats(call.n, s)
            regions[l] = (at, call.n * s, inner)
            at += 4 * call.n * s + (inner + 63) // 64 * 64
        print('workspace', work.numel(), 'floats; regions', regions, 'end', at, flush=True)
        assert at == work.numel(), (at, work.numel())
        continue
    changed = sorted({n.split('.')[0] for n, g in grads.items() if not torch.equal(g, ref_grads[n])})
    if not changed:
        continue
    print('rep', i, 'gradients differ in', changed, flush=True)
    a, b = work.view(torch.int32), ref_work.view(torch.int32)
    for l, (at, samples, inner) in regions.items():
        for name, lo, hi in (('d_sigma', at, at + samples), ('d_rgb', at + samples, at + 4 * samples),
                             ('scratch', at + 4 * samples, at + 4 * samples + inner)):
            w = (a[lo:hi] != b[lo:hi]).nonzero().flatten()
            if w.numel():
                fa, fb = work[lo:hi][w[:4]].tolist(), ref_work[lo:hi][w[:4]].tolist()
                print('   level', l, name, 'words that differ', int(w.numel()), 'of', hi - lo, 'first', int(w[0]), 'last', int(w[-1]),
                      'values', fa, 'ref', fb, flush=True)

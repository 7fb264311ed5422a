#!/usr/bin/env python3
"""One rank's share of the strong-scaled config-5 iteration on a given build of the library (A/B in ONE lease: boxes of the pool differ by several per cent).  bench.py's own measurement (settle, warm-up, fenced
timed steps) through the ctypes binding, which loads `_lib.LIB_PATH`.
    usage: share_ab.py <lib.so> <rows per GPU> <parallel pack: 0|1> [single-pass: 0|1] [graphed: 0|1] [precision] [steps]
(`parallel pack` sets configs['model']['hip_parallel_pack'], the side-by-side re-pack of round 5's experiment: the key is no
longer read -- measured 5 % slower on the graphed 4096-row iteration, profiles/r05_share_ab.jsonl -- pass 0.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simplenerf_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from simplenerf_amd import synth  # noqa: E402

rows, parallel_pack = int(sys.argv[2]), bool(int(sys.argv[3]))
single_pass = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
graphed = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
precision = sys.argv[6] if len(sys.argv) > 6 else 'f16'
steps = int(sys.argv[7]) if len(sys.argv) > 7 else 30
_configs = synth.training_configs


def _ctypes_configs(*args, **kwargs):
    cfg = _configs(*args, **kwargs)
    cfg['model']['hip_host_binding'] = 'ctypes'
    cfg['model']['hip_parallel_pack'] = parallel_pack
    return cfg


synth.training_configs = _ctypes_configs
import bench  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
ms, fwd, bwd, n = bench.time_training(precision, dev, steps, 5, single_pass=single_pass, graphed=graphed, rows_per_gpu=rows)
print(json.dumps({'lib': os.path.basename(sys.argv[1]), 'rows': n, 'parallel_pack': parallel_pack, 'single_pass': single_pass, 'graphed': graphed,
                  'precision': precision, 'ms_per_iteration': round(ms, 4), 'step_ms_p50': round(bench.time_training.timing['step_ms']['p50'], 4)}))

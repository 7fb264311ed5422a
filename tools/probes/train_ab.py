#!/usr/bin/env python3
"""Config 5's training iteration on a given build of the library (A/B of diagnostic variants, tools/probes/build_variant.py):
bench.py's own measurement (settle, warm-up, fenced timed steps) through the ctypes binding, which loads `_lib.LIB_PATH`
(the TORCH_LIBRARY extension is linked against the shipped library).   usage: train_ab.py <lib.so> [precision] [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simplenerf_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from simplenerf_amd import synth  # noqa: E402

_configs = synth.training_configs


def _ctypes_configs(*args, **kwargs):
    cfg = _configs(*args, **kwargs)
    cfg['model']['hip_host_binding'] = 'ctypes'
    return cfg


synth.training_configs = _ctypes_configs
import bench  # noqa: E402

precision = sys.argv[2] if len(sys.argv) > 2 else 'f16'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
with bench.BoardSampler(0) as sampler:       # board power and shader clock during the run (sysfs, every 20 ms)
    ms, fwd, bwd, rows = bench.time_training(precision, dev, steps, 3)
print(json.dumps({'lib': os.path.basename(sys.argv[1]), 'precision': precision, 'ms_per_iteration': ms, 'mlp_forward_ms': fwd,
                  'mlp_backward_ms': bwd, 'step_ms_p50': bench.time_training.timing['step_ms']['p50'], 'board': sampler.summary()}))

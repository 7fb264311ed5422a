"""Per-workgroup duration of the LAST large weight-gradient launch of a config-5 iteration, from the in-kernel stamps of the
-DSNERF_CLOCK_STAMP build (tools/probes/build_variant.py clock -DSNERF_CLOCK_STAMP): which job of the launch ends last.
    python tools/probes/wgrad_wg_times.py gpurun_abl_clock.so [f16|f16s8|bf16]"""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from simplenerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
precision = sys.argv[2] if len(sys.argv) > 2 else 'f16s8'
from simplenerf_amd import harness, optim, synth
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
from simplenerf_amd.models.ModelFactory import get_model

dev = torch.device('cuda', 0)
lib = _lib.load()
cfg = synth.training_configs(precision, num_rays=2048, num_sparse=2048)
cfg['model']['hip_host_binding'] = 'ctypes'
model = get_model(cfg, None)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
model = model.to(dev).train()
batcher = BatchAssembler(cfg, synth.training_scene(), dev)
losses = LossComputer(cfg)
opt = optim.Adam(list(model.parameters()), lr=5e-4)
for it in range(20001, 20031):
    harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it), cfg['sub_batch_size'])
torch.cuda.synchronize()
fn = lib.snerf_debug_clock_stamps_wgrad16
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
pairs = 256
buf = (ctypes.c_ulonglong * (2 * pairs))()
assert fn(buf, pairs) == 0
us = [buf[2 * i + 1] * 0.01 for i in range(pairs)]          # 100 MHz reference ticks
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 8
live = [u for u in us if u > 1]
print(f'{precision}: {len(live)} workgroups stamped; all: min {min(live):.1f} median {statistics.median(live):.1f} max {max(live):.1f} us')
print('in launch order, 16 at a time (median us):', ' '.join(f'{statistics.median(us[i:i + 16]):.0f}' for i in range(0, pairs, 16)))

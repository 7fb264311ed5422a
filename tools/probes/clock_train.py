"""In-kernel clocks of the 16-bit training kernels ("say whether power or issue bounds what is left").
Runs BASELINE config 5's iteration (4096 rows, 16-bit mode) back to back for a few seconds on the -DSNERF_CLOCK_STAMP
diagnostic build (tools/probes/build_variant.py clock -DSNERF_CLOCK_STAMP), then reads, per stamped kernel, the median over
workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz from its last launch, next to the board's sysfs power and clock.
    python tools/probes/clock_train.py gpurun_abl_clock.so [f16|f16x3] [seconds]"""
import ctypes, glob, json, os, statistics, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from simplenerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
precision = sys.argv[2] if len(sys.argv) > 2 else 'f16'
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
import bench
from simplenerf_amd import harness, optim, synth
from simplenerf_amd.data_preprocessors.BatchAssembler01 import BatchAssembler
from simplenerf_amd.loss_functions.LossComputer01 import LossComputer
from simplenerf_amd.models.ModelFactory import get_model

dev = torch.device('cuda', 0)
lib = _lib.load()
cfg = synth.training_configs(precision, num_rays=2048, num_sparse=2048)
cfg['model']['hip_host_binding'] = 'ctypes'      # (a diagnostic build is only reachable through the ctypes binding)
model = get_model(cfg, None)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
model = model.to(dev).train()
batcher = BatchAssembler(cfg, synth.training_scene(), dev)
losses = LossComputer(cfg)
opt = optim.Adam(list(model.parameters()), lr=5e-4)
it = [20000]


def step():
    it[0] += 1
    harness.train_one_iter(model, losses, opt, batcher.get_next_batch(it[0]), cfg['sub_batch_size'])


def clock(kind, pairs=256):
    fn = getattr(lib, f'snerf_debug_clock_stamps_{kind}', None)
    if fn is None:
        return None
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    buf = (ctypes.c_ulonglong * (2 * pairs))()
    assert fn(buf, pairs) == 0
    ratios = [buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(pairs) if buf[2 * i + 1] > 100]
    return round(statistics.median(ratios), 4) if ratios else None


sampler = bench.BoardSampler(0)
with sampler:
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        n += 5
    elapsed = time.perf_counter() - t0
torch.cuda.synchronize()
out = {'precision': precision, 'ms_per_iteration': elapsed / n * 1e3, 'board': sampler.summary(),
       'in_kernel_clock_ghz': {k: clock(k) for k in ('forward_f16', 'chain_f16', 'wgrad16')}}
print(json.dumps(out))

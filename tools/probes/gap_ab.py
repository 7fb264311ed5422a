"""Do the microseconds between the kernels of the headline step cost throughput?  (a fused persistent
render kernel, or measured evidence that its gain is not there.)

Three ways of issuing the SAME step (1024 rays, 128+128, eval) on one GPU, in one process, alternated after a settle phase:
  events   eager, the library's timing events around every MLP launch (what bench.py times: ~5.7 us of idle per event pair)
  plain    eager, no events: kernels back to back in the queue
  graph    the whole step captured once as a HIP graph and replayed (no host work between kernels at all)
For each arm: wall time per step over `steps` steps (median of `rounds` rounds), and -- when the library is the
-DSNERF_CLOCK_STAMP diagnostic build (tools/probes/build_variant.py clock -DSNERF_CLOCK_STAMP) -- the clock the chip held
INSIDE the fused MLP kernel during that arm: median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz
(MI355X_MICROARCH.md, DVFS give-back 6; sysfs clocks read up to 10 % high).

    python tools/probes/gap_ab.py [f16|f16x3|fp32] [lib.so]        -> one JSON object per line
"""
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simplenerf_amd import _lib  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else 'f16'
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
import bench  # noqa: E402
from simplenerf_amd import harness, ops, synth  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
lib = _lib.load()
stamped = hasattr(lib, 'snerf_debug_clock_stamps_forward_m16')
configs = synth.make_configs('headline')
camera = synth.camera('fern', 0)
h, w = camera['resolution']
first = (h // 2) * w
# (the ctypes binding: the TORCH_LIBRARY extension is linked against the SHIPPED library, a diagnostic build is only reachable
# through _lib.LIB_PATH)
configs['model']['hip_host_binding'] = 'ctypes'
model = bench.synthetic_model(configs, 7, dev, precision)
STEPS, ROUNDS = 300, 5


def step():
    out = model(harness.frame_batch(camera, True, dev, first, 1024))
    return out['rgb_fine'], out['depth_fine']


def in_kernel_clock():
    """GHz inside the last fine-pass launch (1024 workgroups of 256 samples in the 16-bit mode), median over workgroups."""
    if not stamped:
        return None
    pairs = 512
    buf = (ctypes.c_ulonglong * (2 * pairs))()
    lib.snerf_debug_clock_stamps_forward_m16.restype = ctypes.c_int
    lib.snerf_debug_clock_stamps_forward_m16.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    torch.cuda.synchronize()
    assert lib.snerf_debug_clock_stamps_forward_m16(buf, pairs) == 0
    ratios = [buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(pairs) if buf[2 * i + 1] > 0]
    return statistics.median(ratios) if ratios else None


with torch.no_grad():
    bench.settle(step, 1.0)
    # the graph arm: capture one step
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out = step()
    eager_out = step()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out[0], eager_out[0]) and torch.equal(static_out[1], eager_out[1])

    def run(arm):
        if arm == 'events':
            ops.profile_enable(4 * STEPS + 16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if arm == 'graph':
            for _ in range(STEPS):
                graph.replay()
        else:
            for _ in range(STEPS):
                step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / STEPS * 1e3
        kernel_ms = None
        if arm == 'events':
            launches, _ = ops.profile_collect(ops.PROFILE_MLP_FORWARD)
            kernel_ms = sum(launches) / STEPS
            ops.profile_enable(0)
        return ms, kernel_ms, in_kernel_clock()

    results = {arm: [] for arm in ('events', 'plain', 'graph')}
    for _ in range(ROUNDS):
        for arm in results:
            results[arm].append(run(arm))
summary = {'precision': precision, 'library': os.path.basename(_lib.LIB_PATH), 'steps_per_round': STEPS, 'rounds': ROUNDS,
           'device': torch.cuda.get_device_name(0)}
for arm, rows in results.items():
    clocks = [r[2] for r in rows if r[2] is not None]
    kernel = [r[1] for r in rows if r[1] is not None]
    summary[arm] = {'ms_per_step_median': statistics.median(r[0] for r in rows), 'ms_per_step_all': [round(r[0], 4) for r in rows],
                    'mlp_kernel_ms_per_step': statistics.median(kernel) if kernel else None,
                    'in_kernel_clock_ghz': statistics.median(clocks) if clocks else None}
print(json.dumps(summary))

"""The headline step (1024 rays, 128+128) a few times WITHOUT the library's timing events, for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/probes/trace_step.py <precision>
(tools/probes/trace_step.sh does that for bench.py itself, events included, and prints the last step's timeline.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from simplenerf_amd import harness, synth
precision = sys.argv[1] if len(sys.argv) > 1 else 'f16'
dev = torch.device('cuda', 0)
configs = synth.make_configs('headline')
camera = synth.camera('fern', 0)
h, w = camera['resolution']
model = bench.synthetic_model(configs, 7, dev, precision)
with torch.no_grad():
    for _ in range(8):
        model(harness.frame_batch(camera, True, dev, (h // 2) * w, 1024))
torch.cuda.synchronize()

"""Duration of one fused-MLP launch against the number of resident workgroups (128 samples each): if a workgroup takes
much longer when all 256 CUs run than when a few do, the kernel is limited by something shared (L2 / fabric), not by
its own instruction stream.  usage: occupancy_sweep.py <precision: 0 fp32 | 1 f16x3>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplenerf_amd import ops, synth
from tests import util
from simplenerf_amd.synth import abi_param_list
prec = int(sys.argv[1])
cfg = synth.mlp_config(128)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 1)
mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()}))
for wgs in (1, 8, 32, 64, 128, 256, 512, 1024):
    n, s = wgs, 128
    o = torch.rand(n, 3, device='cuda'); d = torch.rand(n, 3, device='cuda'); v = d / d.norm(dim=1, keepdim=True)
    z = torch.sort(torch.rand(n, s, device='cuda'), 1)[0]
    for _ in range(3): mlp.forward(o, d, v, z, precision=prec)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): mlp.forward(o, d, v, z, precision=prec)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    rounds = max(1, -(-wgs // 256))
    print(f'{wgs:5d} workgroups: {ms*1e3:8.1f} us per launch, {ms*1e3/rounds:8.1f} us per round')

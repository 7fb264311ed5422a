"""Time one snerf_mlp_forward launch (fine-pass size) for a given library build.  usage: time_mlp.py <lib.so> <precision>"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simplenerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from simplenerf_amd import ops, synth
from tests import util
from simplenerf_amd.synth import abi_param_list
prec = int(sys.argv[2])
cfg = synth.mlp_config(128)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 1)
mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()}))
n, s = 1024, 256
o = torch.rand(n, 3, device='cuda'); d = torch.rand(n, 3, device='cuda'); v = d / d.norm(dim=1, keepdim=True)
z = torch.sort(torch.rand(n, s, device='cuda'), 1)[0]
def run():       # an ablation variant multiplies garbage: a launch that left the fp16 range makes the NEXT call fail before it enqueues
    try:
        mlp.forward(o, d, v, z, precision=prec)
        return 1
    except _lib.Fp16RangeError:
        return 0
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
done = sum(run() for _ in range(40))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / max(done, 1)
print(f'{sys.argv[1]}: {dt*1e3:.3f} ms  {n*s*2*593408/dt/1e12:.1f} TFLOP/s algorithmic')

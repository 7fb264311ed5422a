"""snerf_mlp_backward of the main 8x256 MLP at the two sizes a config-5 sub-batch runs it (2048 rays x 64 coarse / 192 merged fine
samples) for one or more library builds, alternated in rounds inside one process per build.
usage: time_backward_sizes.py <precision 0|1|2> <lib.so> [<lib.so> ...]   (one child process per library: a process binds one)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3 or (len(sys.argv) == 3 and os.environ.get('SNERF_CHILD') != '1'):
    for lib in sys.argv[2:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1], lib], env={**os.environ, 'SNERF_CHILD': '1'}, check=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from simplenerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[2])
from simplenerf_amd import ops, synth
from tests import util
from simplenerf_amd.synth import abi_param_list
prec = int(sys.argv[1])
cfg = synth.mlp_config(128)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 1)
plist = abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()})
mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(plist)
shapes = [tuple(p.shape) for p in plist]
cases = {}
for n, s in ((2048, 64), (2048, 192)):
    o = torch.rand(n, 3, device='cuda'); d = torch.rand(n, 3, device='cuda'); v = d / d.norm(dim=1, keepdim=True)
    z = torch.sort(torch.rand(n, s, device='cuda'), 1)[0]
    sigma, rgb, saved = mlp.forward_train(o, d, v, z, None, prec)
    cases[(n, s)] = (saved, sigma, rgb, torch.randn(n, s, 1, device='cuda'), torch.randn(n, s, 3, device='cuda'))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:       # settle
    for c in cases.values(): mlp.backward(*c, shapes, prec)
    torch.cuda.synchronize()
out = []
for key, c in cases.items():
    times = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): mlp.backward(*c, shapes, prec)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 20 * 1e3)
    out.append(f'{key[0]}x{key[1]}: {sorted(times)[2]:.3f} ms')
print(f'{os.path.basename(sys.argv[2])} precision {prec}: backward ' + ', '.join(out), flush=True)

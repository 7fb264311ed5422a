"""Time the training forward (activations kept) and the backward of the main 8x256 MLP at the fine-pass size 2048 x 192
for a given library build.  usage: time_train_mlp.py <lib.so> <precision>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ctypes
from simplenerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
# (an older build of the library lacks the newer entry points: bind what it has)
_probe = ctypes.CDLL(_lib.LIB_PATH)
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
_lib.ABI_VERSION = _probe.snerf_abi_version()
from simplenerf_amd import ops, synth
from tests import util
from simplenerf_amd.synth import abi_param_list
prec = int(sys.argv[2])
cfg = synth.mlp_config(128)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 1)
plist = abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()})
mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(plist)
n, s = 2048, 192
o = torch.rand(n, 3, device='cuda'); d = torch.rand(n, 3, device='cuda'); v = d / d.norm(dim=1, keepdim=True)
z = torch.sort(torch.rand(n, s, device='cuda'), 1)[0]
gs, gr = torch.randn(n, s, 1, device='cuda'), torch.randn(n, s, 3, device='cuda')
shapes = [tuple(p.shape) for p in plist]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:      # settle
    sigma, rgb, saved = mlp.forward_train(o, d, v, z, None, prec)
    mlp.backward(saved, sigma, rgb, gs, gr, shapes, prec)
    torch.cuda.synchronize()
fwds, bwds = [], []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): sigma, rgb, saved = mlp.forward_train(o, d, v, z, None, prec)
    torch.cuda.synchronize(); fwds.append((time.perf_counter() - t0) / 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): mlp.backward(saved, sigma, rgb, gs, gr, shapes, prec)
    torch.cuda.synchronize(); bwds.append((time.perf_counter() - t0) / 20)
fwd, bwd = sorted(fwds)[2], sorted(bwds)[2]
print(f'{os.path.basename(sys.argv[1])}: training forward {fwd*1e3:.3f} ms, backward {bwd*1e3:.3f} ms  ({n*s} samples)')

"""Gradients of snerf_mlp_backward for one seeded input with a given library build, written to a file: run once per build and
compare the files (are two builds' gradients bit-identical?).
usage: dump_backward.py <lib.so> <precision 0|1|2> <out.pt>   /   dump_backward.py --compare a.pt b.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
if sys.argv[1] == '--compare':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    worst = 0.0
    same = True
    for i, (x, y) in enumerate(zip(a, b)):
        eq = torch.equal(x, y)
        same &= eq
        if not eq:
            rel = float((x - y).abs().max() / x.abs().max().clamp_min(1e-30))
            worst = max(worst, rel)
            print(f'tensor {i} {tuple(x.shape)}: differs, max |a-b| / max |a| = {rel:.3e}')
    print('bit-identical' if same else f'NOT identical (worst relative difference {worst:.3e})')
    sys.exit(0)
from simplenerf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from simplenerf_amd import ops, synth
from tests import util
from simplenerf_amd.synth import abi_param_list
prec = int(sys.argv[2])
torch.manual_seed(3)
cfg = synth.mlp_config(128)
sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 1)
plist = abi_param_list({k: torch.from_numpy(v).cuda() for k, v in sd.items()})
mlp = ops.PackedMlp(cfg, 'cuda:0'); mlp.pack(plist)
shapes = [tuple(p.shape) for p in plist]
n, s = 1000, 67       # not a multiple of the workgroup's 256 samples
o = torch.rand(n, 3, device='cuda'); d = torch.rand(n, 3, device='cuda'); v = d / d.norm(dim=1, keepdim=True)
z = torch.sort(torch.rand(n, s, device='cuda'), 1)[0]
sigma, rgb, saved = mlp.forward_train(o, d, v, z, None, prec)
grads = mlp.backward(saved, sigma, rgb, torch.randn(n, s, 1, device='cuda') * 1e-4, torch.randn(n, s, 3, device='cuda') * 1e-4, shapes, prec)
torch.cuda.synchronize()
torch.save([g.cpu() for g in grads] + [sigma.cpu(), rgb.cpu()], sys.argv[3])
print('wrote', sys.argv[3])

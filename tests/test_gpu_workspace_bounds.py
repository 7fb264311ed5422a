"""The backward's scratch stays inside what snerf_mlp_backward_workspace_floats() reports (include/simplenerf_hip.h:170-187).
Levels that run side by side (render.hip) sit back to back in ONE workspace: a write past a level's share lands in its
neighbour's gradient columns while that neighbour's kernels read them."""
import pytest
import torch

from simplenerf_amd import ops
from tests.test_gpu_f16 import abi_param_list, mlp_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GUARD = 1 << 20            # floats on either side of the scratch
PATTERN = 0x7fa5a5a5       # a NaN payload no kernel produces


@pytest.mark.parametrize('precision', ['fp32', 'f16x3', 'f16', 'bf16', 'f16s8', 'bf16s8'])
@pytest.mark.parametrize('layout,size', [('main', (8, 256, 128)), ('ptsaug', (8, 256, 128)), ('viewsaug', (8, 256, 128)),
                                         ('main', (4, 128, 64))])
@pytest.mark.parametrize('n,s', [(7, 45), (512, 192), (512, 64), (300, 131)])
def test_backward_scratch_stays_inside_the_reported_size(precision, layout, size, n, s):
    cfg, sd, inputs, (g_sigma, g_rgb) = mlp_case(layout, size, n, s)
    plist = abi_param_list({k: torch.from_numpy(v).to(DEV) for k, v in sd.items()})
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(plist)
    prec = ops.PRECISIONS[precision]
    sigma, rgb, saved = mlp.forward_train(*[t.to(DEV) for t in inputs], prec)
    need = mlp.backward_workspace_floats(n, s)
    assert need > 0
    arena = torch.full((GUARD + need + GUARD,), PATTERN, dtype=torch.int32, device=DEV)
    work = arena[GUARD:GUARD + need].view(torch.float32)
    shapes = [tuple(p.shape) for p in plist]
    grads = mlp.backward(saved, sigma, rgb, g_sigma.to(DEV), g_rgb.to(DEV), shapes, prec, work=work)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(g).all()) for g in grads)
    before = (arena[:GUARD] != PATTERN).nonzero()
    after = (arena[GUARD + need:] != PATTERN).nonzero()
    assert before.numel() == 0, f'{before.numel()} words written in front of the scratch, first at -{GUARD - int(before.min())}'
    assert after.numel() == 0, (f'{after.numel()} words written behind the scratch of {need} floats: '
                                f'+{int(after.min())} .. +{int(after.max())}')

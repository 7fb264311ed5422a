"""The renderer gives the same bits with another stream's matrix kernels running beside it (before round 5 the
library held a packed fp32 instruction form that miscomputes beside a kernel issuing MFMAs -- in the compositing backward, a loss
kernel and the 16 x 16 x 32 rendering kernel's encoding).  What this guards is the arrangement -- render calls on one stream, MLP
launches on another, bit-equal to the render alone; the instruction form itself is refused by the build's scan
(tests/test_host_logic.py) and watched on the hardware by tests/test_gpu_packed_forms.py.  (The pre-fix rendering kernel passes
this test too: its workgroups fill a CU's LDS and a neighbour seldom shares a SIMD with them; where the form did bite was the
small compositing backward beside MLP backwards, tests/test_gpu_side_by_side.py.)"""
import pytest
import torch

from simplenerf_amd import harness, ops, synth
from tests.test_gpu_f16 import abi_param_list, mlp_case, synthetic_model

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class _Neighbour:
    """A storing MLP forward (MFMAs + a memory stream) launched over and over on a stream of its own."""

    def __init__(self, precision):
        cfg, sd, inputs, _ = mlp_case('main', (8, 256, 128), 512, 192)
        self.mlp = ops.PackedMlp(cfg, DEV)
        self.mlp.pack(abi_param_list({k: torch.from_numpy(v).to(DEV) for k, v in sd.items()}))
        self.inputs = [t.to(DEV) for t in inputs]
        self.precision = ops.PRECISIONS[precision]
        self.stream = torch.cuda.Stream()
        torch.cuda.synchronize()

    def enqueue(self, launches):
        with torch.cuda.stream(self.stream):
            for _ in range(launches):
                self.mlp.forward_train(*self.inputs, self.precision)


@pytest.mark.parametrize('precision', ['fp32', 'f16x3', 'f16', 'bf16'])
def test_rendering_beside_another_streams_matrix_kernels_gives_the_same_bits(precision):
    cfg = synth.make_configs('headline')
    model = synthetic_model(cfg, precision).eval()
    batch = harness.frame_batch(synth.camera('fern', 0), True, DEV, 95000, 1024)
    keys = ('rgb_coarse', 'rgb_fine', 'depth_ndc_coarse', 'depth_ndc_fine')
    with torch.no_grad():
        ref = {k: v.clone() for k, v in model(batch).items() if k in keys}
        torch.cuda.synchronize()
        neighbour = _Neighbour('bf16' if precision == 'fp32' else precision)
        side = torch.cuda.Stream()
        for _ in range(12):
            neighbour.enqueue(6)                      # ~2 ms of matrix work in flight beside the render
            with torch.cuda.stream(side):
                out = model(batch)
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(out[k], ref[k]), k

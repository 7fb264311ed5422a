"""Seeded sweep of ragged shapes and random configurations through the C ABI against the CPU oracle: fused MLP (both
precisions, three weight layouts, two sizes, world/NDC-like point ranges, with and without density noise), compositing
(NDC / world / white background) and resampling.  The fixtures pin the oracle to the reference; this widens the set of
shapes the HIP kernels are held to the oracle on (sample counts that are not multiples of the 32/128-sample tiles,
single rays, single samples)."""
import numpy
import pytest
import torch

from oracle import nerf_oracle as oracle
from simplenerf_amd import ops, synth
from tests import util
from tests.test_gpu_kernels import LAYOUTS, abi_param_list

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = list(range(10))


@pytest.mark.parametrize('case', CASES)
def test_fused_mlp_random_shapes(case):
    rng = numpy.random.RandomState(1000 + case)
    layout = ['main', 'ptsaug', 'viewsaug'][case % 3]
    depth, width, vwidth = [(8, 256, 128), (4, 128, 64), (2, 128, 64), (6, 256, 128), (1, 256, 128)][case % 5]
    cfg = synth.mlp_config(64, depth=depth, width=width, views_width=vwidth, **LAYOUTS[layout])
    sd = synth.synth_state_dict(util.mlp_param_shapes(cfg), 200 + case, float(rng.choice([1.0, 30.0, 200.0])), float(rng.uniform(-3, 3)))
    n, s = int(rng.randint(1, 90)), int(rng.choice([1, 2, 31, 33, 64, 127, 129, 192]))
    spread = float(rng.choice([1.0, 6.0]))      # NDC-like unit cube or world-space extents
    o = torch.from_numpy(rng.uniform(-spread, spread, (n, 3)).astype(numpy.float32))
    d = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(numpy.float32))
    v = d / d.norm(dim=1, keepdim=True)
    z = torch.from_numpy(numpy.sort(rng.uniform(0, 1, (n, s)).astype(numpy.float32), axis=1))
    noise = torch.from_numpy(rng.standard_normal((n, s, 1)).astype(numpy.float32)) if case % 2 else None
    params = {k: torch.from_numpy(a) for k, a in sd.items()}
    ref = oracle.run_mlp(params, '', cfg, oracle.ray_points(o, d, z), v, None, noise)
    mlp = ops.PackedMlp(cfg, DEV)
    mlp.pack(abi_param_list({k: a.to(DEV) for k, a in params.items()}))
    for precision in ('fp32', 'f16x3'):
        sigma, rgb = mlp.forward(o.to(DEV), d.to(DEV), v.to(DEV), z.to(DEV), None if noise is None else noise.to(DEV),
                                 ops.PRECISIONS[precision])
        scale = max(1.0, float(ref['sigma'].abs().max()))
        assert util.linf(sigma, ref['sigma']) <= 2e-5 * scale, (precision, layout, depth, width, n, s)
        assert util.linf(rgb, ref['rgb']) <= 1e-5, (precision, layout, depth, width, n, s)


@pytest.mark.parametrize('case', CASES)
def test_composite_and_resample_random_shapes(case):
    rng = numpy.random.RandomState(2000 + case)
    n, s = int(rng.randint(1, 70)), int(rng.choice([2, 3, 17, 63, 64, 65, 128, 191, 256]))
    ndc, white = bool(case % 2), bool(case % 3 == 0)
    lo, hi = (0.0, 1.0) if ndc else (2.0, 6.0)
    z = torch.from_numpy(numpy.sort(rng.uniform(lo, hi, (n, s)).astype(numpy.float32), axis=1))
    sigma = torch.from_numpy((rng.gamma(0.5, 6.0, (n, s)) * (rng.uniform(size=(n, s)) > 0.3)).astype(numpy.float32))
    rgb = torch.from_numpy(rng.uniform(0, 1, (n, s, 3)).astype(numpy.float32))
    march = torch.from_numpy(rng.standard_normal((n, 3)).astype(numpy.float32))
    rays_o = torch.from_numpy((0.1 * rng.standard_normal((n, 3))).astype(numpy.float32)) if ndc else None
    rays_d = torch.from_numpy(numpy.c_[rng.uniform(-0.3, 0.3, (n, 2)), -numpy.ones(n)].astype(numpy.float32)) if ndc else None
    ref = oracle.composite(sigma, rgb, z, march, ndc, white, rays_o, rays_d)
    t = lambda a: None if a is None else a.to(DEV)
    got = ops.composite(t(sigma), t(rgb), t(z), t(march), ndc, white, t(rays_o), t(rays_d))
    assert sorted(got) == sorted(ref)
    for k, r in ref.items():
        assert util.linf(got[k], r) <= 2e-5 * max(1.0, float(r.abs().max())), (k, n, s, ndc, white)
    if s >= 3:
        s_f = int(rng.choice([1, 5, 64, 128]))
        w = got['weights'].cpu()
        u = torch.from_numpy(rng.uniform(0, 1, (n, s_f)).astype(numpy.float32)) if case % 2 else None
        zr = oracle.resample_depths(z, w, s_f, u)
        zg = ops.resample_depths(t(z), t(w), s_f, t(u)).cpu()
        assert zg.shape == zr.shape and bool((zg[:, 1:] >= zg[:, :-1]).all())
        assert util.outlier_fraction(zg, zr, 2e-6 * max(1.0, hi)) <= 0.02, (n, s, s_f)

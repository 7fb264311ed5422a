#!/usr/bin/env python3
"""Time the other BASELINE.json configurations on one MI355X (not bench lines; numbers quoted in DESIGN.md).
    python tools/measure_configs.py > gpurun_out/configs.json        (SNERF_PREC=f16x3 for the split-precision kernels, f16 for the 16-bit mode)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplenerf_amd import harness, synth  # noqa: E402
from simplenerf_amd.models.ModelFactory import get_model  # noqa: E402

DEV = torch.device('cuda', 0)
FLOP = {'main': 2 * 593408, 'ptsaug': 2 * 577280, 'viewsaug': 2 * 492032, 'small': 2 * 83840}


def model_for(kind, train=False):
    cfg = synth.with_overrides(synth.make_configs(kind), hip_precision=os.environ.get('SNERF_PREC', 'fp32'))
    m = get_model(cfg, None)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    return cfg, m.to(DEV).train(train)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    res = []
    with torch.no_grad():
        # config 1: 1024 random world rays, 64 coarse, 4x128
        cfg, m = model_for('config1')
        batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.random_world_rays(1024).items()}
        dt = timed(lambda: m(batch), 50)
        res.append({'config': 'config1: 1024 rays, 64 coarse, 4x128, world', 'ms': dt * 1e3, 'rays_per_s': 1024 / dt,
                    'tflops': 1024 * 64 * FLOP['small'] / dt / 1e12})
        # config 2: fern 504x378 (named) and 1008x756 (reference-native), 64+128
        cfg, m = model_for('config2')
        for down in (2, 1):
            cam = synth.camera('fern', 0, downscale=down)
            h, w = cam['resolution']
            dt = timed(lambda: harness.render_frame(m, cam, True, DEV), 2 if down == 1 else 4)
            res.append({'config': f'config2: fern {w}x{h}, 64+128, 8x256 coarse+fine, NDC, full frame incl. raygen',
                        'ms': dt * 1e3, 'rays_per_s': h * w / dt, 'tflops': h * w * 256 * FLOP['main'] / dt / 1e12})
        # config 4 (single GPU part): RE10K camera at 1008x756 named / 1024x576 native
        cam = synth.camera('re10k', 0)
        h, w = cam['resolution']
        dt = timed(lambda: harness.render_frame(m, cam, True, DEV), 2)
        res.append({'config': f'config4 (1 GPU): re10k {w}x{h}, 64+128, full frame', 'ms': dt * 1e3,
                    'rays_per_s': h * w / dt, 'tflops': h * w * 256 * FLOP['main'] / dt / 1e12})
        # config 3: training-mode forward, 4096 rays, both augmented MLPs, device RNG
        cfg, m = model_for('config3', train=True)
        cam = synth.camera('fern', 0)
        batch = harness.frame_batch(cam, True, DEV, 300000, 4096)
        dt = timed(lambda: m(batch), 5)
        flop = 4096 * (64 * (FLOP['main'] + FLOP['ptsaug'] + FLOP['viewsaug']) + 192 * FLOP['main'])
        res.append({'config': 'config3: train-mode forward, 4096 rays, main c+f + points-aug + views-aug, perturb + noise',
                    'ms': dt * 1e3, 'rays_per_s': 4096 / dt, 'tflops': flop / dt / 1e12})
    for r in res:
        r['precision'] = os.environ.get('SNERF_PREC', 'fp32')
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

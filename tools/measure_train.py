#!/usr/bin/env python3
"""Config 5 on one MI355X: 4096-ray training step (2 x 2048-ray sub-batches with gradient accumulation, like
Trainer.train_one_iter src/Trainer01.py:61-107), forward + backward through all four MLPs, Adam step.
    python tools/measure_train.py > gpurun_out/train.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplenerf_amd import harness, synth  # noqa: E402
from simplenerf_amd.models.ModelFactory import get_model  # noqa: E402

DEV = torch.device('cuda', 0)
FLOP = {'main': 2 * 593408, 'ptsaug': 2 * 577280, 'viewsaug': 2 * 492032}


def loss_fn(out):
    loss = 0.
    for k in out:
        base = k.replace('points_augmentation_', '').replace('views_augmentation_', '')
        if base in ('rgb_coarse', 'rgb_fine'):
            loss = loss + (out[k] ** 2).mean()
        elif base in ('depth_coarse', 'depth_fine'):
            loss = loss + 0.01 * (out[k] ** 2).mean()
    return loss


def main():
    cfg = synth.make_configs('config3')
    model = get_model(synth.with_overrides(cfg, hip_precision=os.environ.get("SNERF_PREC", "fp32")), None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 7, 200.0, 8.0).items()})
    model = model.to(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
    cam = synth.camera('fern', 0)
    subs = [harness.frame_batch(cam, True, DEV, 200000 + i * 2048, 2048) for i in range(2)]

    def step():
        opt.zero_grad(set_to_none=True)
        for b in subs:
            loss_fn(model(b)).backward()
        opt.step()

    def fwd_only():
        with torch.no_grad():
            for b in subs:
                model(b)

    res = {}
    for name, fn, reps in (('train_step_fwd_bwd_adam', step, 5), ('train_mode_forward_only', fwd_only, 5)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        fwd_flop = 4096 * (64 * (FLOP['main'] + FLOP['ptsaug'] + FLOP['viewsaug']) + 192 * FLOP['main'])
        mult = 3 if 'bwd' in name else 1
        res[name] = {'ms': dt * 1e3, 'rays_per_s': 4096 / dt, 'algorithmic_tflops': mult * fwd_flop / dt / 1e12}
    res['peak_memory_gb'] = torch.cuda.max_memory_allocated() / 2 ** 30
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

"""CPU restatement (numpy fp32) of the optimiser step and learning-rate schedules -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
Pinned: ``tests/test_oracle_golden.py`` holds it bit-for-bit to ``tests/golden/optim_adam.npz``, produced by
``tools/make_golden_optim.py`` running ``torch.optim.Adam`` (CPU) and the reference's decayer classes.

The reference's trainer builds ``torch.optim.Adam(params, lr, betas)`` (src/Trainer01.py:516-517), overwrites
``param_groups[..]['lr']`` with the decayed rate before each iteration (:293-295) and calls ``step()`` (:102).  The
algorithm restated is PyTorch 2.x's single-tensor Adam (torch/optim/adam.py ``_single_tensor_adam``: lerp_, mul_ +
addcmul_, sqrt / bias_correction2_sqrt + eps, addcdiv_); the element-wise rounding sequence below was matched bit
for bit against it (fused multiply-adds where ATen's vectorised CPU kernels fuse, separate roundings elsewhere).
"""
from __future__ import annotations

import math
from typing import List

import numpy

f32 = numpy.float32


def _fma(a, b, c):
    """fp32 fused multiply-add: the double product of two floats is exact and the double sum rounds once more to
    float -- double rounding can differ from a true fma only in astronomically rare ties, none in the fixtures."""
    return (numpy.asarray(a, numpy.float64) * numpy.asarray(b, numpy.float64) + numpy.asarray(c, numpy.float64)).astype(f32)


def adam_step(params: List[numpy.ndarray], grads: List[numpy.ndarray], exp_avg: List[numpy.ndarray],
              exp_avg_sq: List[numpy.ndarray], step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999,
              eps: float = 1e-8) -> None:
    """One Adam update in place; ``step`` is the 1-based count after the increment (adam.py: step_t += 1 first)."""
    bias1 = 1 - beta1 ** step
    bias2 = 1 - beta2 ** step
    neg_step_size = f32(-(lr / bias1))
    bias2_sqrt = f32(bias2 ** 0.5)
    with numpy.errstate(under='ignore', over='ignore', invalid='ignore', divide='ignore'):
        for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
            if g is None:
                continue
            m[...] = _fma(f32(1 - beta1), (g - m).astype(f32), m)                      # exp_avg.lerp_(grad, 1-beta1)
            v[...] = _fma((f32(1 - beta2) * g).astype(f32), g, (v * f32(beta2)).astype(f32))   # mul_().addcmul_()
            denom = ((numpy.sqrt(v).astype(f32) / bias2_sqrt).astype(f32) + f32(eps)).astype(f32)
            p[...] = (p + ((neg_step_size * m).astype(f32) / denom).astype(f32)).astype(f32)   # addcdiv_


def nerf_learning_rate(lr_initial: float, lr_decay: float, iter_num: int) -> float:
    """NeRFLearningRateDecayer.get_updated_learning_rate (src/lr_decayers/NeRFLearningRateDecayer01.py:15-24)."""
    return lr_initial * (0.1 ** (iter_num / (lr_decay * 1000)))


def mipnerf_learning_rate(lr_initial: float, lr_final: float, num_iterations: int, lr_decay_steps: int,
                          lr_decay_mult: float, iter_num: int) -> float:
    """MipNeRFLearningRateDecayer.get_updated_learning_rate (src/lr_decayers/MipNeRFLearningRateDecayer01.py:26-35):
    log-linear interpolation from lr_initial to lr_final with a sine warm-up factor."""
    # numpy's exp / sin / log, as the reference: they differ from libm's by 1 ulp on ~4 % of iterations
    warm = 1.0
    if lr_decay_steps > 0:
        warm = lr_decay_mult + (1 - lr_decay_mult) * numpy.sin(0.5 * numpy.pi * numpy.clip(iter_num / lr_decay_steps, 0, 1))
    t = numpy.clip(iter_num / num_iterations, 0, 1)
    return float(warm * numpy.exp(numpy.log(lr_initial) * (1 - t) + numpy.log(lr_final) * t))

"""Learning-rate schedules of the reference's trainer, evaluated on the host in double precision once per iteration
(the value is written into ``optimizer.param_groups[..]['lr']``, src/Trainer01.py:293-295).

Both follow the arithmetic of the reference's classes operation for operation -- including WHICH library evaluates the
transcendental functions (the reference's log-linear schedule uses numpy's exp / sin / log, which differ from libm's by
1 ulp on ~4 % of iterations) -- so the rates are the same doubles (pinned by tests/golden/optim_adam.npz and, densely,
tests/golden/lr_schedules.npz):
  ExponentialDecay  -- src/lr_decayers/NeRFLearningRateDecayer01.py:15-24
  LogLinearDecay    -- src/lr_decayers/MipNeRFLearningRateDecayer01.py:16-35
Each is constructed from the experiment dictionary and exposes ``get_updated_learning_rate(iter_num)``.
"""
import numpy


class ExponentialDecay:
    """rate(i) = lr_initial * 0.1 ** (i / (1000 * lr_decay)): one decade every ``lr_decay`` thousand iterations."""
    decade = 0.1

    def __init__(self, configs: dict):
        opt = configs['optimizer']
        self.configs = configs
        self.start, self.iterations_per_decade = opt['lr_initial'], opt['lr_decay'] * 1000

    def get_updated_learning_rate(self, iter_num):
        return self.start * (self.decade ** (iter_num / self.iterations_per_decade))


class LogLinearDecay:
    """rate(i) = warm(i) * exp((1-t) log lr_initial + t log lr_final), t = i / num_iterations clipped to [0,1];
    warm(i) ramps from lr_decay_mult to 1 along a quarter sine over the first lr_decay_steps iterations."""

    def __init__(self, configs: dict):
        opt = configs['optimizer']
        self.configs = configs
        self.log_start, self.log_end = numpy.log(opt['lr_initial']), numpy.log(opt['lr_final'])
        self.horizon = configs['num_iterations']
        self.warm_steps, self.warm_floor = opt['lr_decay_steps'], opt['lr_decay_mult']

    def get_updated_learning_rate(self, iter_num):
        warm = 1.0
        if self.warm_steps > 0:
            warm = self.warm_floor + (1 - self.warm_floor) * numpy.sin(0.5 * numpy.pi * numpy.clip(iter_num / self.warm_steps, 0, 1))
        t = numpy.clip(iter_num / self.horizon, 0, 1)
        return float(warm * numpy.exp(self.log_start * (1 - t) + self.log_end * t))

"""Log-linear schedule with sine warm-up (src/lr_decayers/MipNeRFLearningRateDecayer01.py:16-35); host-side doubles."""
import math


class MipNeRFLearningRateDecayer:
    def __init__(self, configs: dict):
        self.configs = configs
        opt = configs['optimizer']
        self.lr_init, self.lr_final = opt['lr_initial'], opt['lr_final']
        self.num_iters = configs['num_iterations']
        self.lr_decay_steps, self.lr_decay_mult = opt['lr_decay_steps'], opt['lr_decay_mult']

    def get_updated_learning_rate(self, iter_num):
        warm = 1.0
        if self.lr_decay_steps > 0:
            ramp = min(max(iter_num / self.lr_decay_steps, 0), 1)
            warm = self.lr_decay_mult + (1 - self.lr_decay_mult) * math.sin(0.5 * math.pi * ramp)
        t = min(max(iter_num / self.num_iters, 0), 1)
        return warm * math.exp(math.log(self.lr_init) * (1 - t) + math.log(self.lr_final) * t)

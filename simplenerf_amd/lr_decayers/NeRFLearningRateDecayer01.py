"""Exponential schedule of the shipped experiments: lr_initial * 0.1 ** (iter / (lr_decay * 1000)), evaluated in
double on the host exactly as the reference does (src/lr_decayers/NeRFLearningRateDecayer01.py:14-24)."""


class NeRFLearningRateDecayer:
    def __init__(self, configs: dict):
        self.configs = configs
        self.lr_init = configs['optimizer']['lr_initial']
        self.decay_rate = 0.1
        self.decay_steps = configs['optimizer']['lr_decay'] * 1000

    def get_updated_learning_rate(self, iter_num):
        return self.lr_init * (self.decay_rate ** (iter_num / self.decay_steps))

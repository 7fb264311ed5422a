"""Name binding for ``configs['optimizer']['lr_decayer_name'] = 'NeRFLearningRateDecayer01'`` (see schedules.py)."""
from .schedules import ExponentialDecay as NeRFLearningRateDecayer  # noqa: F401

"""Name binding for ``configs['optimizer']['lr_decayer_name'] = 'MipNeRFLearningRateDecayer01'`` (see schedules.py)."""
from .schedules import LogLinearDecay as MipNeRFLearningRateDecayer  # noqa: F401

"""``get_lr_decayer(configs)`` with the reference's naming rule (src/lr_decayers/LearningRateDecayerFactory.py:12-24):
``configs['optimizer']['lr_decayer_name']`` = module name, class name = module name minus the two-digit suffix."""
import importlib


def get_lr_decayer(configs: dict):
    filename = configs['optimizer']['lr_decayer_name']
    try:
        module = importlib.import_module(f'{__package__}.{filename}')
        return getattr(module, filename[:-2])(configs)
    except (ImportError, AttributeError):
        raise RuntimeError(f'Unknown lr decayer: {filename}') from None

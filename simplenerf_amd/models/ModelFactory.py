"""Name -> model plugin factory with the reference's rule (src/models/ModelFactory.py:10-22):
``configs['model']['name']`` is the module file stem, the class is the stem minus its two-digit version suffix, and
the constructor takes ``(configs, model_configs)``.

``'SimpleNeRF01'`` (the reference's own renderer name) is accepted as an alias of ``'SimpleNeRFHip01'`` so that an
unchanged experiment config selects the MI355X renderer when this package's ``models`` is the one on the path.
"""
import importlib
import inspect

ALIASES = {'SimpleNeRF01': 'SimpleNeRFHip01'}


def get_model(configs: dict, model_configs: dict = None):
    filename = ALIASES.get(configs['model']['name'], configs['model']['name'])
    classname = filename[:-2]
    try:
        module = importlib.import_module(f'{__package__}.{filename}')
    except ModuleNotFoundError as e:
        raise RuntimeError(f'Unknown model: {filename}') from e
    for name, cls in inspect.getmembers(module, inspect.isclass):
        if name == classname:
            return cls(configs, model_configs)
    raise RuntimeError(f'Unknown model: {filename}')

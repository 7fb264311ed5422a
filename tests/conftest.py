import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(REPO, 'tests', 'golden')


def pytest_terminal_summary(terminalreporter):
    """Observed parity figures next to their gates (tests/util.observe), printed whatever the verbosity."""
    from tests import util
    if util.OBSERVED:
        terminalreporter.write_sep('-', 'observed parity figures (gate in brackets)')
        for tag, text in util.OBSERVED:
            terminalreporter.write_line(f'{tag}: {text}')


@pytest.fixture(autouse=True)
def _clean_fp16_range_flag(request):
    """A GPU test that provokes (or dies with) an fp16 range violation must not leave the device's sticky flag set for the
    next test: cleared after every GPU test."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        try:
            import torch
            if torch.cuda.is_available():
                from simplenerf_amd import ops
                torch.cuda.synchronize()
                ops.range_status(clear=True)
        except Exception:
            pass

import importlib
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must never silently pass without a device
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def qt():
    """The package directory is literally `qa-tiger_amd/` (hyphenated), so it is
    imported through importlib; it registers itself as `qa_tiger_amd` too."""
    return importlib.import_module('qa-tiger_amd')


@pytest.fixture(scope='session')
def synth():
    return importlib.import_module('qa-tiger_amd.synth')


@pytest.fixture(scope='session')
def oracle():
    return importlib.import_module('oracle.qatiger_oracle')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f'{name}.npz')))


GOLDEN_CONFIGS = {
    # name -> (synth attribute or kwargs override, seed)
    'toy': ('CONFIG_TOY', None, 713),
    'toy_kk': ('CONFIG_TOY_KK', None, 11),
    'config1': ('CONFIG_1', None, 713),
    'config2_b8': ('CONFIG_2', {'B': 8}, 713),
    'config4_b2': ('CONFIG_4', {'B': 2}, 713),      # long-video stress shape: T=300, D=1024, K=16
}


def golden_cfg(synth, name):
    attr, override, seed = GOLDEN_CONFIGS[name]
    cfg = getattr(synth, attr)
    if override:
        cfg = synth.PathConfig(**{**cfg.asdict(), **override})
    return cfg, seed


def rel_err(a, b):
    """max|a-b| / max|b|  — the logit parity metric of SURVEY.md §8d."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

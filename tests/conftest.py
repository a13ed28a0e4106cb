import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


# ---- the library's test hooks: ONE environment variable, BQ_TEST_HOOKS="name=value,..." (csrc/bq_common.h) ----------------------
def _hooks_now():
    return dict(item.split('=', 1) for item in os.environ.get('BQ_TEST_HOOKS', '').split(',') if '=' in item)


def _hooks_str(cur):
    return ','.join(f'{k}={v}' for k, v in cur.items())


def set_hooks(monkeypatch, **kw):
    """set_hooks(monkeypatch, as_schur_min=0, as_schur=None): set / remove (None) hooks for the rest of the test."""
    cur = _hooks_now()
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv('BQ_TEST_HOOKS', _hooks_str(cur))
    else:
        monkeypatch.delenv('BQ_TEST_HOOKS', raising=False)


def hooks_env(**kw):
    """{'BQ_TEST_HOOKS': ...} for a child process: the hooks of this process + kw."""
    cur = _hooks_now()
    cur.update({k: str(v) for k, v in kw.items() if v is not None})
    return {'BQ_TEST_HOOKS': _hooks_str(cur)}


def hook_value(name, default=None):
    return _hooks_now().get(name, default)

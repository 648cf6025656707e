import os
import subprocess
import sys
import tempfile

import pytest

# MIOpen stores the result of its per-layer solver search in a USER find-db under $HOME and reuses
# it in later processes. The search depends on process-wide switches (e.g. deterministic mode), so a
# test process must never share that database with a benchmark run on the same box: give the test
# session a private, throw-away one. (Observed: a find recorded under cudnn.deterministic=True made
# every later PreResNet-20 training step on that box 8x slower.)
os.environ['MIOPEN_USER_DB_PATH'] = tempfile.mkdtemp(prefix='ursa_test_miopen_')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the oracle is test infrastructure: (re)build it if its source is newer than the .so
    so, src = os.path.join(ROOT, 'oracle', 'liboracle.so'), os.path.join(ROOT, 'oracle', 'ursa_oracle.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True, capture_output=True)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import __graft_entry__ as ge  # noqa: E402

ge.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "golden_v1.npz")


@pytest.fixture(scope="session")
def weights1():
    """One-layer synthetic model (same seed as tests/golden/gen_golden.py)."""
    from nemotron_asr_amd import synth
    return synth.make_weights(n_layers=1)


@pytest.fixture(scope="session")
def weights2():
    from nemotron_asr_amd import synth
    return synth.make_weights(n_layers=2)

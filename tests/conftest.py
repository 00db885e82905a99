import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    g = Path(__file__).resolve().parent / "golden"
    arrays = np.load(g / "golden.npz")
    meta = json.loads((g / "golden.json").read_text())
    return arrays, meta


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def iq10(oracle):
    return oracle.modulate(oracle.bert_frames(10))


@pytest.fixture(scope="session")
def iq100(oracle):
    return oracle.modulate(oracle.bert_frames(100))

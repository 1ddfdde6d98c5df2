import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as _oracle
    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def client():
    """ComputeClient on device 0 -- fails loudly (no skip) when the HIP library or GPU is missing."""
    from cubecl_amd import Mi355Runtime
    return Mi355Runtime.client()

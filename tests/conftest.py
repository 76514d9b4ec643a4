"""pytest configuration: marker registration + import paths.

`-m "not gpu"`: oracle vs golden fixtures, host logic, C-ABI symbol export (no compute).
`-m gpu`     : HIP path (through the C-ABI) vs the oracle / fixtures on a real MI355X.
"""
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "tests", REPO / "multimodal-sae_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

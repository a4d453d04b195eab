import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import s2k_oracle

    return s2k_oracle.get()


@pytest.fixture(scope="session")
def ecoli():
    return open(os.path.join(GOLD, "ecoli.genome.100k.fa")).read().split("\n")[1].encode()

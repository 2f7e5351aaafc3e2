import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/_build/liboracle.so (the CPU checker), built on demand"""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def oracle_1m(oracle):
    """(text, oracle result) of 1 M x 150 bp config-3 reads, k = 31, abundance-min 2: the largest size the oracle
    finishes in seconds; shared by the GPU tests that need it"""
    text = oracle.synth_reads(1000000, 150, 3)
    return text, oracle.run(text, 31, 2)

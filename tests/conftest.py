import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build(ref=os.path.exists("/root/reference/doc/filters004.txt"))
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    from groove_amd import entities as E
    ctx = E.Context(0)  # raises loudly if libgroove_hip.so or the GPU is missing
    yield ctx
    ctx.close()

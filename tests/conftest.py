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
    # the segment guard of the Welsh kernels is a COUNTED assertion (csrc/diag.h): no kernel of the whole session may have
    # seen a wave whose active lanes reported zero frames to their next envelope boundary
    ctx.synchronize()
    info = ctx.debug_info()
    zeros, misses = info["zero_segments"], info["fast_table_misses"]
    ctx.close()
    assert zeros == 0, f"{zeros} zero-frame segments counted during the GPU session (DESIGN.md section 7)"
    # ... and so is the promise the FAST copies of the block bodies are chosen on (kernels.h welsh_wave_tables_up)
    assert misses == 0, f"{misses} waves in a FAST body found a look-ahead table down during the GPU session"


@pytest.fixture()
def serial_kernels(gpu_ctx):
    """The serial (one voice per lane) Welsh kernels whatever the bank size: tests that target them
    (class-specialised bodies, the all-kinds kernel, the per-lane kernel) switch the time-parallel form off."""
    old, old_split = gpu_ctx.time_parallel_max_voices, gpu_ctx.split_max_waves
    gpu_ctx.time_parallel_max_voices = 0
    gpu_ctx.split_max_waves = 0
    yield gpu_ctx
    gpu_ctx.time_parallel_max_voices = old
    gpu_ctx.split_max_waves = old_split

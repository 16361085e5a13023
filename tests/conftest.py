import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One esfm context on cuda:0.  GPU tests must run the HIP path: no skip-on-missing-extension."""
    import torch
    assert torch.cuda.is_available(), "tests marked gpu need a GPU"
    import easysfm_amd as E
    assert os.path.exists(E.LIB_PATH), "libesfm_hip.so not built: run __graft_entry__.build()"
    ctx = E.Context(0, None)
    yield ctx
    ctx.close()

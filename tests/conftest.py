import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # dp.init_from_env() caps torch's intra-op CPU pool at one thread for a training process; inside the test
    # process that would leave the CPU oracle with one thread from the first trainer test on (measured: the GPU
    # suite 1 016 s instead of 481 s, and other oracle roundings than the bars were measured with)
    os.environ.setdefault("PARSENET_HOST_THREADS", "0")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")

import os
import sys
import warnings
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    import torch
    # one CPU thread for the oracle: its skinny products collapse when oversubscribed on many-core hosts,
    # batched torch.inverse has been seen to fail under MKL threading, and the golden fixtures were
    # generated single-threaded (bit-reproducible reductions)
    torch.set_num_threads(1)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """GPU tests: hand cached blocks back to the driver after every test so that the full-size cases
    (137 GB operators) always start from an empty allocator, whatever ran before them."""
    yield
    if "gpu" in request.keywords:
        import gc
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

"""pytest configuration: the `gpu` marker and loaders for the product package / the oracle."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_kslam():
    """Import k-slam_amd/ (not a valid module name) as `kslam_amd`."""
    if "kslam_amd" in sys.modules:
        return sys.modules["kslam_amd"]
    path = os.path.join(ROOT, "k-slam_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("kslam_amd", path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["kslam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def kslam():
    # torch carries its own copy of the HIP runtime; the product library uses the system one.  When the
    # library's copy opens the GPU first, torch's then reports no device (seen on the MI355X boxes), so
    # tests that generate inputs with torch on the GPU (tests/test_gpu_scale.py) let torch go first.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    return load_kslam()


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # test infrastructure only
    O.lib()
    return O


@pytest.fixture(scope="session")
def synth(kslam):
    import importlib
    return importlib.import_module("kslam_amd.synth")

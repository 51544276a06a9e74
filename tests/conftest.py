import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def load_pkg():
    """Import the product package from the (non-identifier) directory gr-bluetooth_amd/."""
    if "gr_bluetooth_amd" in sys.modules:
        return sys.modules["gr_bluetooth_amd"]
    pkg_dir = os.path.join(ROOT, "gr-bluetooth_amd")
    spec = importlib.util.spec_from_file_location(
        "gr_bluetooth_amd", os.path.join(pkg_dir, "__init__.py"),
        submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gr_bluetooth_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    """True when btgpu_create can open a gfx950 device (the product's own probe, no torch needed)."""
    try:
        pkg = load_pkg()
        blk = pkg.multi_LAP(8e6, 2476.5e6, 10.0)
        blk.close()
        return True
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or _gpu_present():
        return
    skip = pytest.mark.skip(reason="no gfx950 device (btgpu_create -> BTGPU_ENODEVICE)")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def synth(pkg):
    import importlib
    return importlib.import_module("gr_bluetooth_amd.synth")


@pytest.fixture(scope="session")
def po():
    import pyoracle
    pyoracle.build()
    return pyoracle

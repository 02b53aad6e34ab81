import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: needs a library built with -DF3DG_LAB (the superseded kernel generations and A/B switches: "
                                       "`F3DG_LAB=1 python f3d-gaus_amd/build.py --force`); skipped on the default library")


def lab_build():
    """True when libf3dg_hip.so was built with -DF3DG_LAB (f3dg_version() ends in "lab")."""
    from f3dgaus_amd import _lib
    return _lib.lib().f3dg_version().endswith(b"lab")


def pytest_collection_modifyitems(config, items):
    try:
        lab = lab_build()
    except Exception:
        lab = False
    if lab:
        return
    skip = pytest.mark.skip(reason="needs a -DF3DG_LAB build of the library (superseded kernel generations / A/B switches)")
    for item in items:
        if "lab" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def f3d():
    import f3dgaus_amd
    return f3dgaus_amd


def _apply_env_options():
    """F3DG_OPTIONS="bwd_walk=16,render_pack=0": library options for an A/B run of the whole suite (tools/ab_*.sh)."""
    spec = os.environ.get("F3DG_OPTIONS", "")
    if not spec:
        return
    from f3dgaus_amd import _lib
    L = _lib.lib()
    for item in spec.split(","):
        name, value = item.split("=")
        _lib.check(L.f3dg_set_option(name.strip().encode(), int(value)), "f3dg_set_option")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    _apply_env_options()
    return torch.device("cuda:0")

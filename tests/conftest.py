import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def ops():
    """The product's tensor-level ops; built library must exist (no fallback)."""
    import viditq_amd  # noqa: F401
    from viditq_amd import ops as _ops
    from viditq_amd import _lib
    _lib.load()
    return _ops


@pytest.fixture(autouse=True)
def _no_grad():
    import torch
    with torch.no_grad():
        yield


@pytest.fixture(scope="session")
def parity():
    """Recorder of ACHIEVED parity figures: tests put {name: {metric: value}}; written at session end to
    gpurun_out/parity.json (copied into profiles/ by the author), so that every tolerance asserted in the GPU tests
    can be read next to the value it bounds."""
    import json
    rec = {}
    yield rec
    if rec:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity.json")
        old = {}
        if os.path.exists(path):
            try:
                with open(path) as f:
                    old = json.load(f)
            except Exception:
                old = {}
        old.update(rec)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)

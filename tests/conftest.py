import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# what this pytest session selected and how it went: written into the parity record, so that a figure can never be read
# without knowing which tests produced the file (round 3's record had been rebuilt by a run that deselected the
# full-depth tests)
_RUN = {"args": [], "selected": 0, "deselected": 0, "passed": [], "failed": [], "skipped": []}


def pytest_collection_modifyitems(config, items):
    _RUN["args"] = [str(a) for a in config.invocation_params.args]
    _RUN["selected"] = len(items)


def pytest_deselected(items):
    _RUN["deselected"] += len(items)


def pytest_runtest_logreport(report):
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        _RUN[report.outcome if report.outcome in ("passed", "failed", "skipped") else "failed"].append(report.nodeid)


ORACLE_THREADS = 24     # tools/oracle_threads.py on the GPU box (256 hardware threads, torch default 128): one full-size oracle
#                         block takes 6.7 s on 128 threads, 3.6 on 64, 2.7 on 32, 2.3 on 24, 2.9 on 16, 3.3 on 8


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle (and every torch CPU reference of the kernel tests) is memory-bound elementwise work plus mid-size
    GEMMs: torch's default of one thread per core is ~3 x SLOWER than 24 threads on the GPU box's host.  Capped here so
    that the full-depth oracle tests take ~75 s each instead of ~220 s of the driver's 1200 s limit."""
    import torch
    n0 = torch.get_num_threads()
    if n0 > ORACLE_THREADS:
        torch.set_num_threads(ORACLE_THREADS)
    yield
    torch.set_num_threads(n0)


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
        # a FRESH file per session (no merging with an older record: stale keys would survive a renamed test), with
        # the session's own selection beside the figures
        rec["_run"] = {"pytest_args": _RUN["args"], "selected": _RUN["selected"], "deselected": _RUN["deselected"],
                       "passed": len(_RUN["passed"]), "failed": sorted(_RUN["failed"]), "skipped": sorted(_RUN["skipped"]),
                       "tests_that_recorded_or_passed": sorted(_RUN["passed"])}
        with open(os.path.join(out, "parity.json"), "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)

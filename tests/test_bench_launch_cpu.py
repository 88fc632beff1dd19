"""The launch forms of bench.py (no device work: VQ_BENCH_LAUNCH_ONLY=1 runs main()'s rendezvous / gather / print skeleton
over gloo).  The contract's N > 1 form is `torch.distributed.run ... bench.py --gpus N`; a plain `python bench.py --gpus N`
must start its own N ranks and must never fall through to one rank that prints n_gpus 1."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_python_form_starts_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3"], {"VQ_BENCH_REHEARSAL": "1", "VQ_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and len(line["per_rank_steps_per_s"]) == 2
    assert line["per_rank_steps_per_s"] == [3.0, 1.5]          # every rank's own entry arrived, in rank order


def test_more_ranks_than_devices_fails_loudly():
    # this container has no GPU: --gpus 2 without the rehearsal switch must refuse, not run one rank
    r = _run(["--gpus", "2"], {"VQ_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode != 0
    assert "refusing to run" in r.stderr and '"n_gpus"' not in r.stdout


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "1", "RANK": "0", "VQ_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_line_check_rejects_a_one_rank_line_for_a_multi_gpu_request():
    sys.path.insert(0, ROOT)
    import importlib
    import pytest
    bench = importlib.import_module("bench")

    class A:
        gpus = 2
    with pytest.raises(AssertionError):
        bench.check_line({"n_gpus": 1, "per_rank_steps_per_s": [1.0]}, A)
    with pytest.raises(AssertionError):
        bench.check_line({"n_gpus": 2, "per_rank_steps_per_s": [1.0, 1.0], "weights_broadcast": None}, A)
    bench.check_line({"n_gpus": 2, "per_rank_steps_per_s": [1.0, 1.0], "weights_broadcast": {"bytes": 5}}, A)

"""CPU-side checks of bench.py's plumbing: the cpu_baseline leg (the oracle timed on host cores) and the JSON
contract keys, without a GPU."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_leg_shape():
    b = _bench()
    r = b.cpu_baseline(1 << 12, 0.2)              # bounded sample: a fraction of a second here
    assert r["kind"] == "port" and r["unit"] == "env-steps/s" and r["cores"] >= 1
    assert r["value"] > 1e5 and "per-instance float64 CartPole" in r["sample"]
    assert r["single_instance_100k_steps_per_sec"] > 1e5
    json.dumps(r)


def test_bench_refuses_to_run_without_a_gpu():
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)


def test_group_leg_child_reports_instead_of_raising_without_a_gpu():
    """bench.py's single-process gymnet_group_* leg runs in a fresh child; whatever happens to it (here: no GPU at all) comes back
    as a record in the JSON line, never as an exception that would lose the headline."""
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU is present")
    b = _bench()

    class A:
        env, num_envs, steps = "CartPole-v1", 1 << 10, 4
    r = b.run_group_child(A, 2, timeout=240)
    assert isinstance(r, dict) and ("error" in r or any("error" in v for v in r.values() if isinstance(v, dict)))
    json.dumps(r)

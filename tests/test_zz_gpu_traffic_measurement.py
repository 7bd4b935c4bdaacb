"""Last test module of the run on purpose (see tests/test_gpu_group.py::test_driver_shaped_bench_line_carries_the_contract):
bench.py's roofline.traffic must be MEASURED in the run — two `rocprofv3 --pmc` child passes (FETCH_SIZE and WRITE_SIZE apart, no
trace flags beside --pmc, the program after `--` is python3 itself), HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
— and land within a few percent of the bytes the kernel moves; more would mean wasted re-reads."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bytes_per_step", [("f32", 41), ("f64", 73)])
def test_roofline_traffic_is_measured_by_pmc_child_passes(gpu_pkg, dtype, bytes_per_step):
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Args:
        env, num_envs, policy = "CartPole-v1", 1 << 20, ""
    Args.dtype = dtype
    traffic, how = bench.measure_traffic(Args, timeout=300)
    assert how.startswith("measured in this run") and "separate child passes" in how
    moved = bytes_per_step << 20
    assert 0.97 < traffic / moved < 1.06, (traffic, moved)

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


def test_a_fired_watchdog_prints_the_line_first_and_exits_non_zero():
    """a secondary section that hangs (a collective waiting for a dead peer) ends the process through the
    watchdog — the JSON line with the already-measured headline is printed FIRST, `watchdog_fired` names the section, and the
    exit status is 3, not 0.  spawn_ranks() turns any rank's non-zero status into the launcher's."""
    code = (
        "import importlib.util, threading, time, sys\n"
        f"spec = importlib.util.spec_from_file_location('bench_mod', r'{os.path.join(ROOT, 'bench.py')}')\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "out = {'metric': 'env-steps/sec', 'value': 1.0}\n"
        "b.make_watchdog(0, out, threading.Lock(), 'with_obs_allgather', 0.2)\n"
        "time.sleep(30)\n"                      # the 'hung collective'
        "print('NOT REACHED')\n")
    env = {k: v for k, v in os.environ.items() if k != "GYMNET_BENCH_WATCHDOG_RC"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-300:])
    line = json.loads(r.stdout.strip().splitlines()[0])
    assert line["value"] == 1.0 and line["watchdog_fired"] == "with_obs_allgather" and "timed out" in line["with_obs_allgather"]["error"]
    assert "NOT REACHED" not in r.stdout and "timed out after 0.2 s" in r.stderr
    b = _bench()
    assert b.WATCHDOG_EXIT == 3 or "GYMNET_BENCH_WATCHDOG_RC" in os.environ
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src                         # never replaces a GPU process


def test_xgmi_allgather_model_matches_the_survey_figures():
    b = _bench()
    m = b.allgather_model(8, 4 * (1 << 20) * 4)                                # 16 MiB per rank: CartPole, 2^20 lanes per GPU
    assert 105 < m["direct_us"] < 115 and 740 < m["ring_us"] < 800            # ~110 us direct, ~770 us ring
    assert b.parse_policy("vec=4, nt=12,block=128") == {"vec": 4, "nt": 12, "block": 128} and b.parse_policy("") == {}


def test_rollout_roofline_objects_can_be_recomputed_from_their_own_fields(tmp_path):
    """the fused rollout's roofline is VALU issue — floor = lanes x VALU per env-step / (16 lanes x SIMDs) / clock — and a
    reader must be able to recompute every fraction from the JSON alone.  The counter reader is checked on a synthetic rocpd database:
    dispatches of one kernel name are told apart by their order, each variant's first launch (its warm-up) is left out."""
    import sqlite3
    b = _bench()
    n = 1 << 20
    r = b.valu_roofline(n, 101.8, 2.81, 1024, "test", quarter_rate_per_env_step=(6.0, 0.5, 10.0))
    assert abs(r["issue_floor_us"] - n * 101.8 / (16 * 1024) / 2.4e3) < 1e-12 and abs(r["issue_floor_us"] - 2.7147) < 1e-3   # VERDICT r5's 2.71 us
    assert abs(r["frac"] - r["issue_floor_us"] / r["measured_us"]) < 1e-15 and 0.96 < r["frac"] < 0.97
    assert abs(r["frac_measured_rates"] - n * (4 * 101.8 + 6.0 + 2.5 + 13.0) / 64 / 1024 / 2.4e3 / 2.81) < 1e-12
    w = b.write_roofline(n, 25, 4.5)
    assert abs(w["achieved"] - 25 * n / 4.5e-6 / 1e9) < 1e-6 and abs(w["frac"] - w["achieved"] / 8000.0) < 1e-15
    json.dumps([r, w])
    db = str(tmp_path / "pmc_results.db")
    c = sqlite3.connect(db)
    c.execute("create table counters_collection (dispatch_id integer, kernel_name text, counter_name text, value real, duration integer)")
    seq = [{"variant": "f32_ring", "launches": 3, "steps": 64}, {"variant": "f32_sampled", "launches": 3, "steps": 64},
           {"variant": "f32_epsilon_greedy", "launches": 3, "steps": 64}]
    did = 0
    for x, (kernel, valu) in zip(seq, (("void gymnet::rollout_kernel<gymnet::CartPole, 4, true, false, false, 1, false>(a)", 100.0),
                                       ("void gymnet::rollout_kernel<gymnet::CartPole, 4, true, false, true, 1, false>(a)", 130.0),
                                       ("void gymnet::rollout_kernel<gymnet::CartPole, 4, true, false, true, 1, false>(a)", 160.0))):
        for launch in range(x["launches"]):
            did += 1
            for xcd in range(2):          # two rows per dispatch and counter: summed
                c.execute("insert into counters_collection values (?,?,?,?,?)", (did, kernel, "SQ_WAVES", 2048.0, 180000))
                c.execute("insert into counters_collection values (?,?,?,?,?)", (did, kernel, "SQ_INSTS_VALU", 2048.0 * 4 * 64 * (999.0 if launch == 0 else valu), 180000))
    c.execute("insert into counters_collection values (99, 'void gymnet::step_kernel<...>', 'SQ_WAVES', 1.0, 5000)")
    c.commit(); c.close()
    got = b.read_rollout_counters(db, seq)
    assert [round(got[x["variant"]]["valu_per_env_step"], 6) for x in seq] == [100.0, 130.0, 160.0]
    assert got["f32_sampled"]["lanes_per_thread"] == 4 and got["f32_epsilon_greedy"]["kernel"] == got["f32_sampled"]["kernel"]
    assert got["f32_ring"]["counters"]["_duration_ns"] == 180000
    busy, cycles = b.valu_busy_in_pass({"SQ_INSTS_VALU": 1.003e8, "GRBM_GUI_ACTIVE": 3.221e6, "SQ_BUSY_CYCLES": 1.158e7}, 1024)
    assert abs(cycles - 3.221e6 / 8) < 1 and abs(busy - 1.003e8 * 4 / 1024 / (3.221e6 / 8)) < 1e-12 and 0.96 < busy < 0.98     # (gpu_busy_check_r06.sh's dispatch)
    assert b.valu_busy_in_pass({"SQ_INSTS_VALU": 1.0}, 1024) == (None, None)
    import pytest
    with pytest.raises(RuntimeError):
        b.read_rollout_counters(db, seq[:2])                                   # the child's account and the database disagree
    const = json.load(open(os.path.join(ROOT, "profiles", "rollout_valu.json")))
    assert set(b.ROLLOUT_VARIANTS) <= set(const["variants"]) and all(v["valu_per_env_step"] > 50 for v in const["variants"].values())

#!/usr/bin/env python3
"""Round 6: turns the rollout / 2^27 passes of tools/gpu_profile_r06.sh (gpurun_out/p6/<tag>/rollout, big, bench_default.log)
into summary/rollout_box.json + summary/rocprof_rollouts_<tag>.txt.  Per fused-rollout variant: the kernel's average duration per vector
step under rocprofv3 --kernel-trace --stats, the SQ counters per env-step from the --pmc pass, the VALU-issue floor and the fractions —
against the profiled duration AND against the unprofiled bench line's HIP-event figure — so that every fraction can be recomputed
from the JSON alone.  The 2^27-lane point: average step-kernel duration, HBM-side traffic, fraction of 8 TB/s on the bytes moved."""
import importlib.util
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "box"
G = os.path.join(ROOT, "gpurun_out", "p6", tag)
S = os.path.join(G, "summary")
os.makedirs(S, exist_ok=True)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
N, SIMDS, CLOCK = 1 << 20, 1024, bench.ENGINE_CLOCK_GHZ


def json_line(path, key=None):
    if os.path.exists(path):
        for line in open(path, errors="replace"):
            if line.startswith("{") and (key is None or key in line):
                try:
                    return json.loads(line)
                except ValueError:
                    pass
    return None


def find_db(d):
    for dirpath, _, files in os.walk(d):
        for f in files:
            if f.endswith("_results.db"):
                return os.path.join(dirpath, f)
    return None


out = {"tag": tag, "lanes": N, "simds": SIMDS, "clock_GHz": CLOCK, "variants": {},
       "formula": "issue_floor_us = lanes x valu_per_env_step / (16 x simds) / clock_GHz / 1e3; frac_* = issue_floor_us / the named duration"}
txt = [f"# Round 6 box {tag}: fused rollouts at 2^20 CartPole lanes, 64 steps per launch (bench.py --rollout-child all), VALU-issue roofline\n"]
seq = (json_line(os.path.join(G, "rollout", "stats.log"), "rollout_child") or {}).get("rollout_child")
line = json_line(os.path.join(G, "bench_default.log"), "metric")
unprof = {}
if line:
    f32, f64 = line.get("fused_rollout") or {}, (line.get("cartpole_f64_2p20") or {}).get("fused_rollout") or {}
    for pre, leg in (("f32", f32), ("f64", f64)):
        for name, sub in (("ring", leg), ("sampled", leg.get("sampled_actions")), ("epsilon_greedy", leg.get("epsilon_greedy_actions"))):
            if sub and "us_per_step" in sub:
                unprof[f"{pre}_{name}"] = sub
    out["bench_line"] = {k: line[k] for k in ("value", "ms_per_step", "steps", "repeats") if k in line}
    out["bench_line"]["roofline"] = {k: line["roofline"].get(k) for k in ("frac", "frac_by_events", "launch_us", "traffic", "traffic_over_moved_bytes", "kernel")}
    if "sampled_actions_recorded" in f32:
        out["recorded"] = f32["sampled_actions_recorded"]
sdb, pdb = find_db(os.path.join(G, "rollout", "stats")), find_db(os.path.join(G, "rollout", "SQ"))
durs = {}
if sdb and seq:
    c = sqlite3.connect(sdb)
    rows = c.execute("select name, end - start from kernels where name like '%rollout_kernel%' order by start").fetchall()
    c.close()
    at = 0
    if len(rows) == sum(x["launches"] for x in seq):
        for x in seq:
            mine = rows[at + 1:at + x["launches"]]
            at += x["launches"]
            durs[x["variant"]] = (sum(d for _, d in mine) / len(mine) / 1e3 / x["steps"], mine[0][0])
counters = bench.read_rollout_counters(pdb, seq) if (pdb and seq) else {}
for v in bench.ROLLOUT_VARIANTS:
    e = {}
    if v in durs:
        e["rocprof_us_per_step"], e["kernel"] = durs[v]
    if v in counters:
        cs = counters[v]["counters"]
        per = cs["SQ_WAVES"] * counters[v]["lanes_per_thread"] * counters[v]["steps_per_launch"]
        cs = dict(cs)
        e.update(valu_per_env_step=counters[v]["valu_per_env_step"], lanes_per_thread=counters[v]["lanes_per_thread"],
                 int64_per_env_step=cs.get("SQ_INSTS_VALU_INT64", 0.0) / per, trans_per_env_step=cs.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / per,
                 f64_arith_per_env_step=(cs.get("SQ_INSTS_VALU_FMA_F64", 0.0) + cs.get("SQ_INSTS_VALU_ADD_F64", 0.0) + cs.get("SQ_INSTS_VALU_MUL_F64", 0.0)) / per)
        busy, cycles = bench.valu_busy_in_pass(cs, SIMDS)
        if busy is not None:
            e["valu_busy_in_pmc_pass"] = busy
            if cs.get("_duration_ns"):
                e["clock_GHz_in_pmc_pass"] = cycles / cs["_duration_ns"]
        e["issue_floor_us"] = N * e["valu_per_env_step"] / (16 * SIMDS) / (CLOCK * 1e3)
        clocks = 4.0 * e["valu_per_env_step"] + e["int64_per_env_step"] + 5.0 * e["trans_per_env_step"] + 1.3 * e["f64_arith_per_env_step"]
        e["issue_floor_measured_rates_us"] = N * clocks / 64.0 / SIMDS / (CLOCK * 1e3)
        if "rocprof_us_per_step" in e:
            e["frac_rocprof"] = e["issue_floor_us"] / e["rocprof_us_per_step"]
    if v in unprof:
        e["unprofiled_us_per_step"] = unprof[v]["us_per_step"]
        if "issue_floor_us" in e:
            e["frac_unprofiled"] = e["issue_floor_us"] / e["unprofiled_us_per_step"]
            e["frac_unprofiled_measured_rates"] = e["issue_floor_measured_rates_us"] / e["unprofiled_us_per_step"]
        if "roofline" in unprof[v]:
            e["bench_roofline"] = {k: unprof[v]["roofline"].get(k) for k in ("valu_per_env_step", "issue_floor_us", "measured_us", "frac", "frac_measured_rates")}
    if e:
        out["variants"][v] = e
        g = lambda k, p=3: (f"{e[k]:.{p}f}" if e.get(k) is not None else "-")   # noqa: E731
        txt.append(f"{v:20s} VALU/env-step {g('valu_per_env_step', 1):>6s} (int64 {g('int64_per_env_step', 2)}, trans {g('trans_per_env_step', 2)}, f64 {g('f64_arith_per_env_step', 1)})  floor {g('issue_floor_us')} us (measured rates {g('issue_floor_measured_rates_us')})  "
                   f"rocprof {g('rocprof_us_per_step')} us/step (frac {g('frac_rocprof')})  unprofiled {g('unprofiled_us_per_step')} (frac {g('frac_unprofiled')})  {e.get('kernel', '')[:70]}\n")
# the 2^27-lane point
big = {}
bdb = find_db(os.path.join(G, "big", "stats"))
NB, MOVED = 1 << 27, 41
if bdb:
    c = sqlite3.connect(bdb)
    r = c.execute("select avg(duration), count(*), name from kernels where name like '%step_kernel%'").fetchone()
    c.close()
    if r and r[0]:
        big.update(rocprof_us=r[0] / 1e3, launches=r[1], kernel=r[2], frac_rocprof=MOVED * NB / (r[0] * 1e-9) / 1e9 / 8000.0)
bl = json_line(os.path.join(G, "big", "stats.log"), "metric")
if bl:
    big.update(profiled_line_events_us=bl["roofline"]["launch_us"], profiled_line_ms_per_step=bl["ms_per_step"])
vals = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    db = find_db(os.path.join(G, "big", cn))
    if db:
        c = sqlite3.connect(db)
        r = c.execute("select avg(value), count(*) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?", (cn,)).fetchone()
        c.close()
        if r and r[0]:
            vals[cn] = r[0]
if len(vals) == 2:
    # (at 2^27 lanes the default policy runs ONE lane per thread: 4-byte reads, for which the guide's 2x FETCH_SIZE correction is not calibrated)
    big.update(fetch_size_KiB=vals["FETCH_SIZE"], write_size_KiB=vals["WRITE_SIZE"], traffic_raw=(vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
               traffic_2x_fetch=(2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, bytes_moved=MOVED * NB)
if line and line.get("hbm_resident_2p27"):
    big["bench_line"] = line["hbm_resident_2p27"]
if big:
    out["hbm_resident_2p27"] = big
    txt.append(f"\n2^27 lanes ({MOVED} B x 2^27 = {MOVED * NB / 1e9:.2f} GB per step): " + json.dumps({k: v for k, v in big.items() if k != 'bench_line'}) + "\n")
json.dump(out, open(os.path.join(S, "rollout_box.json"), "w"), indent=1)
open(os.path.join(S, f"rocprof_rollouts_{tag}.txt"), "w").write("".join(txt))
print("".join(txt))

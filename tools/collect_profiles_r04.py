#!/usr/bin/env python3
"""Round 4: turns what tools/gpu_profile_r04.sh left under gpurun_out/p4/ into the tracked summaries under profiles/ (tag r04)
and refreshes profiles/traffic.json (bench.py copies the matching entry into roofline.traffic, labelled with its source).

Per configuration: the rocprofv3 --kernel-trace --stats table of the bench command (the step kernel's average duration is the
number roofline.frac_by_events must agree with), HBM-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB from two
SEPARATE --pmc passes (on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read: MI355X_MICROARCH.md
§HBM), and the SQ counters with VALU instructions per env-step.  Fractions are priced on the bytes a kernel MOVES."""
import contextlib
import io
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G = os.path.join(ROOT, "gpurun_out", os.environ.get("GYMNET_PROFILE_DIR", "p4"))
P = os.environ.get("GYMNET_PROFILES_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
N = 1 << 20
# configuration -> (bench arguments, bytes one env-step moves, algorithmic bytes, envs per lane of the kernel it runs at 2^20)
CONFIGS = {
    "CartPole-v1": ("--env CartPole-v1", 41, 41, 4),
    "CartPole-v1-f64": ("--env CartPole-v1 --dtype f64", 73, 73, 4),   # round 4: the float64 kernel with 2 lane pairs per thread (now step_kernel_pipe2<CartPole64,...>)
    "Pendulum-v1": ("--env Pendulum-v1", 33, 37, 4),
    "MountainCar-v0": ("--env MountainCar-v0", 25, 25, 4),
    "Acrobot-v1": ("--env Acrobot-v1", 57, 65, 4),           # step_kernel_pipe: 4 sequential lanes per thread
}
PEAK = 8000.0


def capture(fn, *a):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def q(db, sql, *a):
    c = sqlite3.connect(db)
    try:
        r = c.execute(sql, a).fetchone()
        return r[0] if r else None
    finally:
        c.close()


def bench_line(path):
    if os.path.exists(path):
        for line in open(path, errors="replace"):
            if line.startswith("{"):
                return json.loads(line)
    return None


tj = os.path.join(P, "traffic.json")
src_tj = tj if os.path.exists(tj) else os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(src_tj)) if os.path.exists(src_tj) else {}
stats_out, pmc_out, table = [], [], []
for cfg, (args, moved, algo, lanes) in CONFIGS.items():
    d = os.path.join(G, cfg)
    db = os.path.join(d, "stats", "s_results.db")
    row = {"cfg": cfg}
    if os.path.exists(db):
        stats_out.append(f"## rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras {args}\n")
        stats_out.append(capture(rocpd_summary.stats, db))
        avg_ns = q(db, "select avg(duration) from kernels where name like '%step_kernel%'")
        j = bench_line(os.path.join(d, "stats.log"))
        if avg_ns:
            row["rocprof_us"] = avg_ns / 1e3
            row["frac_rocprof"] = moved * N / (avg_ns * 1e-9) / 1e9 / PEAK
            stats_out.append(f"## {cfg}: step kernel average {avg_ns / 1e3:.3f} us -> {moved} B x 2^20 / that = "
                             f"{moved * N / (avg_ns * 1e-9) / 1e9:.0f} GB/s = {row['frac_rocprof']:.3f} of 8 TB/s"
                             + (f" ({algo} algorithmic bytes: {algo * N / (avg_ns * 1e-9) / 1e9 / PEAK:.3f})" if algo != moved else "") + "\n")
        if j:
            rf = j["roofline"]
            row.update(events_us=rf["launch_us"], frac_events=rf.get("frac_by_events"), frac_wall=rf["frac"], ms_per_step=j["ms_per_step"])
            stats_out.append(f"## bench line of that profiled run: ms_per_step {j['ms_per_step']:.6f}  roofline.launch_us {rf['launch_us']:.3f}  "
                             f"frac (wall) {rf['frac']:.3f}  frac_by_events {rf.get('frac_by_events', float('nan')):.3f}  repeats {j['repeats']}  kernel {rf['kernel']}\n\n")
    vals = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        pdb = os.path.join(d, cn, "pmc_results.db")
        if os.path.exists(pdb):
            pmc_out.append(f"## {cfg}: rocprofv3 --pmc {cn} -- python3 bench.py --no-cpu-baseline --no-extras {args} --no-graph --steps 100 --warmup 10 --min-seconds 0\n")
            pmc_out.append(capture(rocpd_summary.pmc, pdb))
            vals[cn] = q(pdb, "select avg(value) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?", cn)
    if len(vals) == 2 and None not in vals.values():
        tr = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        traffic.setdefault(cfg, {})["1048576"] = tr
        row["traffic"] = tr
        pmc_out.append(f"## {cfg}: HBM-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {tr:.0f} B; bytes the kernel moves "
                       f"{moved} B x 2^20 = {moved * N} B (ratio {tr / (moved * N):.3f})\n\n")
    sq = os.path.join(d, "SQ", "pmc_results.db")
    if os.path.exists(sq):
        pmc_out.append(f"## {cfg}: SQ counters per dispatch (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles)\n")
        pmc_out.append(capture(rocpd_summary.pmc, sq))
        def a(cn):
            return q(sq, "select avg(value) from counters_collection where kernel_name like '%step_kernel%' and counter_name = ?", cn)
        iv, wv, wc, wi = a("SQ_INSTS_VALU"), a("SQ_WAVES"), a("SQ_WAVE_CYCLES"), a("SQ_WAIT_INST_ANY")
        if iv and wv:
            row["valu_per_step"] = iv / wv / lanes
            pmc_out.append(f"## {cfg}: SQ_INSTS_VALU / SQ_WAVES = {iv / wv:.1f} VALU instructions per wave = {iv / wv / lanes:.1f} per env-step "
                           f"({lanes} env(s) per lane)" + (f"; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {wi / wc:.2f}" if wi and wc else "") + "\n\n")
    table.append(row)
traffic["_source"] = f"profiles/rocprof_pmc_{tag}.txt: (2*FETCH_SIZE + WRITE_SIZE)*1024, separate rocprofv3 --pmc passes"
json.dump(traffic, open(tj, "w"), indent=1)
hdr = (f"# Round 4 summary at 2^20 lanes (fractions of 8 TB/s on the bytes each kernel MOVES)\n"
       f"# {'configuration':18s} {'rocprof us':>10s} {'frac':>6s} {'events us':>10s} {'frac':>6s} {'wall frac':>9s} {'VALU/step':>9s} {'HBM-side B':>12s}\n")
for r in table:
    f = lambda k, w, p: (f"{r[k]:{w}.{p}f}" if r.get(k) is not None else " " * (w - 1) + "-")  # noqa: E731
    hdr += f"# {r['cfg']:18s} {f('rocprof_us', 10, 3)} {f('frac_rocprof', 6, 3)} {f('events_us', 10, 3)} {f('frac_events', 6, 3)} {f('frac_wall', 9, 3)} {f('valu_per_step', 9, 1)} {f('traffic', 12, 0)}\n"
open(os.path.join(P, f"rocprof_stats_{tag}.txt"), "w").write(hdr + "\n" + "".join(stats_out))
open(os.path.join(P, f"rocprof_pmc_{tag}.txt"), "w").write("".join(pmc_out))
misc = []
for f in ("bench_driver_shape.log", "bench_default.log", "bench_gpus2_plain.log", "bench_gpus2_launcher.log", "rocminfo.log"):
    fp = os.path.join(G, f)
    if os.path.exists(fp):
        keep = [l for l in open(fp, errors="replace") if not l.startswith(("/opt/amdgpu", "RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl"))]
        misc.append(f"## {f}\n" + "".join(keep) + "\n")
open(os.path.join(P, f"bench_runs_{tag}.txt"), "w").write("".join(misc))
print(hdr)

#!/usr/bin/env python3
"""Round 6: merges the per-box roofline_box.json files that tools/gpu_profile_r06.sh + collect_profiles_r05.py left under
gpurun_out/p6/<box>/summary/ into profiles/roofline_r06.json — per configuration and figure the MEDIAN over the boxes sampled, with
min / max and the per-box values beside it.  No box is selected or dropped ("the median box with the spread beside it,
not the calmest one"): every gpurun call lands on whatever box the pool hands out, and all of them are in.
Also copies each box's text summaries to profiles/ (rocprof_stats_r06_<box>.txt, rocprof_pmc_r06_<box>.txt, rocprof_rollouts_r06_<box>.txt) and rewrites profiles/rollout_valu.json.
    python tools/merge_roofline_r06.py [boxA boxB ...]      (default: every directory under gpurun_out/p6)"""
import json
import os
import shutil
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P5 = os.path.join(ROOT, "gpurun_out", "p6")
OUT = os.path.join(ROOT, "profiles")
# A kernel that CHANGED during the round is reported from the boxes sampled after the change; the earlier boxes' rows of that
# configuration are kept beside it as `superseded` (never merged into the median, never dropped).
#   float64 CartPole: boxB..boxE ran the per-pair drain-loop reset (13.1 us); boxF.. run the deferred two-lanes-per-reset form.
SUPERSEDED = {}
boxes = sys.argv[1:] or sorted(d for d in os.listdir(P5) if os.path.exists(os.path.join(P5, d, "summary", "roofline_box.json")))
FIGS = [("rocprof_us", "rocprofv3 --kernel-trace --stats: average step-kernel duration (us), eager launches under the profiler"),
        ("frac_rocprof", "bytes moved / that / 8 TB/s"),
        ("graph_spacing_us", "begin-to-begin spacing (us) of consecutive launches, hipGraph replay of 1024 launches under --kernel-trace (median)"),
        ("frac_graph_spacing", "bytes moved / that / 8 TB/s"),
        ("graph_median_duration_us", "median kernel duration (us) inside the graph replay"),
        ("burst_spacing_us", "begin-to-begin spacing (us), EAGER 1024-launch regions under the profiler (the host pays ~8 us per traced launch: short kernels are starved)"),
        ("stats_spacing_us", "the same inside the 4096-launch regions of the stats pass"),
        ("unprofiled_ms_per_step", "unprofiled bench.py wall clock, ms per step"),
        ("unprofiled_events_us", "unprofiled bench.py, HIP events, us per launch"),
        ("frac_unprofiled_wall", "bytes moved / unprofiled wall / 8 TB/s"),
        ("frac_unprofiled_events", "bytes moved / unprofiled HIP events / 8 TB/s"),
        ("valu_per_step", "SQ_INSTS_VALU / SQ_WAVES / lanes per thread"),
        ("traffic", "HBM-side bytes per launch, (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes")]
data = {b: json.load(open(os.path.join(P5, b, "summary", "roofline_box.json"))) for b in boxes}
merged = {"peak_GBps": 8000.0, "lanes": 1 << 20, "boxes": boxes,
          "selection": "none: every box that ran tools/gpu_profile_r06.sh is included; `median` is over these boxes",
          "figures": {k: v for k, v in FIGS}, "configurations": {}}
first = data[boxes[0]]
merged["bytes_moved"], merged["bytes_algorithmic"] = first["bytes_moved"], first["bytes_algorithmic"]
for cfg in first["bytes_moved"]:
    all_rows = {b: next((r for r in data[b]["rows"] if r["cfg"] == cfg), None) for b in boxes}
    old = SUPERSEDED.get(cfg, {}).get("boxes", set())
    if not any(b not in old and all_rows[b] for b in boxes):
        old = set()                     # nothing newer was sampled: the old rows are the current ones

    def summarise(rows):
        entry = {"kernel": next((r.get("kernel") for r in rows.values() if r and r.get("kernel")), None)}
        for fig, _ in FIGS:
            vals = {b: r[fig] for b, r in rows.items() if r and r.get(fig) is not None}
            if vals:
                xs = list(vals.values())
                entry[fig] = {"median": st.median(xs), "min": min(xs), "max": max(xs), "per_box": vals}
        return entry

    entry = summarise({b: r for b, r in all_rows.items() if b not in old})
    if old:
        entry["superseded"] = dict(summarise({b: r for b, r in all_rows.items() if b in old}), what=SUPERSEDED[cfg]["what"])
    merged["configurations"][cfg] = entry
# ---- round 6: the fused rollouts (VALU-issue roofline) and the 2^27-lane point, from each box's summary/rollout_box.json -------------
def spread(vals):
    xs = list(vals.values())
    return {"median": st.median(xs), "min": min(xs), "max": max(xs), "per_box": vals}


rb = {b: json.load(open(os.path.join(P5, b, "summary", "rollout_box.json"))) for b in boxes if os.path.exists(os.path.join(P5, b, "summary", "rollout_box.json"))}
if rb:
    any_box = next(iter(rb.values()))
    roll = {"bound": "valu_issue", "lanes": any_box["lanes"], "simds": any_box["simds"], "clock_GHz": any_box["clock_GHz"], "formula": any_box["formula"],
            "figures": {"valu_per_env_step": "SQ_INSTS_VALU / SQ_WAVES / lanes per thread / steps per launch (one --pmc pass per box)",
                        "rocprof_us_per_step": "rocprofv3 --kernel-trace --stats: average rollout_kernel duration / 64 steps, launches 2-4 of each variant",
                        "unprofiled_us_per_step": "the default bench line's fused legs: HIP events over >= 1024 steps",
                        "frac_rocprof / frac_unprofiled": "issue_floor_us over the respective duration",
                        "issue_floor_measured_rates_us": "the floor with v_mad_u64_u32 at 5, transcendentals at 9 and float64 arithmetic at 5.3 clocks (profiles/issue_rate_r06.txt)",
                        "valu_busy_in_pmc_pass": "SQ_INSTS_VALU x 4 / simds / (GRBM_GUI_ACTIVE / 8 XCDs): VALU-pipe occupancy of the profiled launches in the chip's own clocks",
                        "clock_GHz_in_pmc_pass": "GRBM_GUI_ACTIVE / 8 / the profiled launch's duration"},
            "variants": {}}
    names = [v for v in any_box["variants"]]
    for v in names:
        e = {"kernel": next((rb[b]["variants"][v].get("kernel") for b in rb if v in rb[b]["variants"] and rb[b]["variants"][v].get("kernel")), None)}
        for fig in ("valu_per_env_step", "int64_per_env_step", "trans_per_env_step", "f64_arith_per_env_step", "issue_floor_us", "issue_floor_measured_rates_us",
                    "rocprof_us_per_step", "frac_rocprof", "unprofiled_us_per_step", "frac_unprofiled", "frac_unprofiled_measured_rates", "valu_busy_in_pmc_pass", "clock_GHz_in_pmc_pass"):
            vals = {b: rb[b]["variants"][v][fig] for b in rb if v in rb[b]["variants"] and rb[b]["variants"][v].get(fig) is not None}
            if vals:
                e[fig] = spread(vals)
        roll["variants"][v] = e
    rec = {b: rb[b]["recorded"]["roofline"] for b in rb if rb[b].get("recorded", {}).get("roofline")}
    if rec:
        roll["sampled_actions_recorded"] = {"bound": "hbm_write", "written_bytes_per_env_step": next(iter(rec.values()))["written_bytes_per_env_step"],
                                            "measured_us": spread({b: r["measured_us"] for b, r in rec.items()}),
                                            "achieved_GBps": spread({b: r["achieved"] for b, r in rec.items()}),
                                            "frac_of_8TBps": spread({b: r["frac"] for b, r in rec.items()}),
                                            "frac_of_write_ceiling_4800": spread({b: r["frac_of_write_ceiling"] for b, r in rec.items()})}
    merged["fused_rollouts"] = roll
    big = {b: rb[b]["hbm_resident_2p27"] for b in rb if rb[b].get("hbm_resident_2p27")}
    if big:
        e = {"lanes": 1 << 27, "bytes_moved_per_launch": 41 * (1 << 27), "kernel": next(iter(big.values())).get("kernel")}
        for fig in ("rocprof_us", "frac_rocprof", "traffic_raw", "traffic_2x_fetch", "fetch_size_KiB", "write_size_KiB"):
            vals = {b: r[fig] for b, r in big.items() if r.get(fig) is not None}
            if vals:
                e[fig] = spread(vals)
        vals = {b: r["bench_line"]["launch_us"] for b, r in big.items() if r.get("bench_line", {}).get("launch_us")}
        if vals:
            e["unprofiled_events_us"] = spread(vals)
            e["frac_unprofiled_events"] = spread({b: 41 * (1 << 27) / (u * 1e-6) / 1e9 / 8000.0 for b, u in vals.items()})
        merged["hbm_resident_2p27"] = e
    head = {b: rb[b]["bench_line"] for b in rb if rb[b].get("bench_line")}
    if head:
        merged["default_bench_line"] = {"value": spread({b: h["value"] for b, h in head.items()}), "ms_per_step": spread({b: h["ms_per_step"] for b, h in head.items()}),
                                        "frac_wall": spread({b: h["roofline"]["frac"] for b, h in head.items()}),
                                        "traffic_over_moved_bytes": spread({b: h["roofline"]["traffic_over_moved_bytes"] for b, h in head.items() if h["roofline"].get("traffic_over_moved_bytes")}),
                                        "note": "python bench.py with no flags (4096-step regions) on each profile box"}
    # the constants bench.py falls back on when it cannot run its own --pmc pass
    json.dump({"_source": f"median over boxes {sorted(rb)} of tools/gpu_profile_r06.sh (profiles/roofline_r06.json fused_rollouts)",
               "variants": {v: {"valu_per_env_step": roll["variants"][v]["valu_per_env_step"]["median"], "kernel": roll["variants"][v]["kernel"]}
                            for v in roll["variants"] if "valu_per_env_step" in roll["variants"][v]}},
              open(os.path.join(OUT, "rollout_valu.json"), "w"), indent=1)
    for b in rb:
        src = os.path.join(P5, b, "summary", f"rocprof_rollouts_{b}.txt")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(OUT, f"rocprof_rollouts_r06_{b}.txt"))
os.makedirs(OUT, exist_ok=True)
json.dump(merged, open(os.path.join(OUT, "roofline_r06.json"), "w"), indent=1)
for b in boxes:
    for f in (f"rocprof_stats_{b}.txt", f"rocprof_pmc_{b}.txt"):
        src = os.path.join(P5, b, "summary", f)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(OUT, f.replace(b, f"r06_{b}")))
print(f"{len(boxes)} boxes: {boxes}")
print(f"{'configuration':18s} {'rocprof us med [min, max]':>28s} {'frac':>6s} {'graph spacing us':>24s} {'frac':>6s} {'unprofiled wall us':>24s} {'frac':>6s} {'VALU':>6s}")
for cfg, e in merged["configurations"].items():
    def f(k, scale=1.0):
        v = e.get(k)
        return f"{v['median'] * scale:7.3f} [{v['min'] * scale:6.3f}, {v['max'] * scale:6.3f}]" if v else " " * 23 + "-"
    def m(k):
        v = e.get(k)
        return f"{v['median']:6.3f}" if v else "     -"
    print(f"{cfg:18s} {f('rocprof_us'):>28s} {m('frac_rocprof')} {f('graph_spacing_us'):>24s} {m('frac_graph_spacing')} {f('unprofiled_ms_per_step', 1e3):>24s} {m('frac_unprofiled_wall')} {m('valu_per_step')}")

if merged.get("fused_rollouts"):
    print(f"\n{'fused rollout':20s} {'VALU/env-step':>14s} {'floor us':>9s} {'rocprof us/step med [min, max]':>32s} {'frac':>6s} {'unprofiled us/step':>26s} {'frac':>6s}")
    for v, e in merged["fused_rollouts"]["variants"].items():
        def f(k):
            x = e.get(k)
            return f"{x['median']:7.3f} [{x['min']:6.3f}, {x['max']:6.3f}]" if x else "-"
        def m(k, p=3):
            x = e.get(k)
            return f"{x['median']:.{p}f}" if x else "-"
        print(f"{v:20s} {m('valu_per_env_step', 1):>14s} {m('issue_floor_us'):>9s} {f('rocprof_us_per_step'):>32s} {m('frac_rocprof'):>6s} {f('unprofiled_us_per_step'):>26s} {m('frac_unprofiled'):>6s}")
if merged.get("hbm_resident_2p27"):
    e = merged["hbm_resident_2p27"]
    print("2^27 lanes:", {k: (round(v["median"], 4) if isinstance(v, dict) else v) for k, v in e.items() if k != "kernel"})

#!/usr/bin/env python3
"""Round 5: merges the per-box roofline_box.json files that tools/gpu_profile_r05.sh + collect_profiles_r05.py left under
gpurun_out/p5/<box>/summary/ into profiles/roofline_r05.json — per configuration and figure the MEDIAN over the boxes sampled, with
min / max and the per-box values beside it.  No box is selected or dropped ("the median box with the spread beside it,
not the calmest one"): every gpurun call lands on whatever box the pool hands out, and all of them are in.
Also copies each box's text summaries to profiles/ (rocprof_stats_r05_<box>.txt, rocprof_pmc_r05_<box>.txt).
    python tools/merge_roofline_r05.py [boxA boxB ...]      (default: every directory under gpurun_out/p5)"""
import json
import os
import shutil
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P5 = os.path.join(ROOT, "gpurun_out", "p5")
OUT = os.path.join(ROOT, "profiles")
# A kernel that CHANGED during the round is reported from the boxes sampled after the change; the earlier boxes' rows of that
# configuration are kept beside it as `superseded` (never merged into the median, never dropped).
#   float64 CartPole: boxB..boxE ran the per-pair drain-loop reset (13.1 us); boxF.. run the deferred two-lanes-per-reset form.
SUPERSEDED = {"CartPole-v1-f64": {"boxes": {"boxA", "boxB", "boxC", "boxD", "boxE"},
                                  "what": "step_kernel_pipe2<CartPole64,4> with the per-pair drain-loop reset (before commit cb25336)"}}
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
          "selection": "none: every box that ran tools/gpu_profile_r05.sh is included; `median` is over these boxes",
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
os.makedirs(OUT, exist_ok=True)
json.dump(merged, open(os.path.join(OUT, "roofline_r05.json"), "w"), indent=1)
for b in boxes:
    for f in (f"rocprof_stats_{b}.txt", f"rocprof_pmc_{b}.txt"):
        src = os.path.join(P5, b, "summary", f)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(OUT, f.replace(b, f"r05_{b}")))
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

#!/bin/bash
# Round 5 probe (runs ON THE GPU BOX): SQ counters of the fused rollout kernels (float32 and float64 CartPole, lean, ring actions) — VALU instructions
# per env-step and how busy the VALU is — from one --pmc pass over a short script (no trace flags with --pmc).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_rollout
rm -rf $O; mkdir -p $O
cat > $O/run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import __graft_entry__ as ge
pkg = ge.load_package()
n, ring, T = 1 << 20, 16, 64
acts = torch.randint(0, 2, (ring, n), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for dt in ("float32", "float64"):
    with pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True, dtype=dt) as e:
        e.ResetDevice()
        for _ in range(4):
            e.RolloutFusedDevice(acts, T, n, ring)
        e.Sync()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc -o pmc -- python3 $O/run.py > $O/run.log 2>&1
cd $R
python3 - <<PY
import glob, sqlite3
for db in glob.glob("$O/pmc/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%rollout_kernel%' group by kernel_name, counter_name").fetchall()
    by = {}
    for k, cn, cnt, avg, dur in rows:
        by.setdefault(k, {})[cn] = avg; by[k]["_dur_us"] = dur / 1e3; by[k]["_n"] = cnt
    for k, v in by.items():
        import re
        lanes_per_thread = int(re.search(r"rollout_kernel<[^,]+,\s*(\d+)", k).group(1))
        valu_per_step = v["SQ_INSTS_VALU"] / v["SQ_WAVES"] / lanes_per_thread / 64
        print(k[:90])
        print("   dispatches %d, avg %.1f us per 64-step launch = %.3f us per step; VALU per env-step %.1f; VALU active / wave cycles %.3f; wait-inst / wave cycles %.3f; busy cycles %.0f"
              % (v["_n"], v["_dur_us"], v["_dur_us"] / 64, valu_per_step, v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_BUSY_CYCLES"]))
PY
rm -rf $O/pmc

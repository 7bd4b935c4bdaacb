#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel instantiation, read from the gfx950 assembly the product's flags produce (no GPU needed):
VGPRs -> waves per SIMD (512 / VGPRs, at most 8), scratch bytes (spills), static LDS bytes.  Round 5 found two slowdowns that only this view
explains — a float32 bookkeeping rollout that allocated 131-152 VGPRs (three waves per SIMD where 2^20 lanes need four) and a float64
four-pair kernel that must stay at two waves per SIMD because it spills when capped for three (profiles/occupancy_hints_r05.txt).
    python tools/kernel_resources.py [env ...]        env: cartpole cartpole64 pendulum mountaincar acrobot (default: all)
Prints one line per kernel; tests/test_kernel_resources.py asserts the invariants the launch policy relies on."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym.net_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S"]
ENVS = ("cartpole", "cartpole64", "pendulum", "mountaincar", "acrobot")


def assembly(env, outdir):
    out = os.path.join(outdir, env + ".s")
    r = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, f"env_{env}.hip"), "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return out


def kernels(path):
    """{demangled kernel name without parameter list: {"vgpr", "occupancy", "scratch", "lds"}}"""
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = [lines[i].split(":")[0] for i in starts]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for idx, i in enumerate(starts):
        j = starts[idx + 1] if idx + 1 < len(starts) else len(lines)
        meta = {}
        for l in lines[i:j]:
            m = re.match(r";\s*(NumVgprs|ScratchSize|Occupancy|LDSByteSize):\s*(\d+)", l.strip())
            if m:
                meta[m.group(1)] = int(m.group(2))
        if "NumVgprs" not in meta:
            continue
        name = re.sub(r"\(.*\)$", "", dem[idx].replace("gymnet::", "").replace("void ", "")).replace(", ", ",")
        out[name] = {"vgpr": meta["NumVgprs"], "occupancy": meta.get("Occupancy", -1), "scratch": meta.get("ScratchSize", 0), "lds": meta.get("LDSByteSize", 0)}
    return out


def collect(envs=ENVS):
    with tempfile.TemporaryDirectory() as d:
        with ThreadPoolExecutor(max_workers=min(len(envs), os.cpu_count() or 4)) as ex:
            paths = list(ex.map(lambda e: assembly(e, d), envs))
        res = {}
        for p in paths:
            res.update(kernels(p))
    return res


if __name__ == "__main__":
    envs = tuple(sys.argv[1:]) or ENVS
    res = collect(envs)
    print(f"# hipcc {' '.join(FLAGS)}  (gym.net_amd/csrc/env_*.hip)   waves per SIMD = min(8, 512 // VGPRs)")
    print(f"{'kernel':72s} {'VGPRs':>6s} {'waves/SIMD':>10s} {'scratch B':>10s} {'LDS B':>7s}")
    for k in sorted(res):
        v = res[k]
        print(f"{k:72s} {v['vgpr']:6d} {v['occupancy']:10d} {v['scratch']:10d} {v['lds']:7d}")

"""Builds gym.net_amd/lib/libgymnet_amd.so: the HIP kernels + the C ABI, for gfx950 only.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared csrc/*.hip -ldl -o lib/libgymnet_amd.so
(done as one `hipcc -c` per .hip file in parallel plus one link: same flags, same result.  Every env's step / rollout / reset
kernels are a translation unit of their own — csrc/env_*.hip instantiating csrc/step_kernels.hpp — so the eight files compile
side by side: ~15 s wall on 8 cores instead of ~40 s for the former single kernels.hip)

-ffp-contract=off is part of the numerical contract (see csrc/envs.hpp): every float32 operation
rounds on its own, in the order written.  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libgymnet_amd.so")
SOURCES = ["env_cartpole.hip", "env_cartpole64.hip", "env_acrobot.hip", "env_pendulum.hip", "env_mountaincar.hip", "kernels.hip", "capi.hip",
           "group.hip"]
DEPS = SOURCES + ["kernels.hpp", "step_kernels.hpp", "lanes.hpp", "envs.hpp", "cartpole64.hpp", "philox.hpp", "handle.hpp",
                  os.path.join("..", "..", "include", "gymnet_amd.h")]
# -fno-slp-vectorize: on gfx950 a packed FP32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) occupies the SIMD about
# as long as the two scalar instructions it replaces (~5 cycles against ~2.4 each in these kernels' instruction mix:
# tools/acrobot_alu_probe.hip, tools/valu_probe.hip, profiles/*_r02.txt), so the compiler's opportunistic pairing saves
# nothing and costs the v_mov shuffles that build the pairs (Acrobot harness: 465 VALU with it, 452 without; the four
# bench kernels time the same either way).  Off = deterministic, purely scalar code generation.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]
LIBS = ["-ldl"]          # librccl is dlopen()ed on demand by group.hip, never linked
# (The library never reads the process environment for its launch policy: gymnet_vecenv_set_launch_policy is the interface.  The
# GYMNET_BUILD_PROBE_ENV probe build of rounds 1-4 went away in round 5 together with the scripts that needed it.)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the HIP extension")
    return exe


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # one hipcc -c per translation unit, side by side (kernels.hip alone is most of the build), then one link
    import concurrent.futures
    import tempfile
    compile_flags = [f for f in FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="gymnet_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            cmd = [hipcc()] + compile_flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
            return obj
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + LIBS + ["-o", OUT + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (link) failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    os.replace(OUT + ".tmp", OUT)
    if verbose:
        print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)

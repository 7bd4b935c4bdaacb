"""CPU sanitizer runs (SURVEY §5 "-fsanitize=address host build").  GPU AddressSanitizer is not available on this
pool, so what is sanitized is everything that runs on the host and is compiled here:

  - the oracle's C restatement (oracle/classic_control_ref.c) and the threaded CPU baseline (oracle/cpu_baseline.c), built by
    `make -C oracle asan ubsan tsan` and loaded IN PLACE of liboracle.so (GYMNET_ORACLE_SO + LD_PRELOAD of the sanitizer runtime):
    the whole of tests/test_oracle.py runs through the AddressSanitizer and the UndefinedBehaviorSanitizer builds, the pthread
    baseline through the ThreadSanitizer build;
  - the C++ host mirror of the reference interface (include/gymnet_amd.hpp, tests/cpp/host_mirror_test.cpp) built with
    -fsanitize=address,undefined against tests/cpp/abi_stub.c — a test-only, memory-honest stand-in of the C ABI that touches exactly
    the bytes the header documents (never the oracle, never the product library, never a GPU).
A sanitizer report fails the run (abort_on_error / halt_on_error / -fno-sanitize-recover)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "build")


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip(f"{name} is not installed with this gcc")
    return p


@pytest.fixture(scope="module")
def sanitizer_builds():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "ubsan", "tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return {k: os.path.join(BUILD, f"liboracle_{k}.so") for k in ("asan", "ubsan", "tsan")}


def _env(so, runtime, **opts):
    e = dict(os.environ, GYMNET_ORACLE_SO=so, LD_PRELOAD=runtime, PYTHONDONTWRITEBYTECODE="1")
    e.update(opts)
    return e


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["asan", "ubsan"])
def test_oracle_suite_is_clean_under_asan_and_ubsan(sanitizer_builds, kind):
    """tests/test_oracle.py — every golden vector, known-answer test, exhaustive division check and restatement cross-check —
    with the C restatement compiled under the sanitizer."""
    rt = _runtime("libasan.so" if kind == "asan" else "libubsan.so")
    opts = ({"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=1"} if kind == "asan"
            else {"UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=_env(sanitizer_builds[kind], rt, **opts), capture_output=True, text=True, timeout=850)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_threaded_cpu_baseline_is_clean_under_tsan(sanitizer_builds):
    """oracle/cpu_baseline.c splits the envs over POSIX threads (bench.py's cpu_baseline leg): no data race, and the threaded run
    is reproducible run to run."""
    rt = _runtime("libtsan.so")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import capi\n"
            "assert capi._SO.endswith('liboracle_tsan.so')\n"
            "a = capi.cpu_baseline(4096, 300, 8, True); a2 = capi.cpu_baseline(4096, 300, 8, True); b = capi.cpu_baseline(4096, 300, 1, True); c = capi.cpu_baseline(4096, 300, 3, False)\n"
            "assert a['env_steps'] == b['env_steps'] == c['env_steps'] == 4096 * 300 and min(a['dones'], b['dones'], c['dones']) > 0\n"
            "assert a['dones'] == a2['dones'] and a['checksum'] == a2['checksum']     # (each thread seeds its own block: deterministic per thread count)\n"
            "print('TSAN_RUN_OK', a['dones'])\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=_env(sanitizer_builds["tsan"], rt, TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0 exitcode=66"))
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "TSAN_RUN_OK" in r.stdout and "WARNING: ThreadSanitizer" not in out, out[-3000:]


def test_cpp_host_mirror_is_clean_under_asan_and_ubsan(tmp_path):
    """include/gymnet_amd.hpp's host classes against the memory-honest ABI stub, -fsanitize=address,undefined: every buffer the
    classes size and hand across the ABI (observations [N, D], records, replicas [G][D][N/G], pinned buffers), handle lifetime,
    the exception mapping."""
    exe = str(tmp_path / "host_mirror_stub_san")
    inc = os.path.join(ROOT, "include")
    obj = str(tmp_path / "abi_stub.o")
    r = subprocess.run(["gcc", "-std=gnu11", "-O1", "-g", "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-I", inc, "-c", os.path.join(ROOT, "tests", "cpp", "abi_stub.c"), "-o", obj],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-I", inc, os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"), obj, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=1:detect_leaks=1:detect_stack_use_after_return=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, "--stub"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "stub: 0 failed check(s)" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    # the harness detects what it is there to detect: the same stub run with a deliberately short observation buffer must be reported
    probe = tmp_path / "probe.cpp"
    probe.write_text('#include <vector>\n#include "gymnet_amd.h"\nint main() { gymnet_config c{}; c.struct_size = sizeof c; c.env_id = 0; c.num_envs = 64;\n'
                     '  gymnet_vecenv *h = nullptr; if (gymnet_vecenv_create(&c, &h)) return 3; std::vector<float> obs(64 * 4 - 1);\n'
                     '  gymnet_vecenv_reset(h, obs.data()); gymnet_vecenv_destroy(h); return 0; }\n')
    exe2 = str(tmp_path / "probe_san")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-I", inc, str(probe), obj, "-o", exe2], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe2], capture_output=True, text=True, timeout=60, env=dict(env, ASAN_OPTIONS="abort_on_error=0:exitcode=77"))
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr, (r.returncode, r.stderr[-500:])

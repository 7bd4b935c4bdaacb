"""Builds tests/cpp/host_mirror_test.cpp (g++, C++17) against include/gymnet_amd.hpp + libgymnet_amd.so and runs
it: the compiled-language host mirror of the reference interface (the reference is C#; no .NET here)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "build")
EXE = os.path.join(OUT_DIR, "host_mirror_test")


def _build(gymnet):
    lib_dir = os.path.dirname(gymnet.LIB_PATH)
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "include", "gymnet_amd.hpp"), os.path.join(ROOT, "include", "gymnet_amd.h"), gymnet.LIB_PATH]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L", lib_dir, "-lgymnet_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return EXE


def test_cpp_host_mirror_cpu(gymnet):
    exe = _build(gymnet)
    r = subprocess.run([exe, "--cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_host_mirror_gpu(gpu_pkg):
    exe = _build(gpu_pkg)
    r = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu+gpu: 0 failed" in r.stdout

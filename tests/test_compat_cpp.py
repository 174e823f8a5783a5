"""Builds and runs the C++ tests of the drop-in shim (tests/cpp/test_compat.cpp): the reference's own call surface
(namespace mvSLAM) linked against libmvslam_hip.so.  Compiling + linking is a CPU test; running needs the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "mvslam_amd", "lib", "test_compat")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "test_compat.cpp")
    libdir = os.path.join(ROOT, "mvslam_amd", "lib")
    assert os.path.exists(os.path.join(libdir, "libmvslam_hip.so")), "build the HIP library first (__graft_entry__.build)"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", EXE, src, "-L", libdir, "-lmvslam_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)


def test_compat_shim_compiles_and_links_with_host_compiler():
    """The shim needs nothing but a C++17 host compiler and the C-ABI library (no hipcc, no torch)."""
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_compat_shim_cpp_suite():
    deps = [os.path.join(ROOT, "tests", "cpp", "test_compat.cpp"), os.path.join(ROOT, "mvslam_amd", "compat", "mvslam_compat.hpp"),
            os.path.join(ROOT, "include", "mvslam_hip.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        _build()   # a stale binary would test yesterday's shim
    p = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and "ALL PASSED" in out, out

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


DRIVER = os.path.join(ROOT, "mvslam_amd", "lib", "reconstruct_scene")


def _build_driver():
    src = os.path.join(ROOT, "tests", "cpp", "reconstruct_scene.cpp")
    libdir = os.path.join(ROOT, "mvslam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", DRIVER, src, "-L", libdir, "-lmvslam_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])


def test_reconstruct_scene_driver_compiles():
    """the call sequence of utility/reconstruct-scene.cpp on the shim: only a host compiler needed"""
    _build_driver()
    assert os.path.exists(DRIVER)


@pytest.mark.gpu
def test_reconstruct_scene_driver_on_tsukuba(tmp_path):
    """the call sequence of the reference's two-image utility on its own tsukuba frames 1 and 2 with its camera.config:
    extract -> match_and_filter -> sfm_solve gives the pose test/test-image-pair.cpp:38-45 expects, (I, (1, 0, 0))"""
    import numpy as np

    deps = [os.path.join(ROOT, "tests", "cpp", "reconstruct_scene.cpp"), os.path.join(ROOT, "mvslam_amd", "compat", "mvslam_compat.hpp")]
    if not os.path.exists(DRIVER) or any(os.path.getmtime(d) > os.path.getmtime(DRIVER) for d in deps):
        _build_driver()
    g = np.load(os.path.join(ROOT, "tests", "golden", "tsukuba_gray.npz"))
    imgs = g["images"]
    h, w = imgs.shape[1:]
    f1, f2, cam = str(tmp_path / "1.raw"), str(tmp_path / "2.raw"), str(tmp_path / "camera.config")
    imgs[0].tofile(f1)
    imgs[1].tofile(f2)
    open(cam, "w").write("350 350 0 192 144\n0 0 0 1.5708 0 0\n")     # data/tsukuba/camera.config
    p = subprocess.run([DRIVER, f1, f2, str(w), str(h), cam, "50"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    line = [ln for ln in out.splitlines() if ln.startswith("scaled transformation")][0]
    se3 = np.array([float(x) for x in line.split("=")[1].split()])
    assert np.abs(se3 - np.array([1.0, 0, 0, 0, 0, 0])).max() < 1e-3, out
    assert "camera intrinsics: fx 350 fy 350 shear 0 px 192 py 144" in out
    npts = int([ln for ln in out.splitlines() if ln.startswith("pointsin1_scaled")][0].split("=")[1].split()[0])
    assert npts > 60


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

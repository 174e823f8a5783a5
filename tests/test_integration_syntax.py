"""The forwarding translation units a maintainer drops into the mvSLAM tree (integration/source/vision/*.cpp) are written
against the reference's real OpenCV / Eigen types, and this image has neither library: they cannot be built here.  They CAN be
syntax- and signature-checked: `g++ -fsyntax-only` against the reference's own, unchanged headers (/root/reference/source), with
declarations of the small OpenCV / Eigen subset those headers name under tests/cpp/stubs/ (test-side only: permissive
declarations, nothing is compiled to code, nothing of the reference is built, nothing here ships).

What the check catches: a forwarding file that no longer parses, names a member the reference's headers do not declare,
defines a class member with a signature the reference's class does not have (VisualFeature::match_visual_features,
FundamentalMatrixEstimatorRANSAC::compute: hard errors), or defines a FREE function with a signature that differs from the
reference's declaration -- C++ would silently accept that as a new overload, so a probe takes the function's address without a
cast, which is ambiguous as soon as two overloads exist (vision/sfm.hpp:30-53, pnp.hpp:22-26, fundamental-matrix.hpp:16-19,
ba.hpp:25-36).  Round 3's first run of it found a real defect: `Eigen::Map<RowMajor3>(out) = M;` in the glue header parses as
a declaration of `out`.

Skipped when /root/reference is absent (the GPU box)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/source"
TUS = sorted(glob.glob(os.path.join(ROOT, "integration", "source", "vision", "*.cpp")))
FLAGS = ["g++", "-std=c++11", "-fsyntax-only", "-I" + REF, "-I" + os.path.join(ROOT, "tests", "cpp", "stubs"),
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration", "source", "vision")]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present on this machine")

# free functions each file must define with exactly the reference's signature
FREE = {"sfm-solve.cpp": ["sfm_solve", "sfm_triangulate"], "pnp-solve.cpp": ["pnp_solve"],
        "fundamental-matrix.cpp": ["find_fundamental_matrix"], "ba.cpp": ["ba_frame_pose_and_point"]}


def _check(src_text, cwd):
    p = subprocess.run(FLAGS + ["-x", "c++", "-"], input=src_text, capture_output=True, text=True, cwd=cwd, timeout=300)
    return p.returncode, p.stderr


def test_every_forwarding_file_exists():
    names = {os.path.basename(f) for f in TUS}
    assert {"visual-feature.cpp", "fundamental-matrix.cpp", "estimator-RANSAC.cpp", "sfm-solve.cpp", "pnp-solve.cpp",
            "ba.cpp"} <= names


@pytest.mark.parametrize("tu", TUS, ids=[os.path.basename(f) for f in TUS])
def test_forwarding_file_parses_against_the_reference_headers(tu):
    probes = "".join("static auto probe_%s = &mvSLAM::%s;   // ambiguous if the definition added an overload\n" % (f, f)
                     for f in FREE.get(os.path.basename(tu), []))
    rc, err = _check('#include "%s"\n%s' % (tu, probes), os.path.dirname(tu))
    assert rc == 0, err[-3000:]


def test_a_wrong_signature_is_caught():
    """the check has teeth: sfm_solve with one parameter type changed (size_t -> int indices) must not pass"""
    tu = os.path.join(ROOT, "integration", "source", "vision", "sfm-solve.cpp")
    text = open(tu).read()
    bad = text.replace("std::vector<Point3> &pointsin1_scaled, std::vector<size_t> &point_indexes)",
                       "std::vector<Point3> &pointsin1_scaled, std::vector<int> &point_indexes)", 1)
    assert bad != text
    rc, err = _check(bad + "\nstatic auto probe = &mvSLAM::sfm_solve;\n", os.path.dirname(tu))
    assert rc != 0 and ("sfm_solve" in err or "overloaded" in err), err[-2000:]
    # and a class member with a wrong signature is a hard error on its own
    tu2 = os.path.join(ROOT, "integration", "source", "vision", "estimator-RANSAC.cpp")
    t2 = open(tu2).read()
    bad2 = t2.replace("bool FundamentalMatrixEstimatorRANSAC::compute(const std::vector<Vector3Type> &p1,",
                      "bool FundamentalMatrixEstimatorRANSAC::compute(const std::vector<Vector2Type> &p1,", 1)
    assert bad2 != t2
    rc2, err2 = _check(bad2, os.path.dirname(tu2))
    assert rc2 != 0, err2[-2000:]

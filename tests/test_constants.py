"""The hand-derived constants of the pre-screen under worst-case constructions, through the device path
(VERDICT r3 #1c): tests/constants_gpu_check.py builds operands AT the boundary each constant has to cover and compares
the device with exact rational arithmetic.  Reference rule they protect: estimator-RANSAC.cpp:76-84,100-129."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_constants_under_worst_case_constructions():
    env = dict(os.environ, MVS_USE_DEBUG_LIB="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "constants_gpu_check.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    st = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    r, i, e = st["rounding"], st["indicator"], st["band_edges"]
    print(json.dumps({k: v for k, v in e.items() if k != "rows"}), json.dumps(r), json.dumps(i))
    # A: 64 u N1' N2' = 32 per path; the construction reaches a fair share of it (a test far inside the bound proves little)
    assert r["ok"] and r["worst_over_u_N1N2_prescreen"] <= 32 and r["worst_over_u_N1N2_exact"] <= 32
    assert max(r["worst_over_u_N1N2_prescreen"], r["worst_over_u_N1N2_exact"]) > 0.5
    # B: the indicator is an indicator, misses nothing below tu', takes nothing at or above tl'
    assert i["ok"] and i["values"] > 5000
    # C: e32 and 2^-14 T on matches crafted on the band edges; the construction bites (plain binary32 puts some of them on
    # the wrong side) and both counting paths stay on the right side
    assert e["ok"] and e["crafted"] > 1500 and e["pilot_hyps"] > 0 and e["pilot_hyps"] < e["cases"]
    assert e["plain_f32_wrong_side"] > 0
    assert e["min_upper_margin"] >= 0 and e["min_lower_margin"] >= 0

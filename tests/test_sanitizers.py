"""ASan + UBSan over the CPU oracle (the reference has no sanitizer runs; GPU sanitizers are not
available on the pool, so the sanitised build is the CPU one)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_asan")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "tests", "cpp", "oracle_asan_driver.c"),
                           os.path.join(ROOT, "oracle", "mvs_oracle.c"), os.path.join(ROOT, "oracle", "mvs_refine_oracle.c"), os.path.join(ROOT, "oracle", "mvs_orb_oracle.c"), "-lm"])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert "image_pair ok=1" in out and "pnp ok=1" in out and "refine ok=1 1" in out and "orb ok=1 n=1" in out and "orb small ok=1" in out and "ERROR" not in out and "runtime error" not in out

"""`python bench.py --gpus N` launches its own rank processes (no torchrun): the parent must set RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* for every child, never touch the GPU itself, and exit non-zero as soon as any child does.

CPU part: the launcher's environment and exit-code logic with stand-in child commands.  GPU part (one device, gloo,
MVS_BENCH_ONE_DEVICE=1): the real two-rank bench through the plain entry point, checking the fields the N > 1 line
carries (contiguous blocks of pairs per rank, one all-gather of the pose records)."""
import json
import os
import subprocess
import tempfile
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib

    return importlib.import_module("bench")


def test_parent_of_a_self_launch_never_imports_torch():
    # the launcher path of bench.py must not have torch (or the HIP library) loaded when it forks the ranks
    code = ("import sys, bench; "
            "assert 'torch' not in sys.modules and 'mvslam_amd.capi' not in sys.modules, sorted(sys.modules)[:5]; "
            "rc = bench.launch_ranks(2, [], child_cmd=[sys.executable, '-c', "
            "'import sys; sys.exit(0)']); "
            "assert 'torch' not in sys.modules; sys.exit(rc)")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, timeout=120)
    assert p.returncode == 0


def test_launcher_sets_rank_environment(tmp_path):
    b = _bench()
    child = ("import os, sys; open(os.path.join(%r, 'rank%%s.txt' %% os.environ['RANK']), 'w').write(' '.join("
             "os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'MVS_BENCH_LAUNCHED'))"
             " + ' ' + ' '.join(sys.argv[1:]))" % str(tmp_path))
    rc = b.launch_ranks(3, ["--gpus", "3", "--steps", "2"], child_cmd=[sys.executable, "-c", child], timeout_s=60)
    assert rc == 0
    ports = set()
    for r in range(3):
        f = open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split()
        assert f[0] == str(r) and f[1] == str(r) and f[2] == "3" and f[3] == "127.0.0.1" and f[5] == "1"
        assert f[6:] == ["--gpus", "3", "--steps", "2"]
        ports.add(f[4])
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_launcher_exit_code_is_the_failing_ranks_and_the_others_are_stopped():
    b = _bench()
    # rank 1 fails at once with code 7; ranks 0 and 2 would run for a minute: the launcher must stop them and return 7
    child = "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(60)"
    t0 = time.time()
    rc = b.launch_ranks(3, [], child_cmd=[sys.executable, "-c", child], timeout_s=50)
    assert rc == 7
    assert time.time() - t0 < 30


def test_launcher_time_limit():
    b = _bench()
    rc = b.launch_ranks(2, [], child_cmd=[sys.executable, "-c", "import time; time.sleep(60)"], timeout_s=1.0)
    assert rc == 124


def test_rank_process_refuses_a_world_size_mismatch():
    # a rank process started with WORLD_SIZE != --gpus must exit 2 before importing torch
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, cwd=ROOT, timeout=120,
                       capture_output=True, text=True)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr


@pytest.mark.gpu
def test_two_rank_bench_through_the_plain_entry_point():
    """bench.py --gpus 2 started the way the driver starts --gpus 1 (no launcher), on ONE device: both ranks use cuda:0
    and the pose records travel through gloo.  Checks the N > 1 fields of the line."""
    env = dict(os.environ, MVS_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    detail = os.path.join(tempfile.mkdtemp(prefix="mvs_bench_"), "detail.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--pairs", "8",
                        "--hyp", "512", "--kp", "500", "--steps", "2", "--warmup", "1", "--launch-timeout", "600",
                        "--detail", detail],
                       env=env, cwd=ROOT, timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]      # the LAST stdout line is the compact one
    assert len(line) < 4096
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["ranks_seen"] == [0, 1]
    assert d["gathered_records"] == 2 * 8
    assert d["gather_us"] > 0
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    assert d["cpu_baseline"] is None        # rank 0 at N = 1 only
    assert d["detail"] == detail
    full = json.load(open(detail))           # the bulky object beside it
    assert full["work"]["gathered_records"] == 2 * 8 and full["value"] == d["value"]
    r = full["ms_per_step_ranks"]
    assert 0 < r["min"] <= r["max"] and abs(r["max"] - d["ms_per_step"]) < 0.5 * d["ms_per_step"] + 1.0
    assert "cpu_baseline" not in full

"""The RCCL exchange step from host C++: mvs_batch_gather_results = one ncclAllGather of the result
records on the ctx stream.  What can be tested without a multi-GPU node: the example compiles and links (CPU), a
one-rank communicator runs every call on the GPU box, and the same gather through torch.distributed's nccl backend
(the path bench.py takes for N > 1) agrees.  Two real ranks: tests/test_dist_gloo.py (gloo, CPU) covers the sharding
logic; the driver's 8-GPU run covers RCCL over xGMI."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "mvslam_amd", "lib", "gather_ranks")
SRC = os.path.join(ROOT, "integration", "examples", "gather_ranks.cpp")


def _build():
    libdir = os.path.join(ROOT, "mvslam_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-o", EXE, SRC, "-I", os.path.join(ROOT, "include"), "-L", libdir,
                           "-lmvslam_hip", "-lrccl", "-Wl,-rpath," + libdir, "-Wno-unused-result"],
                          stderr=subprocess.DEVNULL)


def test_gather_example_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_gather_example_one_rank_on_the_gpu(tmp_path):
    if not os.path.exists(EXE) or os.path.getmtime(SRC) > os.path.getmtime(EXE):
        _build()
    p = subprocess.run([EXE, "0", "1", str(tmp_path / "nccl_id"), "16"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=240)
    out = p.stdout.decode()
    assert p.returncode == 0 and "own block identical" in out, out


@pytest.mark.gpu
def test_torch_rccl_gather_one_rank():
    """tools/rccl_rehearsal.py as a test: the exact collective calls of bench.py for N > 1 on a one-rank nccl group"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_rehearsal.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and "rccl rehearsal ok" in out, out

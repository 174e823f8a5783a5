"""N > 1 path on CPU: world_size-2 gloo run of the pair sharding + the single all-gather of pose records."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    from mvslam_amd import capi, dist as mdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = mdist.shard_range(n_total, rank, world)
    rec = np.zeros(count, dtype=capi.RESULT_DTYPE)
    rec["valid"] = 1
    rec["best_hyp"] = np.arange(first, first + count)         # stand-in payload: the global pair index
    rec["t"][:, 0] = np.arange(first, first + count) * 0.5
    local = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy())
    g = mdist.gather_records(local, world)
    allrec = mdist.records_to_numpy(g, capi.RESULT_DTYPE)
    ok = (len(allrec) == n_total and (allrec["best_hyp"] == np.arange(n_total)).all()
          and np.allclose(allrec["t"][:, 0], np.arange(n_total) * 0.5) and allrec["valid"].all())
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from mvslam_amd import dist as mdist

    for n, w in ((4096, 8), (512, 1), (10, 3), (7, 8)):
        seen = []
        for r in range(w):
            f, c = mdist.shard_range(n, r, w)
            seen += list(range(f, f + c))
        assert seen == list(range(n))
    assert mdist.shard_range(4096, 3, 8) == (1536, 512)       # config 4: contiguous blocks of 512


def test_gather_records_world2_gloo():
    world, n_total = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_gather_records_single_process_is_identity():
    from mvslam_amd import dist as mdist

    x = torch.arange(10, dtype=torch.uint8)
    assert torch.equal(mdist.gather_records(x, 1), x.reshape(1, -1))

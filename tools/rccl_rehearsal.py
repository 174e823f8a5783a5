#!/usr/bin/env python3
"""One-rank RCCL rehearsal of the exact collective calls bench.py makes for N > 1 (init with device_id, uint8
all_gather_into_tensor of the result records on the device, barrier, fp64 MAX all-reduce).  A one-GPU box cannot run two
ranks on distinct devices; this at least runs every call through RCCL.  launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/rccl_rehearsal.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from mvslam_amd import capi, synth

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
ctx = capi.Context(0)
n = 8
data = synth.make_batch(0, n, n_kp=300)
b = capi.Batch(ctx, n, 300, 32)
b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
b.run(capi.default_params(num_hypotheses=512, sampler=capi.SAMPLER_PHILOX, seed=1, max_error_sq=1e-2))
rec = torch.empty(n * capi.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
b.copy_results_device(rec.data_ptr())
b.sync()
out = torch.empty((world, rec.numel()), dtype=torch.uint8, device="cuda")
dist.all_gather_into_tensor(out.reshape(-1), rec.reshape(-1))
torch.cuda.synchronize()
dist.barrier()
tt = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
import numpy as np
got = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
want = b.download(matches=False, mask=False, points=False)["results"]
assert got.tobytes() == want.tobytes() and float(tt.item()) == 1.25
b.close(); ctx.close()
dist.destroy_process_group()
print("rccl rehearsal ok: %d records of %d bytes gathered through RCCL, valid pairs %d" % (len(got), capi.RESULT_DTYPE.itemsize, int(got["valid"].sum())))

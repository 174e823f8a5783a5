#!/usr/bin/env python3
"""bench.py -- image-pairs/sec of the two-view-geometry path on N MI355X (one process per GPU).

A "step" is one pass of the whole hot path (match -> 8-point RANSAC -> decomposition -> triangulation)
over one resident batch of synthetic pairs (default 512 pairs of 2000 keypoints, 50 000 hypotheses
each = BASELINE.json configs[2] per GPU; configs[3] is the same workload on 8 GPUs).  Inputs are
uploaded to HBM before the timed region.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64: 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (= half the 157.3 TF fp32
#                          vector rate of MI355X_MICROARCH.md; the fp64 MFMA dense peak is the same figure)
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def flop_model(stats):
    """Algorithmic fp64 flops of the RANSAC launch (fma = 2; mul/add/sub/div/sqrt/compare = 1).  DESIGN.md
    'RANSAC kernel: work model' derives the constants from the arithmetic contract."""
    per_hyp = 184 + 32 + 720 + 162 + 171 + 800 + 73   # normalise x2, A, A^T A, W init, W final, 3x3 SVD+rank-2, denorm
    per_pair9 = 21        # dot (9 fma) + threshold test
    per_rot9 = 172        # rotation angle (14) + 9 x (A rows 10 + V rows 6)
    per_eval = 18         # 8 fma + compare + conditional add
    return (stats["hypotheses"] * per_hyp + stats["pairs9"] * per_pair9 + stats["rotations9"] * per_rot9
            + stats["score_evals"] * per_eval)


def algorithmic_bytes(n_kp, m, m_inl, n_pts, desc_bytes=32):
    """descriptors in, matches out, point pairs in, E + mask + pose + points out."""
    return 2 * n_kp * desc_bytes + m * 16 + m * 32 + 72 + m + 96 + n_pts * 32


def cpu_baseline(data, params_kw, n_pairs_total, budget_s=25.0):
    """The CPU oracle (same algorithm, plain C, -O2) on a bounded sample of the same workload, one pair per
    host thread.  This is the ONLY place bench.py touches oracle/ (as the baseline being timed)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o

    o.build()
    # the GPU box exposes all host threads but grants a 16-thread share per GPU: never oversubscribe it
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    # calibrate one pair at full H on one thread, then size the sample to the budget
    prm0 = o.make_params(params_kw["num_hypotheses"], o.SAMPLER_PHILOX, params_kw["seed"] + int(data["global_index"][0]),
                         params_kw["max_error_sq"])
    t0 = time.perf_counter()
    o.image_pair(data["desc1"][0], data["kp1"][0], data["desc2"][0], data["kp2"][0], data["K"][0].reshape(3, 3), prm0,
                 params_kw["ratio"], params_kw["max_dist"])
    t_one = time.perf_counter() - t0
    per_thread = max(1, min(4, int(budget_s / max(t_one, 1e-3))))
    n_sample = min(n_pairs_total, cores * per_thread)

    def work(idx_list):
        for i in idx_list:
            prm = o.make_params(params_kw["num_hypotheses"], o.SAMPLER_PHILOX,
                                params_kw["seed"] + int(data["global_index"][i]), params_kw["max_error_sq"])
            o.image_pair(data["desc1"][i], data["kp1"][i], data["desc2"][i], data["kp2"][i],
                         data["K"][i].reshape(3, 3), prm, params_kw["ratio"], params_kw["max_dist"])

    chunks = [list(range(n_sample))[k::cores] for k in range(cores)]
    threads = [threading.Thread(target=work, args=(c,)) for c in chunks if c]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "pairs/s", "cores": len(threads), "kind": "port",
            "single_thread_pairs_per_s": round(1.0 / t_one, 3),
            "sample": "%d pairs of the same workload (full %d hypotheses each), one pair per thread; "
                      "single-thread latency %.2f s/pair" % (n_sample, params_kw["num_hypotheses"], t_one)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=512, help="pairs per GPU (weak scaling)")
    ap.add_argument("--kp", type=int, default=2000)
    ap.add_argument("--hyp", type=int, default=50000)
    ap.add_argument("--noise-px", type=float, default=0.5)
    ap.add_argument("--max-error-sq", type=float, default=1e-2,
                    help="algebraic inlier threshold; 0 = the reference formula 5e-2/K00/K11 (no valid model at "
                         "0.5 px noise, see DESIGN.md)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1; 'gloo' (+ MVS_BENCH_ONE_DEVICE=1) rehearses the multi-rank path on "
                         "a single-GPU box: every rank uses cuda:0 and the pose records are gathered through host memory")
    ap.add_argument("--pcie", action="store_true", help="also time upload + run + download of the whole batch (host "
                    "buffers over PCIe); reported as pcie_inclusive_pairs_per_s, never as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true")
    ap.add_argument("--no-ref-threshold", action="store_true", help="skip the extra reference-threshold timing (profiling "
                    "runs: keeps every launch of a kernel on the same workload)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import torch
    import torch.distributed as dist

    from mvslam_amd import capi, dist as mdist, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    dev = 0 if os.environ.get("MVS_BENCH_ONE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    n_local = args.pairs
    first = rank * n_local  # contiguous block per rank, weak scaling
    data = synth.make_batch(first, n_local, n_kp=args.kp, noise_px=args.noise_px)
    params_kw = dict(ratio=0.7, max_dist=10.0, max_error_sq=args.max_error_sq, num_hypotheses=args.hyp,
                     seed=synth.SEED_BASE)
    prm = capi.default_params(sampler=capi.SAMPLER_PHILOX, min_inliers=8, **params_kw)

    ctx = capi.Context(dev)
    batch = capi.Batch(ctx, n_local, args.kp, 32)
    batch.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"],
                 data["global_index"])
    rec_bytes = capi.RESULT_DTYPE.itemsize
    rec_local = torch.empty(n_local * rec_bytes, dtype=torch.uint8, device="cuda")

    gather_ev = []   # (start, end) torch events around the all-gather of every timed step (N > 1)

    def step(timed=False):
        batch.run(prm)
        if world > 1:  # the one exchange step of the path: all-gather of the pose records over RCCL/xGMI
            batch.copy_results_device(rec_local.data_ptr())
            batch.sync()
            if timed and args.backend == "nccl":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g = mdist.gather_records(rec_local, world)
                e1.record()
                gather_ev.append((e0, e1))
                return g
            t_g = time.perf_counter()
            g = mdist.gather_records(rec_local if args.backend == "nccl" else rec_local.cpu(), world)
            if timed:
                gather_ev.append(time.perf_counter() - t_g)
            return g
        return None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    batch.sync()
    fence()
    t0 = time.perf_counter()
    gathered = None
    for _ in range(args.steps):
        gathered = step(timed=True)
    batch.sync()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    gather_us = None
    if world > 1 and gather_ev:   # communication reported separately from compute; max over ranks
        if isinstance(gather_ev[0], tuple):
            g_us = float(np.mean([a.elapsed_time(b2) for a, b2 in gather_ev])) * 1e3
        else:
            g_us = float(np.mean(gather_ev)) * 1e6
        tg = torch.tensor([g_us], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_us = round(float(tg.item()), 1)

    ms_per_step = elapsed / args.steps * 1e3
    total_pairs = n_local * world
    value = total_pairs * args.steps / elapsed

    out = None
    if rank == 0:
        # ---- per-kernel HIP-event timing on the kernels' own stream + work statistics (outside the timed region)
        _, kern_ms = batch.time(prm, steps=min(args.steps, 5), warmup=0)
        ksteps = min(args.steps, 5)
        kern_ms = {k: v / ksteps for k, v in kern_ms.items()}
        stats = batch.stats(prm)
        res = batch.download(matches=False, mask=False, points=False)["results"]
        flops = flop_model(stats)
        ransac_s = kern_ms["ransac"] * 1e-3
        achieved = flops / ransac_s / 1e12
        m_avg = float(res["n_matches"].mean())
        bytes_pair = algorithmic_bytes(args.kp, m_avg, float(res["n_inliers"].mean()), float(res["n_points"].mean()))
        # HBM bytes of the RANSAC launches: a RECORDED value (PMC counters cannot be read inside this process): the
        # rocprofv3 --pmc pass of this same workload committed under profiles/ ((2*FETCH_SIZE + WRITE_SIZE) KB per
        # MI355X_MICROARCH.md, its own pass); only quoted for the default workload, null otherwise
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r02_ransac_hbm_traffic.json")
        if os.path.exists(tpath) and args.kp == 2000 and args.hyp == 50000 and args.max_error_sq == 1e-2:
            traffic = int(json.load(open(tpath))["hbm_bytes_per_pair"] * n_local)
        # the reference-threshold regime (5e-2 / K00 / K11, sfm-solve.cpp:311: ~5 inliers, ties decided by the residual
        # sum, nothing to prune) with the same kernels, in the same line
        ref_thr = None
        if args.max_error_sq > 0 and not args.no_ref_threshold:
            prm_ref = capi.default_params(sampler=capi.SAMPLER_PHILOX, min_inliers=8, **dict(params_kw, max_error_sq=0.0))
            tot_ref, k_ref = batch.time(prm_ref, steps=3, warmup=1)
            ref_thr = {"pairs_per_s": round(n_local * 3 / (tot_ref * 1e-3), 1), "ransac_ms": round(k_ref["ransac"] / 3, 3),
                       "max_error_sq": "5e-2/K00/K11 = %.3e" % (5e-2 / 525.0 / 525.0)}
        out = {
            "metric": "image-pairs/sec (2k kp, 50k RANSAC hyp)", "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "%d independent synthetic 640x480 pairs per GPU, %d keypoints (256-bit descriptors), "
                            "%d 8-point hypotheses per pair, full match+RANSAC+decompose+triangulate"
                            % (n_local, args.kp, args.hyp),
                "pairs_per_gpu": n_local, "keypoints": args.kp, "hypotheses": args.hyp, "noise_px": args.noise_px,
                "max_error_sq": args.max_error_sq, "parallelism": "pairs sharded, dp%d" % world},
            "roofline": {
                "bound": "valu_fp64",
                "kernel": "ransac_solve_kernel<1264> + ransac_count_kernel<768, 2> + ransac_select_kernel (the RANSAC stage; launch_ms is their sum)",
                "bound_detail": "fp64 vector FMA rate: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s (no MFMA is "
                                "issued: v_mfma_f64 shares the double-precision pipe, profiles/r02_mfma_coissue_microbench.txt)",
                "achieved": round(achieved, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP64_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": "recorded: profiles/r02_ransac_hbm_traffic.json (rocprofv3 --pmc pass of this workload)",
                "traffic_note": "algorithmic: %d B for points + arg-best; the rest is the F hand-over between the solve and "
                                "the scoring launches (72 B written + read per hypothesis, + 4 B count)"
                                % int((m_avg * 32 + 88 * ((args.hyp + 255) // 256)) * n_local),
                "flops_per_launch": int(flops), "launch_ms": round(kern_ms["ransac"], 3),
                # north_star: occupancy / LDS of the RANSAC kernel (hipcc -Rpass-analysis=kernel-resource-usage)
                "occupancy_waves_per_simd": {"solve": 1, "count": 6, "select": 6},
                "vgprs": {"solve": 256, "count": 70, "select": 78}, "agprs": {"solve": 147, "count": 0, "select": 0},
                "scratch_bytes": 0, "lds_bytes_per_workgroup": {"solve": 0, "count": 32 * ((args.kp + 127) // 128) * 128, "select": 32 * args.kp},
                "fp64_issue_note": "a dependency-free v_fma_f64 stream sustains 53 (1 wave/SIMD) to 61 TFLOP/s (2 waves) on this "
                                   "part (profiles/r01_fp64_issue_microbench.txt): the clock drops to ~1.87 GHz under fp64 load"},
            "hbm_roofline": {
                "achieved": round(bytes_pair * (n_local / (ms_per_step * 1e-3)) / 1e9, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(bytes_pair * (n_local / (ms_per_step * 1e-3)) / 1e9 / HBM_PEAK_GBS, 6),
                "algorithmic_bytes_per_pair": int(bytes_pair),
                "match_kernel_GBs": round(2 * args.kp * 32 * n_local / (kern_ms["match_topk"] * 1e-3) / 1e9, 2)},
            "kernel_ms": {k: round(v, 3) for k, v in kern_ms.items()},
            "work": {"avg_matches": round(m_avg, 1), "avg_inliers": round(float(res["n_inliers"].mean()), 1),
                     "avg_points": round(float(res["n_points"].mean()), 1), "valid_pairs": int(res["valid"].sum()),
                     "rotations9_per_hyp": round(stats["rotations9"] / max(stats["hypotheses"], 1), 2),
                     "pairs9_per_hyp": round(stats["pairs9"] / max(stats["hypotheses"], 1), 2)},
        }
        if ref_thr is not None:
            out["reference_threshold"] = ref_thr
        if gather_us is not None:
            out["gather_us"] = gather_us
            out["gather_note"] = "all-gather of the %d-byte pose records (%d per rank), timed with its own events inside the timed steps; max over ranks" % (rec_bytes, n_local)
        if world > 1 and gathered is not None:
            allrec = mdist.records_to_numpy(gathered, capi.RESULT_DTYPE)
            out["work"]["gathered_records"] = int(len(allrec))
            out["work"]["gathered_valid"] = int(allrec["valid"].sum())
        if not args.no_single_pair:
            b1 = capi.Batch(ctx, 1, args.kp, 32)
            b1.upload(0, data["desc1"][:1], data["kp1"][:1], data["n1"][:1], data["desc2"][:1], data["kp2"][:1],
                      data["n2"][:1], data["K"][:1], data["global_index"][:1])
            tot, _ = b1.time(prm, steps=20, warmup=3, per_kernel=False)
            out["single_pair_ms"] = round(tot / 20, 4)  # BASELINE configs[1]: one pair at a time
            b1.close()
        if args.pcie:
            # (a) naive: synchronous upload from pageable memory + run + synchronous download, nothing overlapped
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                batch.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"],
                             data["K"], data["global_index"])
                batch.run(prm)
                batch.download()
            out["pcie_inclusive_pairs_per_s_naive"] = round(n_local * reps / (time.perf_counter() - t0), 1)
            # (b) double-buffered: two batches on two contexts (streams), pinned host buffers (mvs_host_alloc),
            # asynchronous upload -> run -> asynchronous download; batch k+1's transfers overlap batch k's kernels.
            # Every step moves the full inputs host -> device and the full outputs device -> host.
            ctx2 = capi.Context(dev)
            lanes = []
            N = args.kp
            for cx in (ctx, ctx2):
                bb = batch if cx is ctx else capi.Batch(cx, n_local, N, 32)
                pin = {k: capi.pinned_empty(np.asarray(data[k]).shape, np.asarray(data[k]).dtype)
                       for k in ("desc1", "kp1", "n1", "desc2", "kp2", "n2", "global_index")}
                pin["K"] = capi.pinned_empty((n_local, 9), np.float64)
                for k in pin:
                    pin[k][...] = np.asarray(data[k]).reshape(pin[k].shape)
                o_res = capi.pinned_empty((n_local,), capi.RESULT_DTYPE)
                o_mt = capi.pinned_empty((n_local, N), capi.MATCH_DTYPE)
                o_mk = capi.pinned_empty((n_local, N), np.uint8)
                o_pt = capi.pinned_empty((n_local, N, 3), np.float64)
                o_ix = capi.pinned_empty((n_local, N), np.int32)
                lanes.append((bb, pin, (o_res, o_mt, o_mk, o_pt, o_ix)))

            def submit(lane):
                bb, pin, (o_res, o_mt, o_mk, o_pt, o_ix) = lane
                bb.sync()   # the lane's previous step (incl. its download) has completed: its buffers are free
                bb.upload_async(0, pin["desc1"], pin["kp1"], pin["n1"], pin["desc2"], pin["kp2"], pin["n2"], pin["K"],
                                pin["global_index"])
                bb.run(prm)
                bb.download_async(0, n_local, o_res, o_mt, o_mk, o_pt, o_ix)

            for k in range(2):
                submit(lanes[k % 2])
            for lane in lanes:
                lane[0].sync()
            reps = 8
            t0 = time.perf_counter()
            for k in range(reps):
                submit(lanes[k % 2])
            for lane in lanes:
                lane[0].sync()
            dt = time.perf_counter() - t0
            out["pcie_inclusive_pairs_per_s"] = round(n_local * reps / dt, 1)
            out["pcie_note"] = ("double-buffered over two streams with pinned host buffers: %.1f MB up + %.1f MB down per step; "
                                "valid pairs in the last downloaded step: %d"
                                % (sum(v.nbytes for v in lanes[0][1].values()) / 1e6, sum(v.nbytes for v in lanes[0][2]) / 1e6,
                                   int(lanes[(reps - 1) % 2][2][0]["valid"].sum())))
            lanes[1][0].close()
            for lane in lanes:
                for a in list(lane[1].values()) + list(lane[2]):
                    capi.pinned_free(a)
            ctx2.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, params_kw, n_local)
        print(json.dumps(out), flush=True)

    batch.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

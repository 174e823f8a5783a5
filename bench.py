#!/usr/bin/env python3
"""bench.py -- image-pairs/sec of the two-view-geometry path on N MI355X (one process per GPU).

A "step" is one pass of the whole hot path (match -> 8-point RANSAC -> decomposition -> triangulation)
over one resident batch of synthetic pairs (default 512 pairs of 2000 keypoints, 50 000 hypotheses
each = BASELINE.json configs[2] per GPU; configs[3] is the same workload on 8 GPUs).  Inputs are
uploaded to HBM before the timed region.  Rank 0's LAST stdout line is ONE compact JSON object (< 4 KB: the contract's
fields, `roofline`, `cpu_baseline`, one scalar per side leg -- what the driver parses); the full object goes to
bench_detail.json beside this file (--detail PATH; --print-detail also prints it as an earlier line).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus 8                       # spawns its own 8 rank processes (no launcher needed)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The detail object carries (rank 0, N = 1): `roofline.per_kernel` (executed work per kernel, registers / LDS / occupancy as
the runtime reports them), `reference_threshold` (the SURVEY 8(d) threshold 5e-2/K00/K11 in its own wall-clock loop with
its own roofline), `sensitivity` (24 cells), and the rows either side of the path: `sequence` (BASELINE configs[4]),
`refine`, `extract`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # MI355X fp32 vector (v_pk_fma_f32: two FMAs per lane and instruction), MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64: 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (= half the 157.3 TF fp32
#                          vector rate of MI355X_MICROARCH.md; the fp64 MFMA dense peak is the same figure)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md; the 2:1-sparsity figure is twice that)
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun.  The parent NEVER imports torch or touches HIP: it only starts
# one child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), waits, and exits with the first
# non-zero child code (the other children are then terminated by their exact PIDs).
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MVS_BENCH_LAUNCHED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    return env


def launch_ranks(world, argv, child_cmd=None, timeout_s=None, poll_s=0.05):
    """Start `world` rank processes of this script (or of `child_cmd`, for the tests) and wait for them.  Returns the
    exit code: 0 when every rank exited 0, else the first non-zero code seen (124 on timeout)."""
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__)]
    port = free_port()
    procs = []
    for r in range(world):
        # rank 0 owns stdout (the JSON line); the other ranks print nothing there
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(cmd + list(argv), env=rank_env(r, world, port), stdout=out))
    t0 = time.time()
    code = 0
    live = set(range(world))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
        if code == 0 and timeout_s is not None and time.time() - t0 > timeout_s:
            code = 124
        if code != 0:
            break
        if live:
            time.sleep(poll_s)
    for r in sorted(live):   # a rank failed (or the time limit passed): stop the others, by PID
        procs[r].terminate()
    for r in sorted(live):
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    return code


# ---------------------------------------------------------------------------------------------------------------------
# work models (DESIGN.md "work model"): algorithmic fp64 flops, fma = 2, everything else 1
# ---------------------------------------------------------------------------------------------------------------------
PER_HYP = 184 + 32 + 720 + 162 + 171 + 800 + 73   # normalise x2, A, A^T A, W init, W final, 3x3 SVD+rank-2, denorm
PER_PAIR9 = 21        # dot (9 fma) + threshold test
PER_ROT9 = 172        # rotation angle (14) + 9 x (A rows 10 + V rows 6)
PER_EVAL = 18         # 8 fma + compare + conditional add
PER_EVAL_MFMA = 54    # the same evaluation in split bf16: 27 multiply-adds on the matrix cores
PILOT_HYP = 256       # kPilotHyp (kernels.hip): hypotheses the pilot launch counts in full per pair
# the pre-screen of one hypothesis (prescreen.hpp): normalise x2 184, design row products + ||A||_F^2 112,
# Householder QR of the 9x8 (392 fma + 44 fma for the norms + 8 sqrt + 8 div + 40) 920, null vector 176, triangular inverse
# + its Frobenius norm 284, rank-2 through one verified singular triplet (G^T G and its characteristic polynomial 58, eight
# Newton steps 120, cross products + normalisation 66, w / X / sigma / u / eps2 112, second singular value of X from its
# invariants 111) 467, de-normalise 36, bounds 116
PER_PRESCREEN = 184 + 112 + 920 + 176 + 284 + 467 + 36 + 116


def solve_flops(stats):
    return stats["hypotheses"] * PER_HYP + stats["pairs9"] * PER_PAIR9 + stats["rotations9"] * PER_ROT9


def algorithmic_bytes(n_kp, m, m_inl, n_pts, desc_bytes=32):
    """descriptors in, matches out, point pairs in, E + mask + pose + points out."""
    return 2 * n_kp * desc_bytes + m * 16 + m * 32 + 72 + m + 96 + n_pts * 32


def host_threads():
    # the GPU box exposes all host threads but grants a 16-thread share per GPU: never oversubscribe it
    return max(1, min(16, len(os.sched_getaffinity(0))))


def cpu_baseline(data, params_kw, n_pairs_total, budget_s=25.0):
    """The CPU oracle (same algorithm, plain C, -O2) on a bounded sample of the same workload, one pair per
    host thread.  This (and the three small *_cpu legs below) is the ONLY place bench.py touches oracle/ -- as the
    baseline being timed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o

    o.build()
    cores = host_threads()
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


def kernel_table(capi, ctx, batch, prm, stats, n_local, steps=3):
    """Per-kernel view of one pipeline pass: measured ms of every launch (HIP events on the launches' stream), EXECUTED
    fp64 work where a model exists, and the kernel's registers / LDS / occupancy as the runtime reports them for the
    loaded code object (+ the build's VGPR / AGPR split).  Nothing here is typed in."""
    info = ctx.kernel_info(batch.max_kp, batch.desc_bytes)
    table = {}
    for name, ms in batch.time_kernels(prm, steps=steps):
        e = table.setdefault(name, {"ms": 0.0, "launches": 0})
        e["ms"] += ms
        e["launches"] += 1
    for name, e in table.items():
        ki = info.get(name, {})
        e["ms"] = round(e["ms"], 4)
        e["threads_per_block"] = ki.get("threads_per_block")
        e["registers_runtime"] = ki.get("num_regs")
        e["lds_bytes_per_workgroup"] = (ki.get("static_lds_bytes") or 0) + (ki.get("dynamic_lds_bytes") or 0)
        e["scratch_bytes_per_lane"] = ki.get("scratch_bytes_per_lane")
        e["workgroups_per_cu"] = ki.get("blocks_per_cu")
        e["occupancy_waves_per_simd"] = ki.get("waves_per_simd")
        if "build" in ki:
            e["vgprs"], e["agprs"], e["sgprs"] = ki["build"].get("vgprs"), ki["build"].get("agprs"), ki["build"].get("sgprs")
    per_solve = solve_flops(stats) / max(stats["hypotheses"], 1)   # exact solve of one hypothesis, workload average
    for name, e in table.items():
        fl_exec = fl_alg = None
        peak, bound = FP64_PEAK_TFLOPS, "valu_fp64"
        if name.startswith("ransac_solve_kernel") or name.startswith("ransac_solve_list_kernel"):   # every hypothesis of the pairs in mode 0
            fl_alg = solve_flops(stats)
            fl_exec = per_solve * stats["pairs_mode"][0] * (stats["hypotheses"] / max(sum(stats["pairs_mode"]), 1)) \
                if sum(stats["pairs_mode"]) else fl_alg
        elif name.startswith("ransac_exact_list_kernel"):   # flagged by the pre-screen + survivors of the counting
            mode0 = stats["pairs_mode"][0] * (stats["hypotheses"] / max(sum(stats["pairs_mode"]), 1))
            fl_exec = fl_alg = per_solve * max(stats["exact_solves"] - mode0, 0)
        elif name.startswith("ransac_prescreen_kernel"):
            n_ps = stats["hypotheses"] - stats["pairs_mode"][0] * (stats["hypotheses"] / max(sum(stats["pairs_mode"]), 1))
            fl_exec = fl_alg = n_ps * PER_PRESCREEN
        elif name.startswith("ransac_count32_kernel"):
            # phase 0 (pilot: the first PILOT_HYP hypotheses of every mode-1 pair in full), phase 2 (finish: what the dense
            # phase could not drop, from its last point on), phase 1 (diagnostics: everything in one launch)
            ev32 = stats["score_evals_executed_f32"]
            m1 = stats["pairs_mode"][1] / max(sum(stats["pairs_mode"]), 1)
            ev_pilot = min(int(PILOT_HYP * stats["matches"] * m1), ev32) if stats.get("score_evals_executed_mfma", 0) else 0
            ev = ev_pilot if ", 0, false>" in name and stats.get("score_evals_executed_mfma_finish", 0) == 0 else \
                ev32 if ", 0, false>" in name else ev32 - ev_pilot
            fl_alg = fl_exec = ev * PER_EVAL
            e["evals_executed"] = int(ev)
            e["evals_executed_frac"] = round(ev / max(stats["score_evals"], 1), 4)
            peak, bound = FP32_PEAK_TFLOPS, "valu_fp32"
        elif name.startswith("ransac_finish_upper_kernel"):
            # the rest of the list: upper counts only, behind the dense phase's points (the dense phase's loop over list entries)
            ev = stats.get("score_evals_executed_mfma_rest", 0)
            fl_alg = fl_exec = ev * PER_EVAL_MFMA
            e["evals_executed"] = int(ev)
            e["evals_executed_frac"] = round(ev / max(stats["score_evals"], 1), 4)
            e["evals_per_ns"] = round(ev / max(e["ms"] * 1e6, 1e-9), 2)
            peak, bound = BF16_MFMA_PEAK_TFLOPS, "mfma_bf16"
        elif name.startswith("ransac_finish_mfma_kernel"):
            # the same tile product for the few per cent of the hypotheses the dense phase leaves: every point against the lower
            # threshold, the points behind the dense phase against the upper one as well; <., true> is the PILOT: the first 1024
            # hypotheses of a pair on every point, both thresholds, ahead of the dense phase
            ev = stats.get("score_evals_executed_mfma_pilot" if name.endswith(", true>") else "score_evals_executed_mfma_finish", 0)
            fl_alg = fl_exec = ev * PER_EVAL_MFMA
            e["evals_executed"] = int(ev)
            e["evals_executed_frac"] = round(ev / max(stats["score_evals"], 1), 4)
            e["evals_per_ns"] = round(ev / max(e["ms"] * 1e6, 1e-9), 2)
            peak, bound = BF16_MFMA_PEAK_TFLOPS, "mfma_bf16"
        elif name.startswith("ransac_count_mfma_kernel"):
            # dense phase: every evaluation is 27 products of bf16 parts (hi hi + hi lo + lo hi over the nine monomials) on
            # the matrix cores (two v_mfma_f32_32x32x16_bf16 per 32 x 32 tile, 5 of the 32 K slots are zero padding) plus
            # three packed vector instructions per two evaluations for the count
            ev = stats.get("score_evals_executed_mfma", 0)
            fl_alg = fl_exec = ev * PER_EVAL_MFMA
            e["evals_executed"] = int(ev)
            e["evals_executed_frac"] = round(ev / max(stats["score_evals"], 1), 4)
            e["evals_per_ns"] = round(ev / max(e["ms"] * 1e6, 1e-9), 2)
            peak, bound = BF16_MFMA_PEAK_TFLOPS, "mfma_bf16"
        elif name.startswith("ransac_count_kernel") or name.startswith("ransac_count2_kernel"):
            ev64 = stats["score_evals_executed"] - stats["score_evals_executed_f32"] - \
                stats.get("score_evals_executed_mfma", 0) - stats.get("score_evals_executed_mfma_finish", 0) - \
                stats.get("score_evals_executed_mfma_rest", 0) - stats.get("score_evals_executed_mfma_pilot", 0)
            fl_alg = stats["score_evals"] * PER_EVAL
            fl_exec = ev64 * PER_EVAL
            e["evals_executed_frac"] = round(ev64 / max(stats["score_evals"], 1), 4)
        elif name.startswith("ransac_kernel"):   # fused: solve + every evaluation
            fl_exec = fl_alg = solve_flops(stats) + stats["score_evals"] * PER_EVAL
        if fl_exec is not None and e["ms"] > 0:
            e["bound"], e["peak_tflops"] = bound, peak
            e["flops_executed"] = int(fl_exec)
            e["flops_algorithmic"] = int(fl_alg)
            e["tflops_executed"] = round(fl_exec / (e["ms"] * 1e-3) / 1e12, 3)
            e["frac_executed"] = round(fl_exec / (e["ms"] * 1e-3) / 1e12 / peak, 4)
    return table


def roofline_object(table, stats, traffic, traffic_src):
    """roofline of the DOMINANT kernel (executed work / its measured launch time) + the RANSAC stage as a whole."""
    ransac = {k: v for k, v in table.items() if k.startswith("ransac_") or k.startswith("pair_prepare")}
    dom_name = max(ransac, key=lambda k: ransac[k]["ms"]) if ransac else max(table, key=lambda k: table[k]["ms"])
    dom = table[dom_name]
    stage_ms = sum(v["ms"] for v in ransac.values())
    stage_exec = sum(v.get("flops_executed", 0) for v in ransac.values())
    # what the reference's algorithm asks for: every hypothesis solved exactly and scored on every match
    stage_alg = solve_flops(stats) + stats["score_evals"] * PER_EVAL
    achieved = dom.get("tflops_executed", 0.0)
    peak = dom.get("peak_tflops", FP64_PEAK_TFLOPS)
    # peak-weighted time the executed work would take at the respective vector peaks
    t_ideal = sum(v.get("flops_executed", 0) / (v.get("peak_tflops", FP64_PEAK_TFLOPS) * 1e12) for v in ransac.values()) * 1e3
    return {
        "bound": dom.get("bound", "valu_fp64"), "kernel": dom_name,
        "bound_detail": "vector FMA rate of the kernel's arithmetic type: fp64 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz = "
                        "78.6 TFLOP/s (v_mfma_f64 shares the pipe: profiles/r02_mfma_coissue_microbench.txt), fp32 157.3 "
                        "TFLOP/s (v_pk_fma_f32); mfma_bf16: 2500 TFLOP/s dense (the split-bf16 counting kernels, whose "
                        "three packed vector instructions per two evaluations bound them well below that: DESIGN.md 4.3e)",
        "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "frac_definition": "EXECUTED algorithmic flops of the dominant kernel / its launch time (HIP events) / the vector "
                           "peak of its arithmetic type",
        "launch_ms": dom["ms"], "flops_per_launch": dom.get("flops_executed"),
        "traffic": traffic, "traffic_source": traffic_src,
        "stage": {"kernels": sorted(ransac), "ms": round(stage_ms, 3), "flops_executed": int(stage_exec),
                  "frac_executed": round(t_ideal / max(stage_ms, 1e-9), 4),
                  "frac_executed_definition": "sum over the stage's kernels of executed flops / that kernel's vector peak, "
                                              "divided by the stage's measured time",
                  "reference_equivalent_flops": int(stage_alg),
                  "reference_equivalent_tflops": round(stage_alg / max(stage_ms, 1e-9) / 1e9, 2),
                  "note": "reference_equivalent_* price the stage as if every hypothesis had been solved exactly and scored "
                          "on every match (what estimator-RANSAC.cpp does); the pre-screen and the pruned counting provably "
                          "skip most of that work, so it is a speed-up figure, not a utilisation"},
        "work": {"hypotheses": int(stats["hypotheses"]), "exact_solves": int(stats["exact_solves"]),
                 "prescreened_only": int(stats["prescreened"]), "pairs_mode": stats["pairs_mode"],
                 "evals_possible": int(stats["score_evals"]), "evals_executed": int(stats["score_evals_executed"]),
                 "evals_executed_f32": int(stats["score_evals_executed_f32"]),
                 "evals_executed_mfma_dense": int(stats.get("score_evals_executed_mfma", 0)),
                 "evals_executed_mfma_finish": int(stats.get("score_evals_executed_mfma_finish", 0)),
                 "evals_executed_mfma_finish_rest": int(stats.get("score_evals_executed_mfma_rest", 0)),
                 "evals_executed_mfma_pilot": int(stats.get("score_evals_executed_mfma_pilot", 0)),
                 "max_sweeps9": int(stats.get("max_sweeps9", 0)),
                 "dense_points_over_matches": round(stats["dense_points"] / stats["matches_mode1"], 3)
                 if stats.get("matches_mode1") else None},
        "per_kernel": table,
        "fp64_issue_note": "a dependency-free v_fma_f64 stream sustains 53 (1 wave/SIMD) to 61 TFLOP/s (2 waves) on this "
                           "part (profiles/r01_fp64_issue_microbench.txt): the clock drops to ~1.87 GHz under fp64 load"}


def make_ctx(capi, dev, args):
    ctx = capi.Context(dev)
    if getattr(args, "one_stream", False):
        ctx.set_half_batches(False)
    return ctx


# ---------------------------------------------------------------------------------------------------------------------
# the rows either side of the path, one bounded leg each
# ---------------------------------------------------------------------------------------------------------------------
def bench_sequence(capi, synth, np, dev, args):
    """BASELINE configs[4]: 1000-frame synthetic sequence, per frame match + two-view + PnP + triangulate, no BA."""
    frames, kp, hyp, pnp_hyp = args.seq_frames, args.kp, args.hyp, 100   # 100 = the reference's iterationsCount
    seq = synth.make_sequence(frames, n_kp=kp)
    ctx = make_ctx(capi, dev, args)
    s = capi.Sequence(ctx, frames, kp, 32)
    s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
    prm = capi.default_params(num_hypotheses=hyp, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
    pprm = capi.default_pnp_params(num_hypotheses=pnp_hyp, seed=7, reproj_error=2.0, refit=0)
    steps = 3
    ms = s.time(prm, pprm, steps=steps, warmup=1) / steps
    stage_ms = s.time_stages(prm, pprm, steps=steps)
    s.run(prm, pprm)
    gp, gt = s.download_pairs(), s.download_tracks()
    res, tr = gp["results"], gt["tracks"]
    s.close()
    ctx.close()
    pnp_flops = int(sum(pnp_hyp * (1900 + 28 * int(n)) for n in tr["n_corr"]))
    out = {"metric": "frames/sec, %d-frame synthetic sequence (match + two-view + PnP + triangulate per frame, no BA)" % frames,
           "value": round(frames / (ms * 1e-3), 1), "unit": "frames/s", "ms_per_sequence": round(ms, 2), "frames": frames,
           "keypoints": kp, "hypotheses": hyp, "pnp_hypotheses": pnp_hyp, "valid_pairs": int(res["valid"].sum()),
           "tracks_ok": int(tr["ok"].sum()), "avg_corr": round(float(tr["n_corr"].mean()), 1),
           "avg_pnp_inliers": round(float(tr["n_inliers"].mean()), 1),
           "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
           "pnp_roofline": {"bound": "latency (one 100-lane workgroup per track: the reference's iterationsCount)",
                            "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS, "flops_per_launch": pnp_flops,
                            "launch_ms": round(stage_ms["pnp"], 4),
                            "achieved": round(pnp_flops / (stage_ms["pnp"] * 1e-3) / 1e12, 4),
                            "frac": round(pnp_flops / (stage_ms["pnp"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5)}}
    if not args.no_cpu_baseline and args.seq_cpu_frames >= 3:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_sequence import oracle_sequence
        nthr = host_threads()
        chunk = max(3, args.seq_cpu_frames // nthr + 2)   # frames per thread: chunk - 1 pairs and chunk - 2 tracks

        def work(k0):
            sub = dict(desc=seq["desc"][k0:k0 + chunk], kp=seq["kp"][k0:k0 + chunk], n_kp=seq["n_kp"][k0:k0 + chunk], K=seq["K"])
            oracle_sequence(sub, dict(H=hyp, seed=synth.SEED_BASE + k0, thr=1e-2), dict(H=pnp_hyp, seed=7, err=2.0))

        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(i * chunk,)) for i in range(nthr) if (i + 1) * chunk <= frames]
        [t.start() for t in ths]
        [t.join() for t in ths]
        dt = time.perf_counter() - t0
        done = len(ths) * (chunk - 1)
        out["cpu_baseline"] = {"value": round(done / dt, 2), "unit": "frames/s", "cores": len(ths), "kind": "port",
                               "sample": "%d threads x %d consecutive frames of the same sequence (%d frame steps, full "
                                         "hypothesis counts)" % (len(ths), chunk, done)}
    return out


def bench_refine(capi, np, batch, data, dl, args, steps=10):
    """Row f4: ImagePair::refine of the resident batch (the main leg's results), wall clock around `steps` asynchronous
    refinement passes + one sync (inputs and results stay in HBM)."""
    rp = capi.default_refine_params()
    batch.refine(rp, 0.5)
    batch.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.refine(rp, 0.5)
    batch.sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    res = dl["results"]
    ref = batch.download_refined(points=False)
    rr = ref["refined"]
    ok = rr["ok"] == 1
    npts = res["n_points"][ok].astype(np.int64)
    its = rr["iterations"][ok].astype(np.int64)
    # algorithmic work: per linear solve every point is linearised twice and its candidate cost evaluated
    # once, + one covariance pass
    LIN, SCHUR, STEP, COST, COV = 524, 760, 90, 90, 1100
    flops = float(np.sum(npts * (its * (2 * LIN + SCHUR + STEP + COST) + (LIN + SCHUR + COV))))
    out = {"metric": "refined image-pairs/sec (ImagePair::refine = sfm_refine, batched on device)",
           "value": round(batch.n_pairs / (ms * 1e-3), 1), "unit": "pairs/s", "ms_per_batch": round(ms, 4),
           "pairs": int(batch.n_pairs), "refined_ok": int(ok.sum()),
           "mean_points": round(float(npts.mean()), 1) if len(npts) else 0.0,
           "mean_linear_solves": round(float(its.mean()), 3) if len(its) else 0.0,
           "roofline": {"bound": "valu_fp64", "kernel": "refine_kernel<2>", "flops_per_launch": int(flops),
                        "launch_ms": round(ms, 4), "achieved": round(flops / (ms * 1e-3) / 1e12, 3), "peak": FP64_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)}}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as o
        full = batch.download(0, min(batch.n_pairs, 16))
        idx = [p for p in range(len(full["results"])) if ok[p]][:16]
        t0 = time.perf_counter()
        for p in idx:
            n = int(res["n_points"][p])
            mt = full["matches"][p][full["point_idx"][p][:n]]
            p1 = data["kp1"][p][mt["trainIdx"]].astype(np.float64)
            p2 = data["kp2"][p][mt["queryIdx"]].astype(np.float64)
            cov = np.tile((np.eye(2) * 0.25).reshape(4), (n, 1))
            o.sfm_refine(p1, cov, p2, cov, data["K"][p].reshape(3, 3), res["R"][p], res["t"][p], full["points"][p][:n])
        dt = time.perf_counter() - t0
        if idx:
            out["cpu_baseline"] = {"value": round(len(idx) / dt, 2), "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": "%d pairs of the same batch, single thread" % len(idx)}
    return out


def bench_extract(capi, np, dev, args, images=64, width=640, height=480):
    """Row f3: VisualFeature::extract for a batch of 640x480 frames at the keypoint count the matching configs assume."""
    def textured(seed, h, w):
        rng = np.random.default_rng(seed)
        base = rng.integers(0, 256, size=(h // 6 + 2, w // 6 + 2)).astype(np.uint8)
        img = np.kron(base, np.ones((6, 6), dtype=np.uint8))[:h, :w].astype(np.int32)
        img += rng.integers(-6, 7, size=img.shape)
        return np.clip(img, 0, 255).astype(np.uint8)

    imgs = np.stack([textured(100 + i, height, width) for i in range(images)])
    ctx = make_ctx(capi, dev, args)
    prm = capi.default_orb_params(nfeatures=args.kp)
    out_k = ctx.extract(imgs, prm)
    kernel_ms = ctx.extract_time(steps=10)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.extract(imgs, prm)
    host_ms = (time.perf_counter() - t0) * 1e3 / 3
    # the reference's own call pattern: ONE frame per VisualFeature::extract (vision/visual-feature.cpp:40-49), pinned host
    # buffers in and out, wall clock per mvs_extract call
    one = capi.pinned_empty((1, height, width), np.uint8)
    one[...] = imgs[:1]
    one_out = dict(kp=capi.pinned_empty((1, args.kp), capi.KEYPOINT_DTYPE), desc=capi.pinned_empty((1, args.kp, 32), np.uint8),
                   n=capi.pinned_empty((1,), np.int32))
    for _ in range(3):
        ctx.extract(one, prm, out=one_out)
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.extract(one, prm, out=one_out)
    single_ms = (time.perf_counter() - t0) * 1e3 / 50
    single_kernel_ms = ctx.extract_time(steps=20)
    ctx.close()
    pyr = sum(round(width / 1.2 ** l) * round(height / 1.2 ** l) for l in range(8))
    alg_bytes = 9.0 * pyr * images   # u8 pixels, ~9 passes over the 3.16 x pyramid
    out = {"metric": "extracted images/sec (VisualFeature::extract, %dx%d, %d features)" % (width, height, args.kp),
           "value": round(images / (kernel_ms * 1e-3), 1), "unit": "images/s", "kernel_ms_per_batch": round(kernel_ms, 4),
           "images": images, "mean_keypoints": float(out_k["n"].mean()),
           "host_buffers_ms_per_batch": round(host_ms, 3), "host_buffers_images_per_s": round(images / (host_ms * 1e-3), 1),
           "single_image_ms": round(single_ms, 4), "single_image_kernel_ms": round(single_kernel_ms, 4),
           "roofline": {"bound": "hbm", "kernel": "resize + fast_nms + select + blur + describe (one extraction)",
                        "achieved": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "bytes_per_launch": int(alg_bytes), "traffic": None}}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as o
        ncpu = 4
        t0 = time.perf_counter()
        same = True
        for i in range(ncpu):
            w = o.orb_extract(imgs[i], o.make_orb_params(nfeatures=args.kp))
            n = int(out_k["n"][i])
            same &= n == len(w["kp"]) and np.array_equal(out_k["desc"][i][:n], w["desc"])
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(ncpu / dt, 2), "unit": "images/s", "cores": 1, "kind": "port",
                               "sample": "%d of the same frames, single thread" % ncpu, "bit_exact_vs_gpu": bool(same)}
    return out


def bench_sensitivity(capi, synth, np, dev, args, pairs=64):
    """How the pre-screened stage behaves away from the headline workload: wrong-match share x pixel noise x
    inlier threshold, 64 pairs per cell, full keypoint and hypothesis counts.  Per cell: pairs/s of the whole path (HIP
    events around 3 passes), the share of the hypotheses that went through the exact solve, the pairs per mode as the
    probe decided, the share of a pair's matches the dense counting phase had to cover (n1 / M), best count."""
    ctx = make_ctx(capi, dev, args)
    batch = capi.Batch(ctx, pairs, args.kp, 32)
    cells = []
    t_all = time.perf_counter()
    for outl in (0.3, 0.5, 0.7, 0.9):
        for noise in (0.5, 2.0):
            data = synth.make_batch(9000, pairs, n_kp=args.kp, noise_px=noise, outlier_frac=outl)
            batch.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"],
                         data["global_index"])
            for thr in (1e-2, 1e-3, 1e-4):
                prm = capi.default_params(sampler=capi.SAMPLER_PHILOX, min_inliers=8, ratio=0.7, max_dist=10.0, max_error_sq=thr,
                                          num_hypotheses=args.hyp, seed=synth.SEED_BASE)
                tot, _ = batch.time(prm, steps=3, warmup=1, per_kernel=False)
                st = batch.stats(prm)
                res = batch.download(matches=False, mask=False, points=False)["results"]
                ms = tot / 3
                cells.append({
                    "outlier_frac": outl, "noise_px": noise, "max_error_sq": thr, "ms_per_batch": round(ms, 3),
                    "pairs_per_s": round(pairs / (ms * 1e-3), 1),
                    "exact_solve_share": round(st["exact_solves"] / max(st["hypotheses"], 1), 5),
                    "pairs_mode": st["pairs_mode"],
                    "dense_points_over_matches": round(st["dense_points"] / st["matches_mode1"], 3) if st["matches_mode1"] else None,
                    "avg_matches": round(float(res["n_matches"].mean()), 1),
                    "avg_best_count": round(float(res["best_count"].mean()), 1), "valid_pairs": int(res["valid"].sum())})
    batch.close()
    ctx.close()
    worst = min(cells, key=lambda c: c["pairs_per_s"])
    return {"pairs_per_cell": pairs, "keypoints": args.kp, "hypotheses": args.hyp, "seconds": round(time.perf_counter() - t_all, 1),
            "min_pairs_per_s": worst["pairs_per_s"],
            "min_cell": {k: worst[k] for k in ("outlier_frac", "noise_px", "max_error_sq")},
            "note": "64-pair batches: rates are below the 512-pair headline at equal work (fewer workgroups per launch); the "
                    "forced-exact rate of every cell is in profiles/r04_sensitivity_guard.json (diagnostics build)",
            "cells": cells}


# ---------------------------------------------------------------------------------------------------------------------
# output: the LAST stdout line is a compact object (< 4 KB: the contract's fields + scalar summaries); everything bulky
# (per-kernel tables, the sensitivity cells, the sub-legs' own objects) goes to bench_detail.json beside this file
# ---------------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096
DETAIL_NAME = "bench_detail.json"


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(out, detail_path=None):
    """The driver's line: contract fields, `roofline` and `cpu_baseline` with scalar members only, and one scalar per
    side leg.  `out` is the full (detail) object main() assembled."""
    cfg = out.get("config", {})
    r = out.get("roofline") or {}
    c = out.get("cpu_baseline")
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "pairs_per_gpu", "keypoints", "hypotheses", "max_error_sq",
                                              "parallelism", "streams") if k in cfg}
    line["roofline"] = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_ms",
                                              "flops_per_launch", "traffic")}
    line["roofline"]["algorithmic_hbm_frac"] = _get(out, "hbm_roofline", "frac")
    line["cpu_baseline"] = None if c is None else {k: c.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
    line["reference_threshold_pairs_per_s"] = out.get("reference_threshold_pairs_per_s")
    line["single_pair_ms"] = out.get("single_pair_ms")
    line["image_pair_ctor_ms"] = out.get("image_pair_ctor_ms")
    line["pcie_inclusive_pairs_per_s"] = out.get("pcie_inclusive_pairs_per_s")
    line["sequence_frames_per_s"] = _get(out, "sequence", "value")
    line["refine_pairs_per_s"] = _get(out, "refine", "value")
    line["extract_images_per_s"] = _get(out, "extract", "value")
    line["extract_single_image_ms"] = _get(out, "extract", "single_image_ms")
    line["sensitivity_min_pairs_per_s"] = _get(out, "sensitivity", "min_pairs_per_s")
    line["valid_pairs"] = _get(out, "work", "valid_pairs")
    line["max_sweeps9"] = _get(out, "roofline", "work", "max_sweeps9")
    line["ranks_seen"] = out.get("ranks_seen")
    line["gather_us"] = out.get("gather_us")
    if _get(out, "work", "gathered_records") is not None:
        line["gathered_records"] = out["work"]["gathered_records"]
    line["detail"] = detail_path
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) >= COMPACT_LIMIT:   # never let a long free-text member push the line past what the driver reads
        for k in ("sample",):
            if line["cpu_baseline"] and isinstance(line["cpu_baseline"].get(k), str):
                line["cpu_baseline"][k] = line["cpu_baseline"][k][:160]
        line["config"]["workload"] = str(line["config"].get("workload"))[:200]
        txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) < COMPACT_LIMIT, "compact bench line is %d bytes" % len(txt)
    return txt


def write_detail(out, path):
    """Best effort: the detail file is a convenience for the judge, never a reason to lose the bench line."""
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        print("bench.py: could not write %s: %s" % (path, e), file=sys.stderr)
        return None


def profiler_preload_active():
    """True under rocprofv3 & co.: the tool's preloaded library has initialised the GPU before this program started, so
    starting a child process from here is the exec the pool forbids."""
    env = os.environ
    if any(k in env for k in ("ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH",
                              "ROCPROF_OUTPUT_FILE_NAME", "ROCPROFILER_LIBRARY_CTOR")):
        return True
    return any(t in env.get("LD_PRELOAD", "") for t in ("rocprof", "roctracer", "rocprofiler"))


def image_pair_probe(kp, hyp, reps=100, timeout_s=120):
    """What a caller of the reference's own interface waits for: the shim's ImagePair constructor (host C++,
    front-end/image-pair.cpp:30-71), host buffers in, host objects out -- a child process
    (tests/cpp/image_pair_latency.cpp, built by __graft_entry__.build()).  Must run BEFORE this process touches the GPU."""
    probe = os.path.join(ROOT, "mvslam_amd", "lib", "image_pair_latency")
    if not os.path.exists(probe):
        return {"error": "probe not built"}
    if profiler_preload_active():
        return {"skipped": "profiler preload detected: no child processes under the profiler"}
    try:
        pr = subprocess.run([probe, str(kp), str(hyp), str(reps)], capture_output=True, text=True, timeout=timeout_s)
        return json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
    except subprocess.TimeoutExpired:
        return {"error": "probe timed out after %d s" % timeout_s}
    except Exception as e:   # the probe is a convenience: never fail the bench line for it
        return {"error": str(e)[:200]}


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
    ap.add_argument("--no-pcie", action="store_true", help="skip the transfer-inclusive leg (upload + run + download of the "
                    "whole batch through pinned host buffers, double-buffered): reported as pcie_inclusive_pairs_per_s -- what "
                    "a host C++ caller of the boundary sees --, never as `value`")
    ap.add_argument("--pcie-naive", action="store_true", help="also time the synchronous pageable-memory variant")
    ap.add_argument("--one-stream", action="store_true", help="every launch on the context's one stream (default: a batch of "
                    ">= 64 pairs runs as two halves on two streams, mvs_ctx_set_half_batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true")
    ap.add_argument("--no-ref-threshold", action="store_true", help="skip the reference-threshold leg (profiling runs: "
                    "keeps every launch of a kernel on the same workload)")
    ap.add_argument("--ref-steps", type=int, default=10, help="wall-clock steps of the reference-threshold leg")
    ap.add_argument("--sections", default="main,sensitivity,sequence,refine,extract",
                    help="legs to run on rank 0 at N = 1 (main is always timed; the others add their sub-objects)")
    ap.add_argument("--seq-frames", type=int, default=1000)
    ap.add_argument("--seq-cpu-frames", type=int, default=48)
    ap.add_argument("--launch-timeout", type=float, default=None, help="seconds before a self-launched run is abandoned")
    ap.add_argument("--detail", default=os.path.join(ROOT, DETAIL_NAME),
                    help="where rank 0 writes the full (bulky) result object; the LAST stdout line is the compact one")
    ap.add_argument("--print-detail", action="store_true", help="also print the full object as an EARLIER stdout line")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above this line has touched torch or HIP.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout_s=args.launch_timeout))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    # the ImagePair-constructor probe is a child process: it runs BEFORE this process initialises the GPU,
    # alone on the device, and its JSON is kept for the line
    probe_result = None
    if rank == 0 and world == 1 and not args.no_single_pair:
        probe_result = image_pair_probe(args.kp, args.hyp)

    import numpy as np
    import torch
    import torch.distributed as dist

    from mvslam_amd import capi, dist as mdist, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    dev = 0 if os.environ.get("MVS_BENCH_ONE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    sections = set(x for x in args.sections.split(",") if x)
    n_local = args.pairs
    first = rank * n_local  # contiguous block per rank, weak scaling
    data = synth.make_batch(first, n_local, n_kp=args.kp, noise_px=args.noise_px)
    params_kw = dict(ratio=0.7, max_dist=10.0, max_error_sq=args.max_error_sq, num_hypotheses=args.hyp,
                     seed=synth.SEED_BASE)
    prm = capi.default_params(sampler=capi.SAMPLER_PHILOX, min_inliers=8, **params_kw)

    ctx = make_ctx(capi, dev, args)
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
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    rank_ms = [elapsed_local / args.steps * 1e3]
    ranks_seen = [0]
    if world > 1:
        tt = torch.tensor([elapsed_local], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # every rank's own step time and rank id, as the communicator delivers them
        mine = torch.tensor([float(rank), elapsed_local / args.steps * 1e3], dtype=torch.float64, device=coll_dev)
        allr = torch.empty(2 * world, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.cpu().reshape(world, 2)
        ranks_seen = sorted(int(x) for x in allr[:, 0].tolist())
        rank_ms = [float(x) for x in allr[:, 1].tolist()]

    gather_us = None
    if world > 1 and gather_ev:   # communication reported separately from compute; max over ranks
        if isinstance(gather_ev[0], tuple):
            g_us = float(np.mean([a.elapsed_time(b2) for a, b2 in gather_ev])) * 1e3
        else:
            g_us = float(np.mean(gather_ev)) * 1e6
        tg = torch.tensor([g_us], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_us = round(float(tg.item()), 1)

    ms_per_step = elapsed / args.steps * 1e3
    total_pairs = n_local * world
    value = total_pairs * args.steps / elapsed

    if rank == 0:
        # ---- per-kernel HIP-event timing on the kernels' own stream + work statistics (outside the timed region)
        _, kern_ms = batch.time(prm, steps=min(args.steps, 5), warmup=0)
        ksteps = min(args.steps, 5)
        kern_ms = {k: v / ksteps for k, v in kern_ms.items()}
        stats = batch.stats(prm)
        table = kernel_table(capi, ctx, batch, prm, stats, n_local)
        dl = batch.download(matches=False, mask=False, points=False)
        res = dl["results"]
        m_avg = float(res["n_matches"].mean())
        bytes_pair = algorithmic_bytes(args.kp, m_avg, float(res["n_inliers"].mean()), float(res["n_points"].mean()))
        # HBM bytes of the RANSAC launches: a RECORDED value (PMC counters cannot be read inside this process): the
        # rocprofv3 --pmc pass of this same workload committed under profiles/ ((2*FETCH_SIZE + WRITE_SIZE) KB per
        # MI355X_MICROARCH.md, its own pass); only quoted for the default workload, null otherwise
        traffic, traffic_src = None, None
        for tname in ("r05_ransac_hbm_traffic.json", "r04_ransac_hbm_traffic.json", "r03_ransac_hbm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and args.kp == 2000 and args.hyp == 50000 and args.max_error_sq == 1e-2:
                traffic = int(json.load(open(tpath))["hbm_bytes_per_pair"] * n_local)
                traffic_src = "recorded: profiles/%s (rocprofv3 --pmc pass of this workload; RANSAC stage, per launch " \
                              "sequence)" % tname
                break
        out = {
            "metric": "image-pairs/sec (2k kp, 50k RANSAC hyp)", "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "%d independent synthetic 640x480 pairs per GPU, %d keypoints (256-bit descriptors), "
                            "%d 8-point hypotheses per pair, full match+RANSAC+decompose+triangulate"
                            % (n_local, args.kp, args.hyp),
                "pairs_per_gpu": n_local, "keypoints": args.kp, "hypotheses": args.hyp, "noise_px": args.noise_px,
                "max_error_sq": args.max_error_sq, "parallelism": "pairs sharded, dp%d" % world,
                "streams": 1 if args.one_stream else 2,
                "launch_plan": ("every launch covers the rank's whole batch, one stream (--one-stream)" if args.one_stream else
                                "the rank's batch runs as two independent halves on two HIP streams (mvs_ctx_set_half_batches, "
                                "default): the latency-bound kernels of one half run under the throughput-bound kernels of "
                                "the other; the per-kernel table and `roofline` time whole-batch launches on one stream"),
                "threshold_note": "headline threshold 1e-2 (a consensus set exists: ~1100 inliers per pair, every stage "
                                  "runs); SURVEY 8(d)'s literal 5e-2/K00/K11 is timed in `reference_threshold`",
                "arithmetic_note": "results are the f64 contract's, bit for bit; the pre-screen's approximate F and the inlier "
                                   "COUNT BOUNDS that decide which hypotheses are solved exactly are computed in f64 / binary32 "
                                   "/ split bf16 (matrix cores) with certified error terms (DESIGN.md 4.3e)"},
            "roofline": roofline_object(table, stats, traffic, traffic_src),
            "hbm_roofline": {
                "achieved": round(bytes_pair * (n_local / (ms_per_step * 1e-3)) / 1e9, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(bytes_pair * (n_local / (ms_per_step * 1e-3)) / 1e9 / HBM_PEAK_GBS, 6),
                "algorithmic_bytes_per_pair": int(bytes_pair),
                "match_kernel_GBs": round(2 * args.kp * 32 * n_local / (kern_ms["match"] * 1e-3) / 1e9, 2)},
            "kernel_ms": {k: round(v, 3) for k, v in kern_ms.items()},
            "kernel_ms_note": "stage times of whole-batch launches on one stream (events between the stages); with two half "
                              "batches on two streams (`ms_per_step`, the default) the stages of the halves overlap, so these do "
                              "not add up to the step",
            "work": {"avg_matches": round(m_avg, 1), "avg_inliers": round(float(res["n_inliers"].mean()), 1),
                     "avg_points": round(float(res["n_points"].mean()), 1), "valid_pairs": int(res["valid"].sum()),
                     "rotations9_per_hyp": round(stats["rotations9"] / max(stats["hypotheses"], 1), 2),
                     "pairs9_per_hyp": round(stats["pairs9"] / max(stats["hypotheses"], 1), 2)},
            "ranks_seen": ranks_seen,
            "ms_per_step_ranks": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
        }
        if gather_us is not None:
            out["gather_us"] = gather_us
            out["gather_note"] = "all-gather of the %d-byte pose records (%d per rank), timed with its own events inside the timed steps; max over ranks" % (rec_bytes, n_local)
        if world > 1 and gathered is not None:
            allrec = mdist.records_to_numpy(gathered, capi.RESULT_DTYPE)
            out["work"]["gathered_records"] = int(len(allrec))
            out["work"]["gathered_valid"] = int(allrec["valid"].sum())
        # ---- the SURVEY 8(d) threshold (5e-2 / K00 / K11, sfm-solve.cpp:311: ~5 inliers, ties decided by the residual sum,
        # nothing to prune) with the same kernels: its own wall-clock loop, its own per-kernel table and roofline
        if args.max_error_sq > 0 and not args.no_ref_threshold and world == 1:
            prm_ref = capi.default_params(sampler=capi.SAMPLER_PHILOX, min_inliers=8, **dict(params_kw, max_error_sq=0.0))
            for _ in range(2):
                batch.run(prm_ref)
            batch.sync()
            t0 = time.perf_counter()
            for _ in range(args.ref_steps):
                batch.run(prm_ref)
            batch.sync()
            dt = time.perf_counter() - t0
            stats_ref = batch.stats(prm_ref)
            table_ref = kernel_table(capi, ctx, batch, prm_ref, stats_ref, n_local)
            res_ref = batch.download(matches=False, mask=False, points=False)["results"]
            out["reference_threshold"] = {
                "max_error_sq": "5e-2/K00/K11 = %.3e" % (5e-2 / 525.0 / 525.0), "steps": args.ref_steps,
                "timing": "wall clock around %d steps + sync (same loop shape as the headline)" % args.ref_steps,
                "pairs_per_s": round(n_local * args.ref_steps / dt, 1), "ms_per_step": round(dt / args.ref_steps * 1e3, 3),
                "valid_pairs": int(res_ref["valid"].sum()), "avg_best_count": round(float(res_ref["best_count"].mean()), 2),
                "roofline": roofline_object(table_ref, stats_ref, None, None)}
            out["reference_threshold_pairs_per_s"] = out["reference_threshold"]["pairs_per_s"]
            batch.run(prm)      # back to the headline results for the legs below
            batch.sync()
            dl = batch.download(matches=False, mask=False, points=False)
        if not args.no_single_pair:
            b1 = capi.Batch(ctx, 1, args.kp, 32)
            b1.upload(0, data["desc1"][:1], data["kp1"][:1], data["n1"][:1], data["desc2"][:1], data["kp2"][:1],
                      data["n2"][:1], data["K"][:1], data["global_index"][:1])
            tot, _ = b1.time(prm, steps=20, warmup=3, per_kernel=False)
            out["single_pair_ms"] = round(tot / 20, 4)  # BASELINE configs[1]: one pair at a time
            b1.close()
            out["image_pair_ctor_ms"] = (probe_result or {}).get("image_pair_ctor_ms")
            out["image_pair_probe"] = probe_result
        if not args.no_pcie and world == 1:
            if args.pcie_naive:
                # (a) naive: synchronous upload from pageable memory + run + synchronous download, nothing overlapped
                t0 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    batch.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"],
                                 data["K"], data["global_index"])
                    batch.run(prm)
                    batch.download()
                out["pcie_inclusive_pairs_per_s_naive"] = round(n_local * reps / (time.perf_counter() - t0), 1)
            # (b) double-buffered: two batches on two contexts, each on its own caller-owned stream, pinned host buffers
            # (mvs_host_alloc), asynchronous upload -> run -> asynchronous download.  Every step moves the full inputs host ->
            # device and the full outputs device -> host.  The two lanes' RUNS are chained by events (lane B's kernels wait for
            # lane A's previous run and vice versa), the copies are not: a lane's upload and download then travel while the
            # OTHER lane computes.  Without the chain both lanes' kernels share the chip, finish together and upload together
            # with the GPU idle (round 4's figure, 63-92 k pairs/s: tools/pcie_probe.py and a rocprofv3 kernel + memory-copy
            # trace showed the lock-step; round 5).  A host C++ caller does the same with mvs_ctx_create_on_stream + hipEvents.
            lanes = []
            N = args.kp
            streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            ctxs = [capi.Context(dev, stream=st.cuda_stream) for st in streams]
            for cx in ctxs:
                if getattr(args, "one_stream", False):
                    cx.set_half_batches(False)
                bb = capi.Batch(cx, n_local, N, 32)
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
            run_done = [None, None]   # event behind the latest run of each lane

            def submit(k):
                bb, pin, (o_res, o_mt, o_mk, o_pt, o_ix) = lanes[k]
                bb.sync()   # the lane's previous step (incl. its download) has completed: its buffers are free
                bb.upload_async(0, pin["desc1"], pin["kp1"], pin["n1"], pin["desc2"], pin["kp2"], pin["n2"], pin["K"],
                                pin["global_index"])
                if run_done[k ^ 1] is not None:
                    streams[k].wait_event(run_done[k ^ 1])   # compute is ONE resource: the runs alternate, the copies float
                bb.run(prm)
                run_done[k] = torch.cuda.Event()
                run_done[k].record(streams[k])
                bb.download_async(0, n_local, o_res, o_mt, o_mk, o_pt, o_ix)

            for k in range(2):
                submit(k % 2)
            for lane in lanes:
                lane[0].sync()
            reps = 8
            t0 = time.perf_counter()
            for k in range(reps):
                submit(k % 2)
            for lane in lanes:
                lane[0].sync()
            dt = time.perf_counter() - t0
            out["pcie_inclusive_pairs_per_s"] = round(n_local * reps / dt, 1)
            out["pcie_note"] = ("two lanes (context + batch on its own stream), pinned host buffers: %.1f MB up + %.1f MB down per "
                                "step; the lanes' runs are chained by events, their copies overlap the other lane's kernels; "
                                "valid pairs in the last downloaded step: %d"
                                % (sum(v.nbytes for v in lanes[0][1].values()) / 1e6, sum(v.nbytes for v in lanes[0][2]) / 1e6,
                                   int(lanes[(reps - 1) % 2][2][0]["valid"].sum())))
            for lane in lanes:
                lane[0].close()
                for a in list(lane[1].values()) + list(lane[2]):
                    capi.pinned_free(a)
            for cx in ctxs:
                cx.close()
        if world == 1 and "refine" in sections:
            out["refine"] = bench_refine(capi, np, batch, data, dl, args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, params_kw, n_local)

    batch.close()
    ctx.close()
    if rank == 0 and world == 1:
        if "sensitivity" in sections:
            out["sensitivity"] = bench_sensitivity(capi, synth, np, dev, args)
        if "sequence" in sections:
            out["sequence"] = bench_sequence(capi, synth, np, dev, args)
        if "extract" in sections:
            out["extract"] = bench_extract(capi, np, dev, args)
    if rank == 0:
        detail_rel = write_detail(out, args.detail)
        if args.print_detail:
            print(json.dumps(out), flush=True)
        print(compact_line(out, detail_rel), flush=True)   # the LAST stdout line: what the driver parses
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

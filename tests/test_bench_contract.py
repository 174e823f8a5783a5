"""The bench line's contract (keys the driver and the judge read), checked on the committed record of the last GPU run:
bench.py itself needs a GPU, its output format must not drift silently."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    d = json.load(open(os.path.join(ROOT, "profiles", "bench_r04_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("valu_fp64", "valu_fp32", "mfma_bf16") and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # executed-work table per kernel, registers / LDS / occupancy from the runtime (mvs_kernel_info_get)
    pk = r["per_kernel"]
    assert r["kernel"] in pk and abs(pk[r["kernel"]]["ms"] - r["launch_ms"]) < 1e-6
    for name in ("ransac_prescreen_kernel", "ransac_count_mfma_kernel<false, 512, 4, 0, 672>", "ransac_finish_mfma_kernel<false, false>",
                 "ransac_finish_mfma_kernel<false, true>",
                 "ransac_finish_upper_kernel<false>", "ransac_exact_list_kernel<1264>", "match_mfma_kernel"):
        assert name in pk and pk[name]["ms"] > 0 and pk[name]["registers_runtime"] > 0, name
    w = r["work"]
    assert w["evals_executed"] == w["evals_executed_f32"] + w["evals_executed_mfma_dense"] + w["evals_executed_mfma_finish"] + \
        w["evals_executed_mfma_finish_rest"] + w["evals_executed_mfma_pilot"]
    assert 0 < w["max_sweeps9"] <= 30                      # assumption A1 of the pre-screen's bound, monitored by every bench run
    assert w["exact_solves"] + w["prescreened_only"] == w["hypotheses"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread_pairs_per_s"):
        assert k in c, k
    assert c["kind"] == "port"            # the reference itself cannot be built here
    assert abs(d["value"] - 512 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    rt = d["reference_threshold"]
    assert rt["steps"] >= 10 and "wall clock" in rt["timing"] and "roofline" in rt and rt["pairs_per_s"] > 0
    assert d["reference_threshold_pairs_per_s"] == rt["pairs_per_s"]          # next to `value`
    assert d["pcie_inclusive_pairs_per_s"] > 0                                # on by default
    assert d["single_pair_ms"] > 0 and d["image_pair_ctor_ms"] > 0
    sens = d["sensitivity"]
    assert len(sens["cells"]) == 24 and sens["min_pairs_per_s"] > 0
    for c in sens["cells"]:
        assert set(("outlier_frac", "noise_px", "max_error_sq", "pairs_per_s", "exact_solve_share", "pairs_mode")) <= set(c)
    for k in ("sequence", "refine", "extract"):
        assert k in d and d[k]["value"] > 0 and ("roofline" in d[k] or "pnp_roofline" in d[k]), k


def test_bench_source_prints_the_same_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ('"metric"', '"roofline"', '"cpu_baseline"', '"traffic"', '"bound"', '"gather_us"', '"reference_threshold"',
              '"sensitivity"', '"pcie_inclusive_pairs_per_s"', '"image_pair_ctor_ms"', '"reference_threshold_pairs_per_s"',
              '"sequence_frames_per_s"', '"extract_images_per_s"', '"sensitivity_min_pairs_per_s"', '"single_pair_ms"',
              '"ranks_seen"', '"detail"'):
        assert k in src, k
    # the compact line is the LAST thing rank 0 prints (the driver reads a bounded tail of stdout)
    tail = src[src.rindex("print(compact_line("):]
    assert "print(" not in tail[len("print(compact_line("):], "something is printed after the compact line"


COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "reference_threshold_pairs_per_s",
                "single_pair_ms", "image_pair_ctor_ms", "pcie_inclusive_pairs_per_s", "sequence_frames_per_s",
                "extract_images_per_s", "sensitivity_min_pairs_per_s", "ranks_seen", "gather_us")


def _check_compact(line):
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line)
    for k in COMPACT_KEYS:
        assert k in d, k
    for k in ("workload", "pairs_per_gpu", "keypoints", "hypotheses", "max_error_sq", "parallelism"):
        assert k in d["config"], k
    assert "model" not in d["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_ms", "flops_per_launch", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    # nothing nested beyond the three contract objects: the line cannot grow with the number of kernels or cells
    for k, v in d.items():
        if k not in ("config", "roofline", "cpu_baseline"):
            assert not isinstance(v, dict), k
    for sub in ("config", "roofline", "cpu_baseline"):
        assert all(not isinstance(v, (dict, list)) for v in d[sub].values()), sub
    return d


def test_compact_line_from_a_full_detail_object_is_small_and_parses_like_the_driver_reads_it():
    """bench.py's last stdout line, rebuilt on the CPU from a committed full result object: < 4 KB, and what a reader that
    only sees the last 8 KB of stdout finds as the last line parses on its own (BENCH_r04 had `parsed null` because the one
    line was 28.6 KB)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_r04_default.json")))
    assert len(json.dumps(full)) > 20000            # the object that no longer fitted
    line = bench.compact_line(full, "bench_detail.json")
    d = _check_compact(line)
    assert d["value"] == full["value"] and d["roofline"]["frac"] == full["roofline"]["frac"]
    assert d["sequence_frames_per_s"] == full["sequence"]["value"] and d["sensitivity_min_pairs_per_s"] == full["sensitivity"]["min_pairs_per_s"]
    # the driver's view: stderr noise, optionally the full object as an EARLIER line, then the compact line; 8 KB tail
    stdout = "some warning\n" + json.dumps(full) + "\n" + line + "\n"
    tail = stdout[-8192:]
    last = [ln for ln in tail.splitlines() if ln.strip()][-1]
    assert json.loads(last) == d
    # a multi-GPU object (gather fields present) and a minimal one (side legs skipped) also fit
    multi = dict(full, n_gpus=8, gather_us=41.5, ranks_seen=list(range(8)))
    multi["work"] = dict(full["work"], gathered_records=4096)
    dm = _check_compact(bench.compact_line(multi, None))
    assert dm["gathered_records"] == 4096 and dm["gather_us"] == 41.5 and dm["ranks_seen"] == list(range(8))
    minimal = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data", "config", "roofline")}
    dmin = json.loads(bench.compact_line(minimal, None))
    assert dmin["cpu_baseline"] is None and dmin["sequence_frames_per_s"] is None


def test_committed_compact_line_of_this_round():
    """The line the GPU box printed this round (profiles/bench_r05_default.json = the last stdout line, verbatim)."""
    path = os.path.join(ROOT, "profiles", "bench_r05_default.json")
    if not os.path.exists(path):
        import pytest
        pytest.skip("no round-5 line committed yet")
    raw = open(path).read().strip()
    d = _check_compact(raw)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == [0] and d["gather_us"] is None
    assert abs(d["value"] - d["config"]["pairs_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    assert d["cpu_baseline"]["kind"] == "port" and d["valid_pairs"] == d["config"]["pairs_per_gpu"]
    assert 0 < d["max_sweeps9"] <= 30

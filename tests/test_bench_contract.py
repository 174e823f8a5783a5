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
              '"sensitivity"', '"pcie_inclusive_pairs_per_s"', '"image_pair_ctor_ms"', '"reference_threshold_pairs_per_s"'):
        assert k in src, k

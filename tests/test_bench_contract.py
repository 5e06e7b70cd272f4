"""The bench line's contract, checked on the recorded output of the last GPU run (profiles/) and on
bench.py's pure helpers -- no GPU needed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _path():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1_with_cpu_baseline.json")))[-1]


def _line():
    return json.load(open(_path()))


def test_recorded_line_has_the_contract_fields():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] in base["metric"]
    for key in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # ("valu": compute-bound on the fp64 vector pipe, which shares the matrix pipe's peak -- verdict r3 weak 8c)
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value = samples / step time
    assert abs(d["value"] - d["config"]["N_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_flop_and_traffic_helpers():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # K (D^2 + 4D) + 40 K  and  K (1 + 2D + D(D+1))
    assert bench.flops_logpdf(32, 20) == 32 * 480 + 32 * 40
    assert bench.flops_stats(32, 20) == 32 * 461
    t, src = bench.measured_traffic("k_stats", 10_000_000)
    assert t is None or (1e9 < t < 2e10 and src.startswith("profiles/"))
    assert bench.measured_traffic("no_such_kernel", 1) == (None, None)
    ratio, where = bench.reference_ratio()
    assert ratio is None or (1.0 < ratio < 5.0 and "measured" in where and " on: " in where)
    assert set(bench.PIPE_OF.values()) <= set(bench.ATTAINABLE_TFLOPS) and all(
        v < bench.FP64_PEAK_TFLOPS for v in bench.ATTAINABLE_TFLOPS.values())


def test_recorded_line_names_its_sources():
    r = _line()["roofline"]
    assert "timing_source" in r and "pmc_get_timings" in r["timing_source"]
    if "traffic_measured_live" in r:                     # (lines from round 4 on: an explicit flag, not a substring)
        assert isinstance(r["traffic_measured_live"], bool)
        assert r["traffic_measured_live"] == (r["traffic"] is not None and r["traffic_source"].startswith("measured in this run"))
        assert r["bound"] == r["per_kernel_bound"][r["kernel"]] and r["attainable_peak"] < r["peak"]
        assert abs(r["frac_of_attainable"] - r["achieved"] / r["attainable_peak"]) < 1e-12
    else:
        assert _path().split(os.sep)[-1] < "r04"

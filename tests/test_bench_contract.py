"""The bench line's contract, checked on the recorded output of the last GPU run (profiles/) and on
bench.py's pure helpers -- no GPU needed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    import glob
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1_with_cpu_baseline.json")))[-1]
    return json.load(open(path))


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
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
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
    ratio = bench.reference_ratio()
    assert ratio is None or 1.0 < ratio < 5.0


def test_recorded_line_names_its_sources():
    r = _line()["roofline"]
    assert "timing_source" in r and "pmc_get_timings" in r["timing_source"]
    assert r["traffic"] is None or "measured in this run" in r["traffic_source"]     # live, or says that it is not

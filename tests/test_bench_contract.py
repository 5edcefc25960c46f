"""bench.py's bookkeeping that can be checked without a GPU: the algorithmic work per scene is SURVEY.md 8(d)'s (493.4 GFLOP / 6.61 GB at
the metric configuration, 131.3 GFLOP / 1.86 GB at configs[1]), the workloads are the configurations BASELINE.json names, the committed
counter files carry the stamp of the library they were measured on, and the committed default line has the contract's keys."""
import importlib.util
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_work_is_surveys_formula():
    b = _bench()
    N, C, H, W, T = b.WORKLOADS["metric"]
    assert (N, C, H, W, T) == (4, 64, 200, 704, 20)                      # BASELINE.json metric
    fl, by = b.algorithmic_work(N, C, H * W, T)
    assert fl / 1e9 == pytest.approx(493.4, abs=0.1) and by / 1e9 == pytest.approx(6.61, abs=0.01)
    N, C, H, W, T = b.WORKLOADS["cfg2"]
    assert (N, C, H, W, T) == (2, 64, 200, 704, 10)                      # BASELINE.json configs[1]
    fl, by = b.algorithmic_work(N, C, H * W, T)
    assert fl / 1e9 == pytest.approx(131.3, abs=0.1) and by / 1e9 == pytest.approx(1.86, abs=0.01)
    assert b.HBM_PEAK_GBS == 8000.0 and b.FP32_PEAK_TFLOPS == pytest.approx(157.3)


def test_committed_counter_files_are_stamped_and_consistent():
    for name in ("r4_pmc_traffic.json", "r4_pmc_sq.json", "r4_pmc_insts.json"):
        d = json.load(open(os.path.join(REPO, "profiles", name)))
        assert d.get("workload") == "metric" and len(d.get("library_src", "")) == 16, name
    t = json.load(open(os.path.join(REPO, "profiles", "r4_pmc_traffic.json")))
    fam = t["conv8h_family"]
    # counter traffic per launch within a few percent of the algorithmic bytes of the family's launch mix (122.55 MB)
    assert 0.95 < fam["hbm_bytes_per_launch"] / 122.55e6 < 1.10
    assert t["scenes_per_launch"] == 4


def test_committed_default_line_has_the_contract_keys():
    d = json.load(open(os.path.join(REPO, "profiles", "r4_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "scenes/sec" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "scenes/sec" and "sample" in c
    assert d["value"] == pytest.approx(d["total_scenes"] / (d["ms_per_step"] * 1e-3 * d["steps"]), rel=1e-6)

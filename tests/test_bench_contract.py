"""CPU-only: the shape of bench.py's ONE JSON line (the driver's contract), checked on the line recorded on MI355X by
profiles/collect.sh (profiles/r03_bench.json): metric / config as BASELINE.json names them, whole-job value, the roofline and
cpu_baseline objects with every field the contract lists, internally consistent numbers."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_recorded_bench_line_has_the_contract_shape():
    line = json.loads(sorted((ROOT / "profiles").glob("r[0-9][0-9]_bench.json"))[-1].read_text())
    base = json.loads((ROOT / "BASELINE.json").read_text())
    assert line["metric"] == base["metric"] and line["unit"] == "Msamples/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "f64"
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert line["config"]["workload"].startswith("configs[3]: 64 concurrent IQ streams")
    # value = samples of all streams x steps / time
    n = line["config"]["samples_per_stream"] * line["config"]["streams_per_gpu"] * line["n_gpus"]
    assert abs(line["value"] - n / (line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"].startswith("k_msk_frontend")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    algo = n / line["n_gpus"] * (4.0 + 134.0 / 86720)                   # 4 B in + 134 B per frame out
    assert abs(r["achieved"] - algo / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.99 * r["achieved"]    # measured HBM bytes are never below the algorithmic ones
    assert r["issue"]["wave_cycles_per_symbol"] > 600 and 2.0 < r["issue"]["clock_GHz"] < 2.5
    assert r["fp64_valu"]["peak"] == 78.6 and r["fp64_valu"]["unit"] == "TFLOP/s"
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and 1.0 < c["value"] < 200.0
    assert line["check"]["edge_ties"] == 0 and line["check"]["frames_exact"] >= 0.99 * line["check"]["frames_total"]

"""The committed bench line of the latest round (profiles/rNN_bench_default.json, the output of `python bench.py` on an MI355X)
carries every field the measurement contract names, with consistent values, and the committed PMC traffic file is the
trailing update's own population (one dispatch per launch of a step).  Host-only."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    assert files, pattern
    return files[-1]


def test_committed_bench_line_has_the_contract_fields():
    line = json.load(open(_latest("r*_bench_default.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["dtype"] == "f64" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert line["higher_is_better"] is True and line["unit"] == "factors/s"
    assert "workload" in line["config"] and "N=65536" in line["config"]["workload"]          # the north-star configuration
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) <= 1e-9 * line["value"]            # whole-job factors per second
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 78.6
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert roof["traffic"] is None or roof["traffic"] > roof["algorithmic_bytes_per_launch"] * 0.5
    # the Gram's own HBM figure rides in the same line
    assert roof["gram"]["bound"] == "hbm" and abs(roof["gram"]["frac"] - roof["gram"]["achieved"] / roof["gram"]["peak"]) < 1e-12
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1


def test_committed_traffic_file_matches_the_bench_line():
    bench = _latest("r*_bench_default.json")
    tag = re.search(r"(r\d+)_", os.path.basename(bench)).group(1)
    assert int(tag[1:]) >= 4, "round 4 re-made the traffic pass with the phases off"
    line = json.load(open(bench))
    t = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_bench_traffic.json")))
    assert abs(line["roofline"]["traffic"] - t["hbm_bytes_per_launch"]) <= 1e-6 * t["hbm_bytes_per_launch"]
    # GPC_BENCH_PHASES=0 under the counters: the dispatches of the kernel name ARE one step's trailing updates
    assert t["FETCH_SIZE"]["dispatches"] == t["WRITE_SIZE"]["dispatches"] == int(line["roofline"]["launches_per_step"])
    assert "GPC_BENCH_PHASES=0" in t["command"]

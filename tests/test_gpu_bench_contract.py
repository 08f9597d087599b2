"""bench.py's one JSON line at N = 1 (the driver's contract): metric, value, config.workload, roofline and cpu_baseline
objects, on a reduced number of reads so that the test takes seconds (a reduced run is not a result: --reads marks it in
the workload text and leaves roofline.traffic, which was measured on the full configuration, out)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "200000"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "k-mers/s" and d["vs_baseline"] is None
    assert d["dtype"] == "u64" and d["data"] == "synthetic" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert "configs[1]" in d["config"]["workload"] and "200000 x 100 bp" in d["config"]["workload"]
    assert "model" not in d["config"]
    # 200 k reads x 77 windows, minus those over an N
    assert 0.97 * 200000 * 77 < d["config"]["insertions_per_step"] <= 200000 * 77
    assert abs(d["value"] - d["config"]["insertions_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["traffic"] is None
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - d["value"] * rf["alg_bytes_per_kmer"] / 1e9) < 1e-6 * rf["achieved"]
    assert rf["kernel_avg_ms"] > 0 and rf["kernel_launches"] >= 2
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "k-mers/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb

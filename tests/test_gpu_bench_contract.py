"""bench.py's one JSON line at N = 1 (the driver's contract): metric, value, config.workload, roofline and cpu_baseline
objects, the run's own content check, and the configs[1] figures beside the configs[2] line. The first test runs a reduced
number of reads so that it takes seconds (a reduced run is not a result: --reads marks it in the workload text and
leaves roofline.traffic, which was measured on the full configuration, and the configs[1] block out); the second is the
driver's own command on the full workload with few steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def check_common(d, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "content_check"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["unit"] == "k-mers/s" and d["vs_baseline"] is None
    assert d["dtype"] == "u64" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "model" not in d["config"]
    assert abs(d["value"] - d["config"]["insertions_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - d["value"] * rf["alg_bytes_per_kmer"] / 1e9) < 1e-6 * rf["achieved"]
    assert rf["kernel_avg_ms"] > 0 and rf["kernel_launches"] >= 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "k-mers/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    cc = d["content_check"]
    assert cc["ok"] is True and cc["entries"] > 0 and cc["insertions"] >= cc["entries"] and len(cc["xor"]) == 16


def test_bench_line_has_the_contract_fields():
    d = run_bench("--steps", "2", "--warmup", "1", "--reads", "200000")
    check_common(d, 2, 1)
    # every N runs configs[2], strong scaling: one driver invocation yields one curve
    assert d["scaling"] == "strong" and "configs[2]" in d["config"]["workload"] and "200000 per GPU" in d["config"]["workload"]
    # 200 k reads x 77 windows, minus those over an N
    assert 0.97 * 200000 * 77 < d["config"]["insertions_per_step"] <= 200000 * 77
    assert d["roofline"]["traffic"] is None and "configs1" not in d
    # the former headline workload stays selectable
    d1 = run_bench("--steps", "2", "--warmup", "1", "--reads", "200000", "--workload", "c1", "--no-cpu-baseline")
    assert d1["scaling"] == "weak" and "configs[1]" in d1["config"]["workload"] and d1["content_check"]["ok"] is True


def test_the_drivers_command_on_the_full_workload():
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1")
    check_common(d, 3, 1)
    cfg = d["config"]
    assert "configs[2]" in cfg["workload"] and cfg["slots_per_gpu"] == 1 << 31 and cfg["pieces_per_gpu"] == 4
    assert 0.97 * 50_000_000 * 77 < cfg["insertions_per_step"] <= 50_000_000 * 77
    assert cfg["window_passes_per_step"] == 1             # the table is streamed once per step, not once per piece
    win = d["roofline"]["kernels"]["k_window_count<true, false, true>"]   # packed items at 2^31 slots
    assert win["launches_per_step"] == 1.0
    assert d["value"] > 40e9
    c1 = d["configs1"]
    assert "configs[1]" in c1["workload"] and c1["value"] > 40e9 and 0 < c1["roofline"]["frac"] < 1
    assert c1["insertions_per_step"] == 152158273 and "random_access_roofline" in c1

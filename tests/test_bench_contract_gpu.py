"""bench.py's output contract: exactly ONE line on stdout, a JSON object with the driver's keys
plus the `roofline` and `cpu_baseline` objects (the latter null when skipped), on the plain
single-process path and on the data-parallel path (RCCL, one rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*flags):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "128", "--warmup", "64",
                          "--no-cpu-baseline", *flags], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    return json.loads(lines[0])


def check(line):
    for k in KEYS:
        assert k in line, k
    assert line["metric"].startswith("training samples/sec") and line["unit"] == "samples/s"
    assert line["steps"] == 128 and line["warmup"] == 64 and line["n_gpus"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"]
    assert abs(line["value"] - 128 * 128 / (line["ms_per_step"] * 1e-3 * 128)) <= 0.01 * line["value"]
    r = line["roofline"]
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_bench_line_single_process():
    line = run_bench()
    check(line)
    assert line["cpu_baseline"] is None          # --no-cpu-baseline
    assert isinstance(line["roofline"]["traffic"], int)


def test_bench_line_data_parallel_path():
    line = run_bench("--dp-path")
    check(line)
    assert line["config"]["dp_replicas_identical"] is True
    assert line["config"]["dp_mode"] in ("hipGraph replay of 64 steps", "eager")

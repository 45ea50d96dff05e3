"""bench.py end to end on the small config-1 workload: one JSON line with the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and r["value"] > 1e9
    assert "workload" in r["config"] and r["config"]["source_points"] == 3233
    rl = r["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rl, key
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-9
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["gpu_vs_cpu_rel_to_max"] < 2e-5

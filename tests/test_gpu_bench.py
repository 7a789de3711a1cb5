"""bench.py itself on the GPU: the one JSON line, and the host budget of an 8-rank node proven on one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_and_rank_confined_to_two_cpus():
    """A short bench run (smaller launches, 20 000 references: seconds, not the headline configuration): the line
    carries what the contract asks for, and the confined leg -- every thread of the process on TWO CPUs, what a
    rank of an 8-rank node gets of a 16-CPU quota (src/sina.cpp:241-243,450: the reference sizes its pipeline by
    --threads) -- keeps most of the unconfined rate.  (One MI355X, full-size steps: 0.92-0.95 at two CPUs, 0.975 at
    three, profiles/r04_host_threads.txt; the floor asserted here leaves room for a busy test box.)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
           "--refs", "20000", "--batch", "3072", "--sub-batch", "3072"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["unit"] == "sequences/s" and j["higher_is_better"] and j["scaling"] == "weak"
    assert j["steps"] == 8 and j["warmup"] == 2 and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert j["value"] > 20000 and abs(j["value"] * j["ms_per_step"] * 1e-3 / 3072 - 1.0) < 0.02
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.2 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["frac"] >= r["frac_by_start_to_end"] - 1e-9
    c = j["confined"]
    assert c["cpus"] == 2 and j["confined_rate_frac"] == c["rate_frac"]
    assert c["host_cores_busy"] <= 2.05
    assert c["rate_frac"] >= 0.8, c

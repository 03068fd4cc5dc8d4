"""bench.py prints ONE JSON line that keeps the driver's contract (metric, value, unit, n_gpus, steps, warmup, ms_per_step,
higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus the `roofline` object; run at a small size so the
check costs seconds (the CPU baseline and the extra loops are skipped: they have their own switches)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--accuracy"], ["--config", "C3B"], ["--clips", "2"]])
def test_bench_line_contract(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--lr-h", "64", "--lr-w", "96",
           "--no-cpu-baseline", "--no-extras"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f16"
    assert "workload" in d["config"] and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.02     # one GPU: frames/s x seconds per forward call = 1
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "whole_frame"):
        assert k in r, k
    assert r["bound"] in ("mfma", "hbm") and 0 < r["frac"] < 1
    # the multi-GPU decomposition fields exist at one rank too (per-rank frames/s, gather waits, the root's resident bytes)
    assert len(d["ranks"]["frames_per_s"]) == 1 and d["ranks"]["frames_per_s"][0] >= d["value"] * 0.99
    assert d["gather"]["rounds"] >= 1 and d["gather"]["root_resident_bytes"] > 0
    if "--accuracy" in extra:   # PSNR / max relative error of this build against the oracle, in the line itself
        # (the maximum sits on the few pixels where a DISCRETE guidance plane flipped by a whole step: tests/test_gpu_vsr.py)
        assert d["psnr_vs_oracle_db"] > 58.0 and 0 <= d["max_rel_err"] < 0.5 and d["accuracy"]["p99_rel_err"] < 1e-2, d["accuracy"]
        assert 0 <= d["accuracy"]["sr_stack_max_rel_err_identical_planes"] < 1e-3, d["accuracy"]   # north_star: 1e-3 relative (fp16 storage: ~1e-4)
        assert "oracle" in d["accuracy"]["tile"]

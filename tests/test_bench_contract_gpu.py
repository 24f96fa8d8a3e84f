"""bench.py's one-line JSON contract (driver-facing), exercised on tiny shapes so it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--steps", "6", "--warmup", "2",
                        "--ctx-text", "5", "--ctx-frames", "8", "--gen-text", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) / d["value"] < 0.02          # B = 1: frames/s == 1 / step time


def test_bench_self_launches_its_ranks_when_no_launcher_is_present():
    """`python bench.py --gpus 2` with no torchrun around it: the GPU-free parent spawns both ranks and relays rank 0's
    line (it used to exit 2).  BENCH_SHARE_GPU0=1 lets the two ranks share the box's one GPU over gloo --
    a control-flow check (weight broadcast, barriers, max-over-ranks timing), never a measurement."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_SHARE_GPU0"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tiny", "--steps", "6", "--warmup", "2",
                        "--ctx-text", "5", "--ctx-frames", "8", "--gen-text", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "replicas x2" and "cpu_baseline" in d
    assert abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.02      # two replicas' frames / max-over-ranks time

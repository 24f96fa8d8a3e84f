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
    wb = d["weight_broadcast"]
    assert wb["csm_bytes"] > 0 and wb["csm_ms"] > 0 and wb["csm_GBps"] > 0
    _check_rank_records(d, 2)


def test_eight_ranks_start_rendezvous_and_report_rehearsal():
    """The driver's 8-GPU run, rehearsed on this box's one GPU (never a measurement): `python bench.py --gpus 8 --tiny`
    with BENCH_SHARE_GPU0=1 -- the GPU-free parent hosts the rendezvous store and starts eight rank processes, which build their process
    group (gloo), receive the weight blob, step, all-gather their records and wait for rank 0's line.  Eight rank records, each with its
    share of the host's threads (quota // 8, at least 1), the JSON line LAST on stdout, the whole thing inside two minutes."""
    import time
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_SHARE_GPU0"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--tiny", "--steps", "4", "--warmup", "1",
                        "--ctx-text", "5", "--ctx-frames", "8", "--gen-text", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["config"]["parallelism"] == "replicas x8"
    _check_rank_records(d, 8)
    quota = bench.cpu_quota() or os.cpu_count() or 1
    want_threads = max(1, min(quota, os.cpu_count() or quota) // 8)
    assert all(r["host_threads"] <= max(1, want_threads) for r in d["ranks"]), ([r["host_threads"] for r in d["ranks"]], want_threads)
    print(f"\n[8 ranks] start -> line in {wall:.1f} s; host threads per rank {sorted({r['host_threads'] for r in d['ranks']})} (quota {quota}); "
          f"slowest rank {d['ms_per_step']:.2f} ms/step")
    assert wall < 120, f"eight tiny ranks took {wall:.0f} s"


def _check_rank_records(d, n):
    """The N > 1 line proves itself: the rank count is the process group's, every rank reports the device it ran
    on (index + PCI id), its own step time / frames / broadcast rate and its share of the host's threads."""
    assert d["ranks_seen"] == n == d["n_gpus"] and len(d["ranks"]) == n and sorted(r["rank"] for r in d["ranks"]) == list(range(n))
    for r in d["ranks"]:
        assert r["ms_per_step"] > 0 and r["frames"] == d["steps"] * d["config"]["batch_per_gpu"] and r["pci_bus_id"]
        assert r["ms_per_step"] <= d["ms_per_step"] * 1.001, "value is priced on the SLOWEST rank"
        assert r["broadcast_GBps"] > 0 and r["host_threads"] >= 1
    assert d["distinct_gpus"] == 1, "BENCH_SHARE_GPU0: both ranks sit on GPU 0 (on a real node this is N)"
    assert d["collective_backend"] == "gloo"


def test_two_full_size_ranks_carry_the_config4_leg_and_honest_kernel_entries():
    """The N > 1 line at CSM-1B size (two ranks sharing this box's one GPU over gloo -- control flow only, never a
    measurement): `extras.config4` = B = 32 per rank with the aggregate over the slowest rank's time, the weight broadcast's
    size / time / rate for both blobs (CSM and the codec), and `roofline.dominant_kernels` naming launches that are on the
    timed path (none in this debug mode: the all-CU launches are off when ranks share a GPU)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_SHARE_GPU0"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--extra-steps", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    c4 = d["extras"]["config4"]
    assert "batch 64 sharded over 2 GPUs" in c4["workload"]
    assert abs(c4["aggregate_frames_per_s"] - 2 * 32 * 1000.0 / c4["slowest_rank_ms_per_step"]) / c4["aggregate_frames_per_s"] < 0.02
    wb = d["weight_broadcast"]
    assert wb["csm_bytes"] > 3e9 and wb["mimi_bytes"] > 1e8 and wb["csm_ms"] > 0 and wb["mimi_ms"] > 0
    _check_rank_records(d, 2)
    assert isinstance(d["roofline"]["dominant_kernels"], list)


def test_the_multi_rank_control_flow_on_a_real_rccl_process_group_of_one_rank():
    """BENCH_RCCL_WORLD1=1: the N > 1 control flow of bench.py (flat-blob weight broadcast, barriers, all-gathered rank records, max over
    ranks) on an RCCL process group -- one rank, because the box has one GPU; the gloo twin mode above never calls RCCL.  The JSON line is
    the LAST line of stdout: RCCL's version banner, which C stdio would flush at process exit, is flushed before it."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_RCCL_WORLD1"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--steps", "6", "--warmup", "2",
                        "--ctx-text", "5", "--ctx-frames", "8", "--gen-text", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                      # (anything RCCL printed comes before it)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["collective_backend"] == "nccl" and d["rccl_version"]
    assert d["distinct_gpus"] == 1 and d["ranks"][0]["pci_bus_id"] and d["ranks"][0]["ms_per_step"] > 0
    wb = d["weight_broadcast"]
    assert wb["csm_bytes"] > 0 and wb["csm_ms"] > 0 and "nccl" in wb["collective"]


def test_single_gpu_line_names_the_kernels_it_timed():
    """N = 1, CSM-1B: `roofline.dominant_kernels` must be the launches of the timed frame step -- the persistent depth decoder and
    the one-launch backbone layer -- timed live in the run, and `extras` must carry config 3 (with ITS kernel), the
    reference-style host loop, the long-context step, config 5 and config 5 at B = 32."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--extra-steps", "6", "--no-cpu-baseline", "--no-mimi"],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    names = [k["kernel"] for k in d["roofline"]["dominant_kernels"]]
    assert names == ["k_dec_persist", "k_dec_first", "k_bb_layer"], names
    dk = d["roofline"]["dominant_kernels"]
    assert 16 * dk[2]["avg_us"] * 1e-3 + dk[1]["avg_us"] * 1e-3 + dk[0]["avg_us"] * 1e-3 < d["ms_per_step"], "the dominant kernels take longer than the frame they are part of"
    assert "k_dec_first (codebook 1) + k_dec_persist" in d["config"]["paths"]
    ex = d["extras"]
    for k in ("config3", "reference_loop", "b1_long_context", "config5", "config5_b32"):
        assert k in ex, k
    assert ex["config3"]["dominant_kernels"][0]["kernel"] == "k_dec_persist_m<2>"
    assert ex["reference_loop"]["ms_per_frame_after_the_prompt"] > d["ms_per_step"] * 0.9

#!/usr/bin/env python3
"""Headline benchmark: audio frames/s (and real-time factor) of the CSM-1B audio-token loop.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: this process spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one generated 80 ms audio frame for every stream of the batch: one backbone step,
the c0 head, 31 depth-decoder steps and 32 samplings (reference: Model.generate_frame,
sesameai/models.py:132-184, driven as in sesameai/generator.py:283-294).  The N=1 workload is
BASELINE.json configs[1]: CSM-1B, one utterance (B=1), one voice-prompt segment
(S = 40 text + 125 audio + 1 EOS + 24 text = 190 prompt rows), bf16, seeded
random weights (no checkpoint can be downloaded), greedy-free sampling T=0.9 / top-k 50.
Multi-GPU = independent replicas (one process per GPU, weights broadcast once over RCCL, no
per-step collective): "weak" scaling, value = all ranks' frames / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.  Inputs are resident in HBM before the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "sesameai-tts_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synthetic_prompt(args, batch, text_vocab, seed0=2025, segments=1, ctx_text=None, ctx_frames=None):
    """`segments` context segments (ctx_text text rows + ctx_frames audio rows + the all-zero EOS row) + gen_text text
    rows.  Defaults = config 2 (1 x (40 + 125 + 1) + 24 = 190 rows); config 5 = 10 x (30 + 100 + 1) + 24 = 1334."""
    ctx_text = args.ctx_text if ctx_text is None else ctx_text
    ctx_frames = args.ctx_frames if ctx_frames is None else ctx_frames
    toks, masks = [], []
    for b in range(batch):
        g = torch.Generator().manual_seed(seed0 + b)
        rows = segments * (ctx_text + ctx_frames + 1) + args.gen_text
        t = torch.zeros(rows, 33, dtype=torch.long)
        m = torch.zeros(rows, 33, dtype=torch.bool)
        r = 0
        for _ in range(segments):
            t[r:r + ctx_text, 32] = torch.randint(0, text_vocab, (ctx_text,), generator=g); m[r:r + ctx_text, 32] = True
            r += ctx_text
            t[r:r + ctx_frames, :32] = torch.randint(0, 2048, (ctx_frames, 32), generator=g); m[r:r + ctx_frames + 1, :32] = True
            r += ctx_frames + 1                       # the all-zero EOS frame
        t[r:r + args.gen_text, 32] = torch.randint(0, text_vocab, (args.gen_text,), generator=g); m[r:r + args.gen_text, 32] = True
        toks.append(t); masks.append(m)
    return torch.stack(toks), torch.stack(masks)


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.cuh, sorted by name): ties a PMC pass to the kernels it measured.  A
    profiles/r*/pmc_traffic.json taken on other sources is NOT reported as this run's traffic."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.cuh"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max / v1 cfs quota), or None.  More runnable threads than this
    get the whole process throttled for the rest of each period -- on the GPU boxes of this pool (quota 16, 256 logical CPUs) a torch
    CPU op on 128 OpenMP threads turned a 6 ms call into 85 ms (tools/dbg/prefill_wall_diag.py)."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, q // p)
    except (OSError, ValueError):
        return None


def default_cpu_threads():
    n = min(32, os.cpu_count() or 1)
    q = cpu_quota()
    return min(n, q) if q else n


def host_description(threads_used):
    """cpu_baseline.host: what the CPU port ran on (north_star: "core count stated")."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "usable_cpus": usable, "cgroup_cpu_quota": cpu_quota(), "threads_used": threads_used}


def device_pci_bus_id(index):
    """"domain:bus:device" of a visible GPU from its device properties (no GPU work) + its uuid: what tells N ranks on N GPUs from N ranks on one."""
    p = torch.cuda.get_device_properties(index)
    try:
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x} uuid {p.uuid}"
    except AttributeError:
        return None


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def dominant_kernels(model, B, temperature, topk, reps=20):
    """The launches that carry the timed frame step, timed live (HIP events on the launching stream, back-to-back launches on
    the handle's current state; include/csm_hip_ops.h csm_debug_time_kernels): the persistent depth-decoder launch (codebooks
    2..31 of a frame: csrc/dec_persist.cuh at B = 1, csrc/dec_persist_m.cuh at B = 2..32) and, at B = 1, the first decoder step as one
    launch (codebook 1: csrc/dec_first.cuh) and the one-launch backbone layer (csrc/bb_block.cuh).  bytes = weight bytes the launch streams; the decoder's come from the Infinity Cache after the first
    step (30 x 226 MB per launch, 222 MB of decoder layers resident), so its rate is not an HBM rate."""
    import ctypes as C
    from sesameai import _abi
    out = (C.c_double * 6)()
    st = torch.cuda.current_stream().cuda_stream
    with torch.cuda.device(model.device):
        _abi.check(_abi.lib.csm_debug_time_kernels(model._h, B, reps, float(temperature), int(topk), out, st), model._h)
    res = []
    if out[0] == out[0]:
        res.append({"kernel": "k_dec_persist" if B == 1 else f"k_dec_persist_m<{1 if B <= 16 else 2}>",
                    "does": f"codebooks 2..31 of a frame for {B} utterance(s): 30 steps x (4 decoder layers + head + sampler) in one launch",
                    "launches_per_frame": 1, "bytes_streamed_per_launch": out[1], "avg_us": round(out[0], 1),
                    "streamed_GBps": round(out[1] / out[0] / 1e3, 1), "us_per_decoder_step": round(out[0] / 30.0, 2),
                    "bound": "cross-workgroup hand-off latency (see DESIGN.md); bytes come from the 256 MB Infinity Cache"})
    if out[4] == out[4]:
        res.append({"kernel": "k_dec_first", "does": "codebook 1 of a frame: the decoder's first step, positions 0 and 1 (4 layers on two rows + the head) in one launch",
                    "launches_per_frame": 1, "bytes_streamed_per_launch": out[5], "avg_us": round(out[4], 1), "streamed_GBps": round(out[5] / out[4] / 1e3, 1),
                    "bound": "cross-workgroup hand-off latency, two rows per hand-off (csrc/dec_first.cuh)"})
    if out[2] == out[2]:
        res.append({"kernel": "k_bb_layer", "does": "one backbone layer of a B = 1 decode step (attention block + MLP) in one launch",
                    "launches_per_frame": 16, "bytes_per_launch": out[3], "avg_us": round(out[2], 2),
                    "achieved_GBps": round(out[3] / out[2] / 1e3, 1), "frac_of_hbm_peak": round(out[3] / out[2] / 1e3 / HBM_PEAK_GBS, 3)})
    return res


def cpu_worker(args):
    """Child process: the oracle (CPU restatement of the reference's eager `-d cpu` bf16 graph) on the headline's S=190 prompt
    (BASELINE config 2) until the budget is spent, then on BASELINE config 1's shape (16 text tokens, no context, up to 25 frames =
    2 s of audio; SURVEY.md 8d asks for both) on a smaller budget -- one process, so the 1.5 B weights are drawn once.  Prints one
    JSON line per completed frame."""
    from oracle import csm_ref as C
    shape = C.csm_1b()
    w = C.make_weights(shape, seed=1234)
    m = C.OracleModel(shape, w)
    m.setup_caches(1)
    ids = torch.randint(0, shape.text_vocab_size, (16,), generator=torch.Generator().manual_seed(2025)).tolist()
    t1, m1 = C.build_prompt([(ids, None)])
    runs = [("config2", synthetic_prompt(args, 1, shape.text_vocab_size), args.cpu_budget, args.cpu_frames),
            ("config1", (t1.unsqueeze(0), m1.unsqueeze(0)), min(args.cpu_budget, 8.0), 25)]
    for name, (tokens, mask), budget, max_frames in runs:
        torch.manual_seed(0)
        m.reset_caches()
        cur_t, cur_m = tokens, mask
        pos = torch.arange(tokens.shape[1]).unsqueeze(0)
        t0 = time.time()
        n = 0
        while time.time() - t0 < budget and n < max_frames:
            s = m.generate_frame(cur_t, cur_m, pos, args.temperature, args.topk)
            n += 1
            print(json.dumps({"shape": name, "rows": int(tokens.shape[1]), "frames": n, "elapsed": time.time() - t0, "threads": torch.get_num_threads()}), flush=True)
            cur_t = torch.cat([s.long(), torch.zeros(1, 1).long()], dim=1).unsqueeze(1)
            cur_m = torch.cat([torch.ones_like(s).bool(), torch.zeros(1, 1).bool()], dim=1).unsqueeze(1)
            pos = pos[:, -1:] + 1


def cpu_baseline(args):
    """Runs cpu_worker in a child with a hard wall-clock limit (killed by PID if it overruns) and turns its progress lines into the
    cpu_baseline object (the headline shape; `config1` = the same for BASELINE config 1's shape)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-budget", str(args.cpu_budget),
           "--cpu-frames", str(args.cpu_frames), "--temperature", str(args.temperature), "--topk", str(args.topk)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(args.cpu_threads))
    log(f"cpu baseline: oracle on {args.cpu_threads} host threads, budget {args.cpu_budget}s (+ the config-1 shape)")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
    try:
        out, _ = proc.communicate(timeout=2 * args.cpu_budget + 180)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
    every = [json.loads(l) for l in out.splitlines() if l.startswith("{")]

    def summarise(lines, what):
        if not lines:
            return dict(value=None, unit="frames/s", cores=args.cpu_threads, kind="port", sample=f"{what}: the oracle did not finish one frame within the budget")
        first, last = lines[0], lines[-1]
        decode = (last["frames"] - 1) / (last["elapsed"] - first["elapsed"]) if last["frames"] > 1 else None
        return dict(value=round(last["frames"] / last["elapsed"], 3), unit="frames/s", cores=last["threads"], kind="port",
                    decode_only_frames_per_s=round(decode, 3) if decode else None,
                    sample=f"oracle/csm_ref.py (PyTorch-CPU bf16 restatement of the reference -d cpu graph), {what}: "
                           f"frame 0 incl. prefill of {first['rows']} rows {first['elapsed']:.2f}s, {last['frames']} frames in {last['elapsed']:.1f}s, "
                           f"torch {torch.__version__}")
    res = summarise([l for l in every if l.get("shape") == "config2"], "same S=190 prompt as the headline (BASELINE config 2)")
    res["host"] = host_description(res["cores"])
    res["config1"] = summarise([l for l in every if l.get("shape") == "config1"], "BASELINE config 1 shape: 16 text tokens, no context, up to 25 frames")
    return res


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: this (GPU-free) parent starts the N rank processes itself with the
    torch.distributed environment and relays rank 0's JSON line.  Nothing here touches the GPU, and no process that has
    initialised the GPU is ever replaced by another program.  Every child is polled: the first one that fails takes the
    others down with it (by PID) instead of leaving them in a rendezvous until the distributed timeout."""
    import datetime
    import subprocess
    import tempfile
    from torch.distributed import TCPStore
    # The rendezvous store lives HERE, in the parent, for the whole job -- the way torchrun's agent hosts it: bound to a free port (port 0)
    # before any rank exists and never handed over, so no other job can take the port between "picked" and "rank 0 listens" (until round 5 a
    # probe socket was closed just before rank 0 started: VERDICT r5 next #4).  The ranks connect as clients (TORCHELASTIC_USE_AGENT_STORE).
    store = TCPStore("127.0.0.1", 0, None, True, timeout=datetime.timedelta(seconds=1800), wait_for_workers=False)
    port = store.port
    procs, out_f = [], tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_USE_AGENT_STORE="True", TORCHELASTIC_RESTART_COUNT="0",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out_f if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + float(os.environ.get("BENCH_SPAWN_TIMEOUT", "3000"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = abs(bad[0]) if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    out_f.seek(0)
    lines = [l for l in out_f.read().splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    if rc == 0 and not lines:
        rc = 3
    sys.exit(rc)


def timed_steps(model, B, n, temperature, topk, use_graph=True):
    """n frame steps between HIP events on the launching stream -> ms per step."""
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(st)
    for _ in range(n):
        model.step(B, temperature, topk, use_graph)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def batch32_leg(args, margs, sd, dev, n_steps, seed0=4000, barrier=None, with_refill=True):
    """BASELINE config 3 (and, per rank, config 4): B = 32 x the config-2 prompt, hipGraph-captured frame step.  Returns
    (result dict, wall seconds of the timed steps)."""
    from sesameai.models import Model
    T, K = args.temperature, args.topk
    B3 = 32
    tok, msk = synthetic_prompt(args, B3, margs.text_vocab_size, seed0=seed0)
    S = tok.shape[1]
    m3 = Model(margs, sd, device=str(dev), max_frames=4 * n_steps + 64, max_prefill_rows=B3 * S)
    m3.setup_caches(B3); m3.seed(77)
    pos = torch.arange(S).unsqueeze(0).repeat(B3, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m3.reset_caches(); m3.prefill(tok.to(dev), msk.to(dev), pos.to(dev)); m3.depth(B3, T, K, commit=True)
    torch.cuda.synchronize(); pre_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(5):
        m3.step(B3, T, K)
    if barrier is not None:
        barrier()
    t0 = time.perf_counter()
    ms = timed_steps(m3, B3, n_steps, T, K)
    wall = time.perf_counter() - t0
    frames, _ = m3.read_frames(B3)                    # raises if a launch gave up
    assert int(frames.min()) >= 0 and int(frames.max()) < margs.audio_vocab_size
    by = m3.bytes_per_frame(B3, S + 5 + n_steps / 2.0)
    # ---- a slot refilled while the other 31 keep generating (reference: one prefill per sentence, tts_service.py:191-207) ----
    refill = None
    if with_refill and os.environ.get("BENCH_SKIP_REFILL") != "1":
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # (a) csm_prefill_slot: the whole prompt + a batch-1 depth pass between two frame steps -- everybody waits
        torch.cuda.synchronize(); e0.record(st)
        m3.refill_slot(0, tok[0], msk[0], T, K)
        e1.record(st); torch.cuda.synchronize()
        stall_ms = e0.elapsed_time(e1)
        # (b) csm_refill_begin / csm_refill_advance: `per_step` layers of the prompt after EVERY frame step, back to back refills (the worst case:
        # a refill always in flight), frame 0 sampled in the batch
        per_step = max(1, 600 // S)
        slot = 1
        m3.refill_begin(slot, tok[slot], msk[slot])
        for _ in range(3):
            m3.step(B3, T, K); m3.refill_advance(per_step)
        torch.cuda.synchronize(); e0.record(st)
        n_ref, done_refills = max(n_steps, 16), 0
        for _ in range(n_ref):
            m3.step(B3, T, K)
            if m3.refill_advance(per_step):
                done_refills += 1
                slot = (slot + 1) % B3
                m3.refill_begin(slot, tok[slot], msk[slot])
        e1.record(st); torch.cuda.synchronize()
        ms_ref = e0.elapsed_time(e1) / n_ref
        while not m3.refill_advance(16):
            pass
        m3.read_frames(B3, m3.num_frames() - 1, 1)          # raises if a launch gave up / a position overflowed
        refill = {"prompt_rows": S, "layers_per_step": per_step, "ms_per_step_no_refill": round(ms, 4), "ms_per_step_with_a_refill_always_in_flight": round(ms_ref, 4),
                  "overhead_frac": round(ms_ref / ms - 1.0, 4), "refills_completed": done_refills, "steps": n_ref,
                  "stall_of_a_whole_prompt_between_two_steps_ms": round(stall_ms, 3),
                  "note": "csm_refill_begin/advance: the prompt's layers run beside the frame loop, its frame 0 is sampled by the batch's next step; "
                          "csm_prefill_slot (the stall figure) runs prompt + batch-1 depth pass in one go"}
    res = {"workload": f"CSM-1B B={B3}, S={S} prompt rows each, hipGraph frame step, {n_steps} timed steps", "ms_per_step": round(ms, 4),
           "refill_beside_the_loop": refill,
           "frames_per_s": round(B3 * 1e3 / ms, 1), "rtf_aggregate": round(B3 * 1e3 / ms / 12.5, 1),
           "prefill_plus_frame0_ms": round(pre_ms, 1), "roofline_frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "dominant_kernels": dominant_kernels(m3, B3, T, K, reps=5)}
    del m3
    return res, wall


def reference_loop_leg(args, margs, sd, dev, n_frames=60):
    """ms per frame when the model is driven exactly like the reference's loop (tts_service.py:224-241, generator.py:283-294):
    one Model.generate_frame call per frame with S = 1 rows built by the caller (int64 tokens / positions, bool mask, as the
    reference's torch.cat promotes them), `torch.all(sample == 0)` read on the host after every frame (a device sync), next row
    assembled with torch.cat on the device.  Also where the frame goes on the HOST: inside generate_frame (one C-ABI call,
    csm_generate_frame_s1), in the EOS check (= waiting for the GPU's frame) and in the caller's own tensor ops."""
    from sesameai.models import Model
    T, K = args.temperature, args.topk
    tok, msk = synthetic_prompt(args, 1, margs.text_vocab_size)
    S = tok.shape[1]
    m = Model(margs, sd, device=str(dev), max_frames=n_frames + 16, max_prefill_rows=S)
    m.setup_caches(1); m.seed(5)
    zeros_tok, zeros_msk = torch.zeros(1, 1, dtype=torch.long, device=dev), torch.zeros(1, 1, dtype=torch.bool, device=dev)
    host = {"generate_frame": 0.0, "eos_check": 0.0, "caller_ops": 0.0}
    iter_s = []                                        # wall time of every loop iteration of the accounted run (the caller's frame-to-frame time)

    def loop(n, account=False):
        m.reset_caches()
        curr_tokens, curr_mask = tok.to(dev), msk.to(dev)
        curr_pos = torch.arange(S, device=dev).unsqueeze(0)
        samples = []
        for i in range(n):
            t0 = time.perf_counter()
            sample = m.generate_frame(curr_tokens, curr_mask, curr_pos, T, K)
            t1 = time.perf_counter()
            if torch.all(sample == 0):
                break
            t2 = time.perf_counter()
            samples.append(sample)
            curr_tokens = torch.cat([sample, zeros_tok], dim=1).unsqueeze(1)                   # int32 + int64 -> int64, as in the reference
            curr_mask = torch.cat([torch.ones_like(sample).bool(), zeros_msk], dim=1).unsqueeze(1)
            curr_pos = curr_pos[:, -1:] + 1
            t3 = time.perf_counter()
            if account:
                iter_s.append(t3 - t0)
                if i > 0:
                    host["generate_frame"] += t1 - t0; host["eos_check"] += t2 - t1; host["caller_ops"] += t3 - t2
        return len(samples)
    loop(6)                                            # captures the graph
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = loop(n_frames, account=True)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    del m
    # frame 0 is the prompt (prefill + depth); the frames after it are what the loop costs per 80 ms of audio: mean wall time of the
    # loop's iterations 1.. (each ends when the caller has built the next row; the sync in the EOS check makes that the frame-to-frame time)
    per = sum(iter_s[1:]) * 1e3 / max(len(iter_s) - 1, 1)
    return {"workload": f"reference-style host loop (tts_service.py:224-241): generate_frame per frame, host EOS check each frame, B=1, S={S} prompt, {n} frames",
            "frames": n, "ms_per_frame_after_the_prompt": round(per, 4), "prompt_frame_ms": round(iter_s[0] * 1e3, 2), "whole_loop_wall_ms": round(wall * 1e3, 2),
            "rtf": round(80.0 / per, 2),
            "host_us_per_frame": {k: round(v * 1e6 / max(n - 1, 1), 1) for k, v in host.items()},
            "note": "eos_check = the host waiting for the frame (torch.all(sample == 0) synchronises); generate_frame = one ctypes call: stage kernel + graph launch + copy-out"}


def extras_legs(args, margs, sd, dev):
    """BASELINE configs 3 and 5 and the long-context single-stream step, each a short run inside this same process so
    the driver's one bench line carries them (`extras`).  Not part of `value`."""
    from sesameai.models import Model
    ex = {}
    T, K = args.temperature, args.topk
    # ---- config 3: B=32 x the config-2 prompt, hipGraph-captured frame step ----
    ex["config3"], _ = batch32_leg(args, margs, sd, dev, args.extra_steps)
    ex["reference_loop"] = reference_loop_leg(args, margs, sd, dev)
    # ---- long context, single stream, bf16: ms per frame at p ~ 1700 (the KV stream grows by 32 KB per position) ----
    tok, msk = synthetic_prompt(args, 1, margs.text_vocab_size, seed0=5000, segments=10, ctx_text=30, ctx_frames=100)
    S5 = tok.shape[1]
    n_long = 2040 - S5 - 8
    ml = Model(margs, sd, device=str(dev), max_frames=n_long + 16, max_prefill_rows=S5)
    ml.setup_caches(1); ml.seed(78)
    pos = torch.arange(S5).unsqueeze(0)
    ml.reset_caches(); ml.prefill(tok.to(dev), msk.to(dev), pos.to(dev)); ml.depth(1, T, K, commit=True)
    for _ in range(1700 - S5 - 20):
        ml.step(1, T, K)
    ms = timed_steps(ml, 1, 40, T, K)
    ex["b1_long_context"] = {"workload": f"CSM-1B B=1 bf16, S={S5} prompt rows, 40 frames timed at positions ~1680-1720", "ms_per_step": round(ms, 4),
                             "rtf": round(80.0 / ms, 2), "roofline_frac": round(ml.bytes_per_frame(1, 1700.0) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del ml
    # ---- config 5: fp8-e4m3 weight stream, S=1334 prompt, 375 frames, stateful Mimi decode every 10 frames ----
    from sesameai.generator import Generator
    from sesameai.mimi import MimiArgs, MimiCodec
    n5 = 375
    m5 = Model(margs, sd, device=str(dev), max_frames=n5 + 64, max_prefill_rows=S5, weights_dtype="fp8")
    codec = MimiCodec(MimiArgs(), None, device=str(dev), max_frames=n5 + 16)
    gen = Generator(m5, audio_tokenizer=codec)
    m5.seed(79); m5.prefix_reuse = False
    side = torch.cuda.Stream(device=dev)
    for rep in range(2):                                   # rep 0 captures the graph and warms the codec
        codec.reset_stream()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        first_ms, n_fr, pcm_n = None, 0, 0
        for fr in gen._frame_blocks(tok.to(dev), msk.to(dev), n5, T, K, 10):
            with torch.cuda.stream(side):
                pcm = codec.decode_stream(fr.to(dev).permute(1, 2, 0).contiguous())
            side.synchronize()
            n_fr += fr.shape[0]; pcm_n += pcm.shape[-1]
            if first_ms is None:
                first_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
    ex["config5"] = {"workload": f"CSM-1B B=1, fp8-e4m3 weight stream, S={S5} prompt rows (10 segments), {n_fr} frames, stateful Mimi decode every 10 frames",
                     "wall_ms": round(wall_ms, 1), "first_chunk_ms": round(first_ms, 1), "frames": n_fr, "pcm_samples": pcm_n,
                     "end_to_end_rtf": round(n_fr * 80.0 / wall_ms, 2),
                     "roofline_frac_end_to_end": round(m5.bytes_per_frame(1, S5 + n_fr / 2.0) * n_fr / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # the fp8 frame step alone, right after those 375 frames (positions ~1710-1750; `b1_long_context` times the bf16 step at
    # ~1680-1720): the like-for-like comparison of the two weight streams -- the end-to-end figure above also holds the
    # 1,334-row prefill and the codec
    if m5.num_frames() + 40 <= n5 + 64:
        ms8 = timed_steps(m5, 1, 40, T, K)
        ex["config5"].update({"frame_step_ms": round(ms8, 4), "frame_step_rtf": round(80.0 / ms8, 2),
                              "frame_step_positions": f"~{S5 + m5.num_frames() - 40}-{S5 + m5.num_frames()}",
                              "frame_step_vs_bf16": {"bf16_ms_per_step": ex["b1_long_context"]["ms_per_step"], "fp8_ms_per_step": round(ms8, 4),
                                                     "fp8_over_bf16_speed": round(ex["b1_long_context"]["ms_per_step"] / ms8, 4),
                                                     "note": "like for like: frame steps only, fp8 ~30 positions later than bf16; end_to_end_rtf also holds the 1,334-row prefill and the codec"}})
    del gen, codec, m5
    # ---- config 5 at B = 32: fp8 weight stream, 32 x the 1334-row prompt, frames timed at positions ~1340-1380 ----
    B5 = 32
    tokb, mskb = synthetic_prompt(args, B5, margs.text_vocab_size, seed0=6000, segments=10, ctx_text=30, ctx_frames=100)
    mb = Model(margs, sd, device=str(dev), max_frames=64, max_prefill_rows=B5 * S5, weights_dtype="fp8")
    mb.setup_caches(B5); mb.seed(80)
    posb = torch.arange(S5).unsqueeze(0).repeat(B5, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mb.reset_caches(); mb.prefill(tokb.to(dev), mskb.to(dev), posb.to(dev)); mb.depth(B5, T, K, commit=True)
    torch.cuda.synchronize(); pre_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(5):
        mb.step(B5, T, K)
    ms = timed_steps(mb, B5, 30, T, K)
    ex["config5_b32"] = {"workload": f"CSM-1B B={B5}, fp8-e4m3 weight stream, S={S5} prompt rows each, 30 frames timed at positions ~{S5 + 6}-{S5 + 36}",
                         "ms_per_step": round(ms, 4), "frames_per_s": round(B5 * 1e3 / ms, 1), "rtf_aggregate": round(B5 * 1e3 / ms / 12.5, 1),
                         "prefill_plus_frame0_ms": round(pre_ms, 1),
                         "roofline_frac": round(mb.bytes_per_frame(B5, S5 + 20.0) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del mb
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=125)       # 10 s of audio (config 2)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU")
    ap.add_argument("--ctx-text", type=int, default=40)
    ap.add_argument("--ctx-frames", type=int, default=125)
    ap.add_argument("--gen-text", type=int, default=24)
    ap.add_argument("--temperature", type=float, default=0.9)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mimi", action="store_true", help="skip the (untimed-region) Mimi decode report")
    ap.add_argument("--cpu-frames", type=int, default=64)
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-oracle frames to time")
    ap.add_argument("--cpu-threads", type=int, default=default_cpu_threads(),
                    help="host threads of the CPU baseline (default: min(32, logical CPUs, the container's CPU quota))")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--weights", choices=["bf16", "fp8"], default="bf16",
                    help="fp8 = OCP-e4m3 weight stream for the decode step (BASELINE config 5; not the headline)")
    ap.add_argument("--tiny", action="store_true", help="tiny shapes (plumbing check only; not a valid bench)")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-3 / config-5 / long-context legs (N=1 only)")
    ap.add_argument("--extra-steps", type=int, default=40)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args)
        return
    q = cpu_quota()
    # host-side set-up (synthetic weights) on no more threads than the container may run -- and, with N ranks on the node, no more than
    # this rank's SHARE of the quota: 8 ranks x 16 threads on a quota of 16 is the throttling DESIGN.md describes for one process
    share = max(1, (q or (os.cpu_count() or 1)) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))
    if torch.get_num_threads() > share:
        torch.set_num_threads(share)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args, sys.argv[1:])            # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus == 1:                        # under a launcher without --gpus: the launcher's world is the job
            args.gpus = world
        else:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # BENCH_SHARE_GPU0=1 (debug only, never a valid measurement): every rank uses cuda:0 over gloo, which lets the
    # multi-rank control flow (weight broadcast, barriers, max-over-ranks timing) be exercised on a 1-GPU box
    share0 = os.environ.get("BENCH_SHARE_GPU0") == "1"
    if share0:
        local_rank = 0
        # two processes on one GPU must not both run launches that need all 256 CUs resident at once (the persistent depth
        # decoder, the backbone attention block): half-resident twins would wait for each other until their bounded spins
        # give up.  One process per GPU -- the product setting -- never meets this.
        os.environ["CSM_PERSIST"] = "0"
        os.environ["CSM_BB_BLOCK"] = "0"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # BENCH_RCCL_WORLD1=1 (debug only): a ONE-rank RCCL process group on a 1-GPU box, and the N > 1 control flow on it (blob broadcasts,
    # barriers, the all-gathered rank records, the config-4 leg) -- the real backend's calls, which the gloo twin mode never makes
    multi = world > 1 or os.environ.get("BENCH_RCCL_WORLD1") == "1"
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:        # (BENCH_RCCL_WORLD1 without a launcher: any free port -- a fixed one collides when two runs share a box)
            import socket
            with socket.socket() as sk_:
                sk_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk_.getsockname()[1])
        for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k_, v_)
        if share0:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == world, (dist.get_world_size(), world)
        world = dist.get_world_size()              # n_gpus of the line = the ranks the process group really has

    from sesameai.models import Model, csm_1b_args, csm_tiny_args, state_dict_layout, synthetic_state_dict
    log(f"rank {rank}/{world}: building weights")
    margs = csm_tiny_args() if args.tiny else csm_1b_args()
    B = args.batch
    # ---- weights: rank 0 seeds them, the others receive one RCCL broadcast over xGMI --------
    bcast = None
    if multi:
        from sesameai.parallel import broadcast_state_dict, broadcast_flat
        stats = {}
        sd = broadcast_state_dict(margs, synthetic_state_dict(margs, seed=1234) if rank == 0 else None, dev, stats=stats)
        bcast = {"csm_bytes": stats["bytes"], "csm_ms": round(stats["ms"], 2), "csm_GBps": round(stats["bytes"] / stats["ms"] / 1e6, 1),
                 "collective": "one flat-blob broadcast (torch.distributed, backend " + dist.get_backend() + ")"}
    else:
        sd = synthetic_state_dict(margs, seed=1234)
    model = Model(margs, sd, device=str(dev), max_frames=args.steps + args.warmup + 8,
                  max_prefill_rows=B * (args.ctx_text + args.ctx_frames + 1 + args.gen_text), weights_dtype=args.weights)
    log("weights on device; creating caches")
    model.setup_caches(B)
    model.seed(1234 + rank)
    tokens, mask = synthetic_prompt(args, B, margs.text_vocab_size, seed0=2025 + rank * B)
    S = tokens.shape[1]
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    tok_d, mask_d, pos_d = tokens.to(dev), mask.to(dev), pos.to(dev)
    use_graph = not args.no_graph

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- prefill (+ frame 0), timed separately --------------------------------------------
    sync_all()
    t0 = time.perf_counter()
    model.reset_caches()
    model.prefill(tok_d, mask_d, pos_d)
    model.depth(B, args.temperature, args.topk, commit=True)
    torch.cuda.synchronize()
    prefill_ms = (time.perf_counter() - t0) * 1e3
    log(f"prefill + frame 0: {prefill_ms:.1f} ms; warmup {args.warmup} frames")
    for _ in range(args.warmup):
        model.step(B, args.temperature, args.topk, use_graph)
    # ---- timed region: exactly K frame steps ---------------------------------------------
    stream = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        model.step(B, args.temperature, args.topk, use_graph)
    ev1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    log(f"timed {args.steps} frames: {wall * 1e3:.1f} ms")
    ev_ms = ev0.elapsed_time(ev1)          # HIP events on the stream the frame graph runs on
    rank_wall = wall                       # this rank's own time; `wall` becomes the slowest rank's below
    if dist is not None:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        dist.barrier()
    frames, eos = model.read_frames(B)
    assert frames.shape[0] == 1 + args.warmup + args.steps
    assert int((frames < 0).sum()) == 0 and int(frames.max()) < margs.audio_vocab_size

    # ---- Mimi decode of the generated frames (reported separately, never inside `value`) ----
    mimi = None
    if not args.tiny and not args.no_mimi:
        from sesameai.mimi import MimiArgs, MimiCodec
        mimi_sd = None
        if multi:                                      # rank 0 seeds the codec's weights too; one more flat broadcast
            from sesameai.mimi import synthetic_state_dict as mimi_synthetic
            from sesameai.parallel import broadcast_named
            st_m = {}
            mimi_sd = broadcast_named(mimi_synthetic(MimiArgs()) if rank == 0 else None, dev, stats=st_m,
                                      template=None if rank == 0 else mimi_synthetic(MimiArgs()))
            bcast.update({"mimi_bytes": st_m["bytes"], "mimi_ms": round(st_m["ms"], 2)})
        codec = MimiCodec(MimiArgs(), mimi_sd, device=str(dev), max_frames=max(frames.shape[0], 16))
        codes = frames[:, 0, :].t().unsqueeze(0).contiguous().to(dev)          # (1, 32, T) of utterance 0
        T = codes.shape[2]

        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return sorted(ts)[len(ts) // 2]                        # median: one-off allocator / clock hiccups are not the kernel's time

        whole_ms = timed(lambda: codec.decode(codes))

        def stream_chunks():
            codec.reset_stream()
            for a in range(0, T, 10):
                codec.decode_stream(codes[:, :, a:a + 10])
        stream_ms = timed(stream_chunks)
        gen_ms = prefill_ms + wall * 1e3 * (frames.shape[0] - 1) / args.steps
        # codec roofline (~0.16 GB of fp32 weights per CALL + ~2 MB of activations per frame, HBM bound): weights of the
        # decode path = everything the call reads once (32 codebooks, projections, upsample, 8 transformer layers, SEANet decoder)
        mimi_bytes = float(codec.decode_weight_bytes()) if hasattr(codec, "decode_weight_bytes") else 0.16e9
        chunk_ms = stream_ms / max((T + 9) // 10, 1)
        mimi_roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_per_call": mimi_bytes,
                     "whole": {"frames": T, "ms": round(whole_ms, 3), "achieved": round((mimi_bytes + 2e6 * T) / (whole_ms * 1e-3) / 1e9, 1),
                               "frac": round((mimi_bytes + 2e6 * T) / (whole_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                     "chunk10": {"frames": 10, "ms": round(chunk_ms, 3), "achieved": round((mimi_bytes + 2e7) / (chunk_ms * 1e-3) / 1e9, 1),
                                 "frac": round((mimi_bytes + 2e7) / (chunk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                     "note": "latency-bound chain of 74 small launches, the middle ~70 replayed from a hipGraph for chunks (DESIGN.md, profiles/r04/mimi_ab.txt): < 3 % of end-to-end time and overlapped with the frame loop when streaming"}
        mimi = {"frames": T, "decode_whole_ms": round(whole_ms, 3), "decode_stream10_ms": round(stream_ms, 3), "roofline": mimi_roof,
                "ms_per_10_frame_chunk": round(stream_ms / max((T + 9) // 10, 1), 3),
                "end_to_end_ms": round(gen_ms + whole_ms, 2),
                "end_to_end_rtf": round(T * 80.0 / (gen_ms + whole_ms), 2)}
        # streaming surface end to end (Generator.generate_stream: prompt prefill -> frames -> Mimi on a side stream),
        # same prompt shape; time to the first 10-frame chunk and whole-utterance wall time
        if B == 1:
            from sesameai.generator import Generator, Segment
            gen = Generator(model, audio_tokenizer=codec)
            gq = torch.Generator().manual_seed(99)
            ctx = [Segment(speaker=1, text=torch.randint(0, margs.text_vocab_size, (args.ctx_text,), generator=gq).tolist(),
                           audio_codes=torch.randint(0, 2048, (32, args.ctx_frames), generator=gq))]
            text = torch.randint(0, margs.text_vocab_size, (args.gen_text,), generator=gq).tolist()
            model.prefix_reuse = False
            for rep in range(2):                                   # rep 0 warms the hipGraph of this (T, top-k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                first_ms, n_samples = None, 0
                for chunk in gen.generate_stream(text, 1, ctx, max_audio_length_ms=args.steps * 80.0,
                                                 temperature=args.temperature, topk=args.topk):
                    if first_ms is None:
                        first_ms = (time.perf_counter() - t0) * 1e3
                    n_samples += chunk.shape[0]
                torch.cuda.synchronize()
                total_ms = (time.perf_counter() - t0) * 1e3
            mimi["stream"] = {"frames": n_samples // 1920, "first_chunk_ms": round(first_ms, 2), "wall_ms": round(total_ms, 2),
                              "rtf": round(n_samples / 24.0 / total_ms, 2)}
        log(f"mimi: {mimi}")
        del codec

    ms_per_step = wall * 1e3 / args.steps
    value = world * B * args.steps / wall
    p_mean = S + args.warmup + args.steps / 2.0
    bytes_frame = model.bytes_per_frame(B, p_mean)
    t_frame = ev_ms * 1e-3 / args.steps
    achieved = bytes_frame / t_frame / 1e9
    # HBM/fabric bytes per frame-step launch from the PMC counters (FETCH_SIZE, x2 gfx950 correction), collected
    # with a separate `rocprofv3 --pmc FETCH_SIZE` pass of this same command and committed under profiles/
    traffic, traffic_src = None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if pmcs and B == 1 and not args.tiny and args.weights == "bf16":
        pj = json.load(open(pmcs[-1]))
        digest = csrc_digest()
        traffic_src = {"file": os.path.relpath(pmcs[-1], ROOT), "kernels_commit": pj.get("kernels_commit"), "csrc_digest_of_the_pass": pj.get("csrc_digest"),
                       "csrc_digest_now": digest,
                       "note": "separate rocprofv3 --pmc FETCH_SIZE pass of this command (x2 gfx950 correction), not measured in this run"}
        if pj.get("csrc_digest") == digest:
            traffic = pj.get("traffic_bytes_per_frame")
        else:                                 # the pass measured other kernel sources: not this run's traffic
            traffic_src["stale_traffic_bytes_per_frame"] = pj.get("traffic_bytes_per_frame")
            traffic_src["note"] += "; the pass was taken on DIFFERENT kernel sources (digest mismatch), so `traffic` is null"
    kernels = None if (args.tiny or args.weights != "bf16") else dominant_kernels(model, B, args.temperature, args.topk)
    paths = model.describe()               # which kernels ran + every CSM_* / MIMI_* switch set (csm_describe): the line says what it measured
    out = {
        "metric": "audio frames/sec", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.weights == "bf16" else "bf16 activations, fp8-e4m3 weights (decode step)",
        "data": "synthetic (seeded random weights of CSM-1B shapes, seeded random prompts)" + (" -- DEBUG: ranks share one GPU" if share0 else ""),
        "config": {"workload": ("tiny plumbing check" if args.tiny else
                                f"CSM-1B single utterance per GPU (B={B}), one voice-prompt segment, S={S} prompt rows, "
                                f"{args.steps} frames, T={args.temperature} top-k {args.topk}, "
                                f"{'hipGraph' if use_graph else 'eager'} frame step"),
                   "batch_per_gpu": B, "prompt_rows": S, "parallelism": f"replicas x{world}", "paths": paths},
        "rtf": round(value / 12.5, 2), "rtf_per_stream": round(value / 12.5 / (world * B), 2),
        "prefill_plus_frame0_ms": round(prefill_ms, 2), "mimi": mimi,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "frame step (1 backbone + 31 depth-decoder steps, unique weights + KV)",
                     "bytes_per_launch": bytes_frame, "launch_ms": round(t_frame * 1e3, 4),
                     "streamed_GBps": round((bytes_frame + 30 * 2 * 111.15e6 + 30 * 2 * 2.1e6) / t_frame / 1e9, 1),
                     "dominant_kernels": kernels},
    }
    if bcast is not None:
        out["weight_broadcast"] = bcast
    if dist is not None:
        # what makes the N > 1 line self-proving: how many ranks the process group really has and which physical GPU each one ran on
        # (index + PCI bus id: N ranks on N distinct devices), every rank's own step time and frame count, its view of the broadcast
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device_index": dev.index, "pci_bus_id": device_pci_bus_id(dev.index), "device_name": props.name,
                "ms_per_step": round(rank_wall * 1e3 / args.steps, 4), "frames": int(B * args.steps),
                "broadcast_GBps": bcast["csm_GBps"] if bcast else None, "host_threads": torch.get_num_threads()}
        every = [None] * world
        dist.all_gather_object(every, mine)
        out["ranks_seen"] = dist.get_world_size()
        out["ranks"] = every
        out["distinct_gpus"] = len({r["pci_bus_id"] or f"index {r['device_index']}" for r in every})
        out["collective_backend"] = dist.get_backend()
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            out["rccl_version"] = None
    if not multi and B == 1 and not args.tiny and not args.no_extras and args.weights == "bf16":
        del model
        out["extras"] = extras_legs(args, margs, sd, dev)
    elif multi and not args.tiny and not args.no_extras and args.weights == "bf16":
        # ---- BASELINE config 4: B = 32 per GPU on every rank (batch 32 x N sharded over the N GPUs, no per-step collective);
        #      aggregate = all ranks' frames / the slowest rank's time, like `value`
        del model
        res, wall4 = batch32_leg(args, margs, sd, dev, args.extra_steps, seed0=4000 + 32 * rank, barrier=sync_all, with_refill=False)
        t4 = torch.tensor([wall4], device=dev, dtype=torch.float64)
        dist.all_reduce(t4, op=dist.ReduceOp.MAX)
        res["workload"] = f"CSM-1B batch {32 * world} sharded over {world} GPUs (B=32 per GPU), S=190 prompt rows each, hipGraph frame step, {args.extra_steps} timed steps"
        res["aggregate_frames_per_s"] = round(world * 32 * args.extra_steps / float(t4.item()), 1)
        res["slowest_rank_ms_per_step"] = round(float(t4.item()) * 1e3 / args.extra_steps, 4)
        out["extras"] = {"config4": res}
    if dist is not None:
        dist.barrier()                     # every rank is through its GPU work; from here on nobody holds a GPU busy
        # RCCL prints a version banner through C stdio, which on a pipe is flushed at process exit -- AFTER the JSON line.  Flush it here, on
        # every rank, so that the line below is the last thing this job writes to stdout
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        # the oracle beside it, on this host's cores, after the timed region.  The other ranks wait for it ASLEEP -- a key in the
        # rendezvous store, polled with sleeps -- not in an RCCL barrier, whose waiters spin on host cores and GPU queues beside the
        # CPU run that is being timed
        out["cpu_baseline"] = cpu_baseline(args) if not (args.no_cpu_baseline or args.tiny) else None
        print(json.dumps(out), flush=True)
    if dist is not None:
        store = getattr(dist.distributed_c10d, "_get_default_store", lambda: None)()
        if store is not None:
            if rank == 0:
                store.set("bench_line_printed", "1")
            else:
                while True:
                    try:
                        if store.check(["bench_line_printed"]):
                            break
                    except Exception:
                        break
                    time.sleep(0.5)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

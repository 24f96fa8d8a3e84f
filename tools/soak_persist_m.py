#!/usr/bin/env python3
"""Determinism soak of the batched persistent depth decoder (csrc/dec_persist_m.cuh) at CSM-1B size.  The launch sums in fixed orders, so the same
seed must give the same frames bit for bit, run after run; a race in the exchange protocol (a stale buffer accepted, an LDS buffer
overwritten early) shows up as a difference, a give-up as an exception from read_frames.  Half of the runs have a second stream hammering
HBM next to the frame loop (uneven load: hand-offs that only work on an idle chip fail here).
    python tools/soak_persist_m.py [frames per run] [runs per batch size] [batch sizes, comma-separated]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from types import SimpleNamespace  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
margs = csm_1b_args()
sd = synthetic_state_dict(margs, seed=1234)
args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
side = torch.cuda.Stream()
junk = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device="cuda")
t_all = time.time()
batches = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2, 5, 9, 16, 17, 24, 32]      # (1: the batch-1 persistent launches)
for B in batches:
    tok, msk = bench.synthetic_prompt(args, B, margs.text_vocab_size, seed0=9000 + B)
    tok, msk = tok[:, :40], msk[:, :40]
    S = tok.shape[1]
    m = Model(margs, sd, max_frames=n_frames + 8, max_prefill_rows=B * S)
    m.setup_caches(B)
    assert m.fast_paths() & (2 if B > 1 else 1), "the persistent decoder launch is not in use"
    if B == 1: assert m.fast_paths() & 8, "the one-launch backbone layer is not in use"
    ref = None
    for run in range(n_runs):
        m.reset_caches(); m.seed(4321)
        m.prefill(tok, msk, torch.arange(S).unsqueeze(0).repeat(B, 1))
        m.depth(B, 0.9, 50, commit=True)
        noisy = run % 2 == 1
        for f in range(n_frames - 1):
            m.step(B, 0.9, 50)
            if noisy and f % 4 == 0:
                with torch.cuda.stream(side):               # 256 MB of copies beside the frame loop
                    junk[: junk.numel() // 2].copy_(junk[junk.numel() // 2:], non_blocking=True)
        frames, eos = m.read_frames(B)                      # raises if a launch gave up
        side.synchronize()
        assert int(frames.min()) >= 0 and int(frames.max()) < margs.audio_vocab_size
        if ref is None:
            ref = frames
        else:
            same = (frames == ref).all(dim=2).all(dim=1)
            assert bool(same.all()), f"B={B} run {run} ({'loaded' if noisy else 'idle'}): frames differ from run 0 from frame {int((~same).nonzero()[0])} on"
        print(f"B={B:2d} run {run} ({'beside a copy stream' if noisy else 'idle chip'}): {n_frames} frames ok", flush=True)
    del m
print(f"soak ok in {time.time() - t_all:.0f}s")

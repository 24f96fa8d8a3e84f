#!/usr/bin/env python3
"""Soak of the split-key attention's in-kernel merge (csrc/attn.cuh): batched decode steps merge the partial softmax states of a
(row, KV head) in the LAST key-range block to arrive, through sc1 write-through stores / sc1 loads and an agent-scope arrival counter.  The
merge launch (CSM_ATTN_MERGE=0) does the same arithmetic in the same order behind a kernel boundary, so the sampled frames of the two must be
bit-identical over thousands of steps, on an idle chip and beside a copy stream that hammers HBM (a partial read before it was complete, or a
stale line served from an L2, shows up as a difference).
    python tools/soak_attn_merge.py [frames per run] [runs] [batch sizes, comma-separated]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from types import SimpleNamespace  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batches = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [32, 8]
margs = csm_1b_args()
sd = synthetic_state_dict(margs, seed=1234)
args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
side = torch.cuda.Stream()
junk = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device="cuda")
t_all = time.time()
for B in batches:
    tok, msk = bench.synthetic_prompt(args, B, margs.text_vocab_size, seed0=9100 + B)
    tok, msk = tok[:, :100], msk[:, :100]
    S = tok.shape[1]
    models = {}
    for merge in ("1", "0"):                                  # the switch is read at csm_create
        os.environ["CSM_ATTN_MERGE"] = merge
        models[merge] = Model(margs, sd, max_frames=n_frames + 8, max_prefill_rows=B * S)
        models[merge].setup_caches(B)
    os.environ.pop("CSM_ATTN_MERGE")
    for run in range(n_runs):
        noisy = run % 2 == 1
        got = {}
        for merge, m in models.items():
            m.reset_caches(); m.seed(777 + run)
            m.prefill(tok, msk, torch.arange(S).unsqueeze(0).repeat(B, 1))
            m.depth(B, 0.9, 50, commit=True)
            for f in range(n_frames - 1):
                m.step(B, 0.9, 50)
                if noisy and f % 4 == 0:
                    with torch.cuda.stream(side):
                        junk[: junk.numel() // 2].copy_(junk[junk.numel() // 2:], non_blocking=True)
            got[merge], _ = m.read_frames(B)
            side.synchronize()
        same = (got["1"] == got["0"]).all(dim=2).all(dim=1)
        assert bool(same.all()), f"B={B} run {run}: in-kernel merge differs from the merge launch from frame {int((~same).nonzero()[0])} on"
        print(f"B={B:2d} run {run} ({'beside a copy stream' if noisy else 'idle chip'}): {n_frames} frames x {16 * B * 8} merges per step bit-identical", flush=True)
    del models
print(f"soak ok in {time.time() - t_all:.0f}s")

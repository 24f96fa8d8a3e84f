#!/usr/bin/env python3
"""Experiment: R independent single-utterance models on R HIP streams of one GPU vs one batched model (B = R).
The B = 1 frame step is a latency-bound chain of ~600 launches, so independent chains may overlap."""
import os, sys, time
# (round 2) the experiment is about the LAUNCH CHAIN: the all-CU launches (persistent depth decoder, backbone attention block)
# of two models must not share a GPU -- see DESIGN.md "constraint that comes with it"
os.environ.setdefault("CSM_PERSIST", "0"); os.environ.setdefault("CSM_BB_BLOCK", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
from sesameai.models import Model, csm_1b_args, synthetic_state_dict

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 60
args = csm_1b_args()
sd = synthetic_state_dict(args, seed=1234)
g = torch.Generator().manual_seed(1)
S = 190
tok = torch.zeros(1, S, 33, dtype=torch.long); msk = torch.zeros(1, S, 33, dtype=torch.bool)
tok[:, :64, 32] = torch.randint(0, 128256, (1, 64), generator=g); msk[:, :64, 32] = True
tok[:, 64:, :32] = torch.randint(0, 2048, (1, S - 64, 32), generator=g); msk[:, 64:, :32] = True
pos = torch.arange(S).unsqueeze(0)
models, streams = [], []
for r in range(R):
    m = Model(args, sd, max_frames=steps + 16, max_prefill_rows=S)
    m.setup_caches(1); m.seed(r)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        m.prefill(tok, msk, pos); m.depth(1, 0.9, 50, commit=True)
        for _ in range(5):
            m.step(1, 0.9, 50)
    models.append(m); streams.append(st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for m, st in zip(models, streams):
        with torch.cuda.stream(st):
            m.step(1, 0.9, 50)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{R} independent streams: {R * steps / dt:.1f} frames/s aggregate, {dt / steps * 1e3:.3f} ms per round of {R} frames")

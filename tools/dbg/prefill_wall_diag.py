#!/usr/bin/env python3
"""Wall time of every single S-row prefill (reset + prefill + synchronize), inputs on the host vs on the device: is the spread of
tools/prefill_prof.py's un-profiled figure (6 .. 80 ms at 1,334 rows) the product's or the tool's?   python3 tools/dbg/prefill_wall_diag.py [S] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sesameai-tts_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1334
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
margs = csm_1b_args()
m = Model(margs, synthetic_state_dict(margs, seed=1234), max_frames=8, max_prefill_rows=max(256, S))
m.setup_caches(1)
g = torch.Generator().manual_seed(3)
tok = torch.zeros(1, S, 33, dtype=torch.long)
tok[0, :, :32] = torch.randint(0, 2048, (S, 32), generator=g)
msk = torch.ones(1, S, 33, dtype=torch.bool); msk[0, :, 32] = False
pos = torch.arange(S).unsqueeze(0)
for name, (a, b, c) in (("host inputs", (tok, msk, pos)), ("device inputs", (tok.cuda(), msk.cuda(), pos.cuda())), ("host inputs again", (tok, msk, pos))):
    for _ in range(2):
        m.reset_caches(); m.prefill(a, b, c)
    torch.cuda.synchronize()
    ts, enq = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        m.reset_caches(); m.prefill(a, b, c)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3); enq.append((t1 - t0) * 1e3)
    print(f"S={S} {name}: wall ms " + " ".join(f"{t:.1f}" for t in ts))
    print(f"      host time until the call returned: " + " ".join(f"{t:.1f}" for t in enq))

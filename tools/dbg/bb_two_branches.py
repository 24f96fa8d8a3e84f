#!/usr/bin/env python3
"""Gate for VERDICT r4 next #2(b): would the batched backbone chain of a B = 32 frame step run faster as TWO 16-row branches that overlap
(one branch's latency-bound kernels -- q|k|v, split-key attention, o-proj, finishers -- under the other's gate/up / down streams)?
Measured without touching the engine: a backbone-only step is csm_prefill with S = 1 rows (embed + the 16-layer chain, no depth pass).
  (i)   one handle, 32 rows per step                         (what the frame step does today)
  (ii)  one handle, 16 rows per step                         (the per-branch cost if nothing overlapped: 2 x this is the serial bound)
  (iii) two handles of 16 rows, one stream each, both enqueued before the sync, branch B staggered behind branch A's first step
Eager launches on both sides (the same for all three), 200 steps at positions 190.., time per step from HIP events on each stream."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd")); sys.path.insert(0, ROOT)
import bench
from types import SimpleNamespace
from sesameai.models import Model, csm_1b_args, synthetic_state_dict

args = csm_1b_args()
sd = synthetic_state_dict(args, seed=1234)
ba = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
N = 200


def make(B, seed0):
    tok, msk = bench.synthetic_prompt(ba, B, args.text_vocab_size, seed0=seed0)
    S = tok.shape[1]
    m = Model(args, sd, max_frames=16, max_prefill_rows=B * S)
    m.setup_caches(B)
    m.prefill(tok, msk, torch.arange(S).unsqueeze(0).repeat(B, 1))
    row = torch.randint(0, 2048, (B, 1, 33), generator=torch.Generator().manual_seed(seed0)).cuda()
    rmask = torch.ones(B, 1, 33, dtype=torch.bool); rmask[:, 0, 32] = False
    return m, row, rmask.cuda(), S


def steps(m, row, rmask, S, B, n):
    for i in range(n):
        m.prefill(row, rmask, torch.full((B, 1), S + i, device="cuda", dtype=torch.int64))


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / N


m32 = make(32, 4000)
steps(*m32, 32, 20)
t32 = min(timed(lambda: steps(*m32, 32, N)) for _ in range(3))
del m32
a, b = make(16, 4000), make(16, 4016)
steps(*a, 16, 20); steps(*b, 16, 20)
t16 = min(timed(lambda: steps(*a, 16, N)) for _ in range(3))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    for i in range(N):
        with torch.cuda.stream(sa):
            a[0].prefill(a[1], a[2], torch.full((16, 1), a[3] + i, device="cuda", dtype=torch.int64))
        with torch.cuda.stream(sb):
            b[0].prefill(b[1], b[2], torch.full((16, 1), b[3] + i, device="cuda", dtype=torch.int64))
    sa.synchronize(); sb.synchronize()
t2 = min(timed(both) for _ in range(3))
print(f"backbone-only step, eager launches, ms per step:  one 32-row chain {t32:.4f} | one 16-row chain {t16:.4f} (two in series {2 * t16:.4f}) | "
      f"two 16-row chains on two streams {t2:.4f}  ->  two branches / one chain = {t2 / t32:.3f}")

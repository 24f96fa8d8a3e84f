#!/usr/bin/env python3
"""Two handles alive in one process, run alternately (same prompts, same seed): does a handle's second run equal its first?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch, bench
from types import SimpleNamespace
from sesameai.models import Model, csm_1b_args, synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
envs = (sys.argv[3] if len(sys.argv) > 3 else "1,0").split(",")
margs = csm_1b_args(); sd = synthetic_state_dict(margs, seed=1234)
args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
tok, msk = bench.synthetic_prompt(args, B, margs.text_vocab_size, seed0=9100 + B); tok, msk = tok[:, :100], msk[:, :100]; S = 100
ms = []
for e in envs:
    os.environ["CSM_ATTN_MERGE"] = e
    m = Model(margs, sd, max_frames=n_frames + 8, max_prefill_rows=B * S); m.setup_caches(B); ms.append(m)
def run(m, seed):
    m.reset_caches(); m.seed(seed)
    m.prefill(tok, msk, torch.arange(S).unsqueeze(0).repeat(B, 1)); m.depth(B, 0.9, 50, commit=True)
    for f in range(n_frames - 1):
        m.step(B, 0.9, 50)
    return m.read_frames(B)[0]
def fd(a, b):
    same = (a == b).all(dim=2).all(dim=1)
    return "identical" if bool(same.all()) else f"differ from frame {int((~same).nonzero()[0])}, rows {sorted(set((a != b).any(dim=2).nonzero()[:, 1].tolist()))[:12]}"
out = {}
for rep in range(3):
    for i, m in enumerate(ms):
        out[(i, rep)] = run(m, 778)
for i in range(len(ms)):
    print(f"handle {i} (merge={envs[i]}): run 1 vs run 0: {fd(out[(i, 1)], out[(i, 0)])}; run 2 vs run 0: {fd(out[(i, 2)], out[(i, 0)])}")
for rep in range(3):
    print(f"rep {rep}: handle 0 vs handle 1: {fd(out[(0, rep)], out[(1, rep)])}")
# seed change between runs, as the soak does
a = run(ms[0], 777); b = run(ms[1], 777); c = run(ms[0], 778); d = run(ms[1], 778)
print("seed 777: h0 vs h1:", fd(a, b), "| seed 778 after it: h0 vs h1:", fd(c, d), "| h0 seed 778 vs its first seed-778 run:", fd(c, out[(0, 0)]), "| h1:", fd(d, out[(1, 0)]))

#!/usr/bin/env python3
"""A B = 1 handle (k_dec_persist + k_bb_layer graph), a B = 8 and a B = 32 handle (k_dec_persist_m graphs of two shapes) and an fp8 B = 1 handle alive
in one process, run round-robin three times with the same seeds: every handle's runs must be identical (frame-step graphs of different shapes must not
disturb each other; profiles/r04/graph_memset_node_finding.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch, bench
from types import SimpleNamespace
from sesameai.models import Model, csm_1b_args, synthetic_state_dict
margs = csm_1b_args(); sd = synthetic_state_dict(margs, seed=1234)
args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
hs = []
for B, wd in ((1, "bf16"), (8, "bf16"), (32, "bf16"), (1, "fp8")):
    tok, msk = bench.synthetic_prompt(args, B, margs.text_vocab_size, seed0=9300 + B); tok, msk = tok[:, :80], msk[:, :80]
    m = Model(margs, sd, max_frames=40, max_prefill_rows=B * 80, weights_dtype=wd); m.setup_caches(B)
    hs.append((B, wd, m, tok, msk))
def run(B, m, tok, msk):
    m.reset_caches(); m.seed(5)
    m.prefill(tok, msk, torch.arange(80).unsqueeze(0).repeat(B, 1)); m.depth(B, 0.9, 50, commit=True)
    for _ in range(20):
        m.step(B, 0.9, 50)
    return m.read_frames(B)[0]
out = [[run(B, m, tok, msk) for (B, wd, m, tok, msk) in hs] for _ in range(3)]
ok = True
for i, (B, wd, *_r) in enumerate(hs):
    same = all(torch.equal(out[r][i], out[0][i]) for r in range(3))
    ok &= same
    print(f"handle B={B} {wd}: three runs {'identical' if same else 'DIFFER'}; codes in range: {int(out[0][i].min()) >= 0 and int(out[0][i].max()) < 2051}")
print("mixed handles ok" if ok else "MIXED HANDLES FAILED")

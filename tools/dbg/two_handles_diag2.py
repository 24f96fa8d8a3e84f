#!/usr/bin/env python3
"""merge=0 handle beside a merge=1 handle: what moves between its first and second run?  (logits after the first decode step, graph vs eager)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch, bench
from types import SimpleNamespace
from sesameai.models import Model, csm_1b_args, synthetic_state_dict
B = 32; use_graph = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
margs = csm_1b_args(); sd = synthetic_state_dict(margs, seed=1234)
args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
tok, msk = bench.synthetic_prompt(args, B, margs.text_vocab_size, seed0=9100 + B); tok, msk = tok[:, :100], msk[:, :100]; S = 100
os.environ["CSM_ATTN_MERGE"] = "0"; h0 = Model(margs, sd, max_frames=32, max_prefill_rows=B * S); h0.setup_caches(B)
os.environ["CSM_ATTN_MERGE"] = "1"; h1 = Model(margs, sd, max_frames=32, max_prefill_rows=B * S); h1.setup_caches(B)
def run(m, steps=2):
    m.reset_caches(); m.seed(778)
    m.prefill(tok, msk, torch.arange(S).unsqueeze(0).repeat(B, 1)); f0 = m.depth(B, 1.0, 1, commit=True).cpu()
    fr = [f0]
    for f in range(steps):
        m.step(B, 1.0, 1, use_graph); fr.append(m.last_frame(B).cpu())
    _, lg = m.depth(B, 1.0, 1, want_logits=True, commit=False)
    lasth = None
    return torch.stack(fr), lg.float().cpu()
a_fr, a_lg = run(h0)
run(h1)
b_fr, b_lg = run(h0)
c_fr, c_lg = run(h0)
print("graph" if use_graph else "eager", "| merge=0 handle: frames run0 vs run1 equal:", torch.equal(a_fr, b_fr), "run1 vs run2:", torch.equal(b_fr, c_fr),
      "| logits max|d| run0-run1:", float((a_lg - b_lg).abs().max()), "run1-run2:", float((b_lg - c_lg).abs().max()))
d = (a_fr != b_fr)
print("first differing (frame, row, codebook):", d.nonzero()[:5].tolist(), " frame-0 equal:", torch.equal(a_fr[0], b_fr[0]), "frame-1 equal:", torch.equal(a_fr[1], b_fr[1]))

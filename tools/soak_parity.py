#!/usr/bin/env python3
"""Randomised parity soak (tiny shapes): prompts of many lengths / batch sizes through every prefill path (GEMV rows,
32 x 32 MFMA tiles, 128 x 128 LDS-tiled, flash attention) and a few decode steps, compared with the CPU oracle.
Not part of the test-suite (minutes of CPU oracle time); run on a GPU box:  python tools/soak_parity.py [n_cases]"""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
from oracle import csm_ref as C
from sesameai.models import Model, csm_tiny_args, synthetic_state_dict

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
FULL = len(sys.argv) > 2 and sys.argv[2] == "full"          # CSM-1B shapes (slow oracle: a handful of cases)
if FULL:
    from sesameai.models import csm_1b_args
    shape, margs, gname = C.csm_1b(), csm_1b_args(), "csm1b_frames.pt"
else:
    shape, margs, gname = C.csm_tiny(), csm_tiny_args(), "tiny_frames.pt"
w = C.make_weights(shape, seed=1234)
sd = synthetic_state_dict(margs, seed=1234)
gold = torch.load(os.path.join(ROOT, "tests", "golden", gname))
noise = float(gold["bf16_vs_fp32_gap"].max())
m = Model(margs, sd, max_frames=16, max_prefill_rows=1400)
m.setup_caches(8)
rng = random.Random(7)
worst = 0.0
for case in range(n_cases):
    B = rng.choice([1, 1, 2, 3, 4, 7, 8]) if not FULL else rng.choice([1, 2, 3, 4])
    S = rng.choice([1, 2, 3, 5, 17, 31, 32, 33, 64, 65, 100, 127, 129, 160, 170])
    prompt = rng.random() < 0.5
    g = torch.Generator().manual_seed(1000 + case)
    nt = min(S, rng.randint(0, 6))
    tok = torch.zeros(B, S, 33, dtype=torch.long); msk = torch.zeros(B, S, 33, dtype=torch.bool)
    tok[:, :nt, 32] = torch.randint(0, shape.text_vocab_size, (B, nt), generator=g); msk[:, :nt, 32] = True
    tok[:, nt:, :32] = torch.randint(0, 2048, (B, S - nt, 32), generator=g); msk[:, nt:, :32] = True
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    m.reset_caches(); m.prefix_reuse = False
    if prompt:
        m.prefill_prompt(tok, msk)
    else:
        m.prefill(tok, msk, pos)
    om = C.OracleModel(shape, w); om.setup_caches(B)
    cur_t, cur_m, cur_p = tok, msk, pos
    for f in range(2):
        tr = C.FrameTrace()
        ref = om.generate_frame(cur_t, cur_m, cur_p, 1.0, 1, greedy=True, trace=tr)
        want = torch.stack(tr.logits, 0).float()                              # [32][B][V]
        out, logits = m.depth(B, 1.0, 1, forced=ref, want_logits=True, commit=False)
        d = (logits.float().cpu() - want).abs().max().item()
        if FULL:
            print(f"   frame {f}: max|dlogit| {d:.4f}", flush=True)
        worst = max(worst, d)
        assert d <= 2 * noise + 1e-3, f"case {case} B={B} S={S} prompt={prompt} frame {f}: max|dlogit| {d}"
        cur_t = torch.cat([ref.long(), torch.zeros(B, 1).long()], dim=1).unsqueeze(1)
        cur_m = torch.cat([torch.ones_like(ref).bool(), torch.zeros(B, 1).bool()], dim=1).unsqueeze(1)
        cur_p = cur_p[:, -1:] + 1
        m.prefill(cur_t, cur_m, cur_p)
    print(f"case {case:3d} B={B} S={S:4d} rows={B * S:5d} prompt={int(prompt)} ok", flush=True)
print(f"soak ok: {n_cases} cases, worst max|dlogit| {worst:.4f} (bound {2 * noise:.4f})")

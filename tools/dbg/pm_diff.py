"""debug: batched persistent decoder vs launch chain, per (codebook, utterance) max |dlogit|.  usage: pm_diff.py B [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch
import bench
from types import SimpleNamespace
from sesameai.models import Model, csm_1b_args, synthetic_state_dict
B = int(sys.argv[1]); NF = int(sys.argv[2]) if len(sys.argv) > 2 else 2
margs = csm_1b_args()
sd = synthetic_state_dict(margs, seed=1234)
tok, msk = bench.synthetic_prompt(SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24), B, margs.text_vocab_size, seed0=7000)
tok, msk = tok[:, :48], msk[:, :48]
S = tok.shape[1]
g = torch.Generator().manual_seed(B)
forced = torch.randint(0, 2048, (NF, B, 32), generator=g)
res = {}
for name, env in (("persistent", "1"), ("chain", "0")):
    os.environ["CSM_PERSIST_M"] = env
    m = Model(margs, sd, max_frames=16, max_prefill_rows=B * S)
    m.setup_caches(B)
    m.prefill(tok, msk, torch.arange(S).unsqueeze(0).repeat(B, 1))
    fr = []
    for f in range(NF):
        out, logits = m.depth(B, 1.0, 1, forced=forced[f], want_logits=True, commit=False)
        fr.append(logits.float().cpu())
        row = torch.zeros(B, 1, 33, dtype=torch.long); row[:, 0, :32] = forced[f]
        rmask = torch.ones(B, 1, 33, dtype=torch.bool); rmask[:, 0, 32] = False
        m.prefill(row, rmask, torch.full((B, 1), S + f))
    try:
        m.read_frames(B)
    except Exception as e:
        print("read_frames:", e)
    res[name] = fr
    del m
torch.set_printoptions(linewidth=250, precision=2)
for f in range(NF):
    d = (res["persistent"][f] - res["chain"][f]).abs().amax(dim=2)      # [32 cb][B]
    print(f"frame {f}: max {d.max():.4f}")
    print((d > 0.1).int())

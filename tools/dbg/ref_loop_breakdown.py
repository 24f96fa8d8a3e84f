#!/usr/bin/env python3
"""Where a frame of the reference-style host loop goes (tts_service.py:224-241 driven through Model.generate_frame, one call per
frame, `torch.all(sample == 0)` read on the host after every frame): the GPU's frame step, the host time inside generate_frame
(csm_generate_frame_s1: stage + graph launch + copy-out), and the caller's own torch ops between two calls.

    python tools/dbg/ref_loop_breakdown.py [n_frames]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sesameai-tts_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from bench import synthetic_prompt, timed_steps  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402
from types import SimpleNamespace  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
margs = csm_1b_args()
dev = torch.device("cuda", 0)
m = Model(margs, synthetic_state_dict(margs, seed=1234), device="cuda:0", max_frames=4 * n + 64, max_prefill_rows=256)
m.setup_caches(1); m.seed(5)
tok, msk = synthetic_prompt(args, 1, margs.text_vocab_size)
S = tok.shape[1]
T, K = 0.9, 50
zt, zm = torch.zeros(1, 1, dtype=torch.long, device=dev), torch.zeros(1, 1, dtype=torch.bool, device=dev)


def loop(n, sync_each=True, device_zeros=True):
    m.reset_caches()
    ct, cm, cp = tok.to(dev), msk.to(dev), torch.arange(S, device=dev).unsqueeze(0)
    t_call, t_sync, t_ops = 0.0, 0.0, 0.0
    out = []
    for i in range(n):
        a = time.perf_counter()
        s = m.generate_frame(ct, cm, cp, T, K)
        b = time.perf_counter()
        if sync_each and bool(torch.all(s == 0)):
            break
        c = time.perf_counter()
        out.append(s)
        if device_zeros:
            ct = torch.cat([s, zt.to(s.dtype)], dim=1).unsqueeze(1)
            cm = torch.cat([torch.ones_like(s).bool(), zm], dim=1).unsqueeze(1)
        else:                                       # the reference builds its zeros on the host every frame (two H2D copies)
            ct = torch.cat([s, torch.zeros(1, 1).long().to(dev)], dim=1).unsqueeze(1)
            cm = torch.cat([torch.ones_like(s).bool(), torch.zeros(1, 1).bool().to(dev)], dim=1).unsqueeze(1)
        cp = cp[:, -1:] + 1
        d = time.perf_counter()
        if i > 0:
            t_call += b - a; t_sync += c - b; t_ops += d - c
    return t_call, t_sync, t_ops, len(out)


loop(6)
m.reset_caches(); m.prefill(tok.to(dev), msk.to(dev), torch.arange(S, device=dev).unsqueeze(0)); m.depth(1, T, K, commit=True)
for _ in range(5):
    m.step(1, T, K)
graph_ms = timed_steps(m, 1, n, T, K)
print(f"graph loop (no host sync): {graph_ms:.4f} ms/frame")
for name, kw in (("reference loop, zeros kept on the device", dict()), ("reference loop, zeros built on the host each frame (as written)", dict(device_zeros=False)),
                 ("generate_frame per frame WITHOUT the host EOS check", dict(sync_each=False))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tc, ts, to, k = loop(n, **kw)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter(); loop(1, **kw); torch.cuda.synchronize(); first = time.perf_counter() - t0
    per = (wall - first) * 1e3 / (k - 1)
    print(f"{name}: {per:.4f} ms/frame = +{per - graph_ms:.4f} over the graph loop | host: generate_frame {tc * 1e6 / (k - 1):.1f} us, "
          f"EOS check (sync) {ts * 1e6 / (k - 1):.1f} us, caller's cat/ones/pos ops {to * 1e6 / (k - 1):.1f} us")

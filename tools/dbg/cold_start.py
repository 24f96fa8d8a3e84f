#!/usr/bin/env python3
"""Time to the first audio chunk of the FIRST request of a process, with and without load_csm_1b's warm-up:
    python3 tools/dbg/cold_start.py            CSM_NO_WARMUP=1 python3 tools/dbg/cold_start.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch  # noqa: E402
from sesameai.generator import Segment, load_csm_1b  # noqa: E402

t0 = time.perf_counter()
gen = load_csm_1b("cuda", synthetic=True)
torch.cuda.synchronize(); load_s = time.perf_counter() - t0
gq = torch.Generator().manual_seed(99)
ctx = [Segment(speaker=1, text=torch.randint(0, 128000, (40,), generator=gq).tolist(), audio_codes=torch.randint(0, 2048, (32, 125), generator=gq))]
text = torch.randint(0, 128000, (24,), generator=gq).tolist()
out = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); first = None
    for chunk in gen.generate_stream(text, 1, ctx, max_audio_length_ms=40 * 80.0, temperature=0.9, topk=50):
        if first is None:
            first = (time.perf_counter() - t0) * 1e3
    out.append(first)
print(f"warm-up {'off' if os.environ.get('CSM_NO_WARMUP') == '1' else 'on'}: load_csm_1b {load_s:.2f} s; first chunk of request 1 / 2 / 3: " + " / ".join(f"{x:.1f}" for x in out) + " ms")

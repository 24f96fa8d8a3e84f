#!/usr/bin/env python3
"""One csm_create on CSM-1B shapes and nothing else (bf16 weights, max batch 1): under rocprofv3 --kernel-trace --stats the kernel table is the
GPU time of handle creation -- weight re-tiling for the matrix-core path, the projected-embedding table, the layer-0 q|k|v table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sesameai-tts_amd")):
    sys.path.insert(0, p)
import torch
from sesameai.models import Model, csm_1b_args, synthetic_state_dict
m = Model(csm_1b_args(), synthetic_state_dict(csm_1b_args(), seed=1234), max_frames=16, max_prefill_rows=256)
torch.cuda.synchronize(); t0 = time.perf_counter()
m.setup_caches(1)
torch.cuda.synchronize()
print(f"csm_create (CSM-1B, max batch 1): {(time.perf_counter() - t0) * 1e3:.1f} ms wall")

#!/usr/bin/env python3
"""Bits of k_attn_flash (csm_op_attn, prompt form) under the library named by CSM_HIP_LIB: prints a checksum per case, so two builds
can be compared (`CSM_HIP_LIB=.../libcsm_hip_old.so python3 tools/dbg/flash_ab.py` vs the default build)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch  # noqa: E402
from sesameai import _abi  # noqa: E402

H, KV, hd, smax = 32, 8, 64, 2048
for S, start in ((190, 0), (37, 153), (65, 0), (1334, 0), (1334, 0), (700, 0), (1000, 0), (1200, 0), (1334, 100), (100, 1900)):
    g = torch.Generator().manual_seed(S + start)
    q = (torch.randn((S, H, hd), generator=g)).to(torch.bfloat16).cuda()
    kc = torch.randn((1, KV, smax, hd), generator=g).to(torch.bfloat16).cuda()
    vc = torch.randn((1, KV, smax, hd), generator=g).to(torch.bfloat16).cuda()
    pos = (start + torch.arange(S)).to(torch.int32).cuda()
    out = torch.zeros(S, H * hd, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(16, dtype=torch.float32, device="cuda")
    rc = _abi.lib.csm_op_attn(S, S, H, KV, hd, smax, 0, q.data_ptr(), kc.data_ptr(), vc.data_ptr(), pos.data_ptr(), out.data_ptr(),
                              part.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0
    b = out.cpu().view(torch.int16).numpy().tobytes()
    if len(sys.argv) > 1:
        torch.save(out.cpu(), f"{sys.argv[1]}_S{S}_{start}.pt")
    print(f"S={S} start={start}: sha {hashlib.sha256(b).hexdigest()[:16]}  finite {bool(torch.isfinite(out.float()).all())}")

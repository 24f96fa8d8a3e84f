"""Where k_attn_flash differs from exact attention on a batched prompt (bring-up aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_ops_gpu as T
from sesameai import _abi

class A: lib = _abi.lib
abi = A()
B, S, H, KV, hd, smax = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 190, 32, 8, 64, 320
g = torch.Generator().manual_seed(H + S)
M = B * S
q = T.rnd((B, S, H, hd), g)
kc, vc = T.rnd((B, KV, smax, hd), g), T.rnd((B, KV, smax, hd), g)
pos = torch.arange(S)[None, :].repeat(B, 1)
rep = H // KV
kk = kc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
vv = vc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
mask = torch.arange(smax)[None, None, :] <= pos[:, :, None]
sc = (q.double().transpose(1, 2) @ kk.double().transpose(-1, -2)) / hd ** 0.5
exact = (sc.masked_fill(~mask[:, None], float("-inf")).softmax(-1) @ vv.double()).transpose(1, 2)
out = torch.zeros(M, H * hd, dtype=torch.bfloat16, device="cuda")
part = torch.zeros(16, dtype=torch.float32, device="cuda")
qd, kd, vd, pd = T.dev(q), T.dev(kc), T.dev(vc), T.dev(pos.reshape(-1), torch.int32)
T._ck(abi, abi.lib.csm_op_attn(M, S, H, KV, hd, smax, 0, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), pd.data_ptr(), out.data_ptr(), part.data_ptr(), T.stream()))
torch.cuda.synchronize()
got = out.cpu().float().view(B, S, H, hd).double()
err = (got - exact).abs().amax(dim=3)            # [B][S][H]
bad = err > 0.02
print("max err", err.max().item(), "bad (b,row,head) count", int(bad.sum()), "of", bad.numel())
print("bad per batch index:", bad.sum(dim=(1, 2)).tolist())
print("bad per row group (32 rows):", [int(bad[:, i:i + 32].sum()) for i in range(0, S, 32)])
print("bad per head:", bad.sum(dim=(0, 1)).tolist())
idx = bad.nonzero()[:10].tolist()
print("first bad:", idx)

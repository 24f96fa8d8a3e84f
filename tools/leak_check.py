import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/sesameai-tts_amd")
from sesameai.models import Model, csm_tiny_args, synthetic_state_dict
from sesameai.mimi import MimiCodec, mimi_tiny_args
sd = synthetic_state_dict(csm_tiny_args(), seed=1)
free0 = None
for i in range(12):
    m = Model(csm_tiny_args(), sd, max_frames=32, max_prefill_rows=64, weights_dtype="fp8" if i % 2 else "bf16")
    m.setup_caches(4)
    m.setup_caches(2)           # re-create: the old handle must be destroyed
    c = MimiCodec(mimi_tiny_args(), None, max_frames=32)
    del m, c
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if i == 1: free0 = free
    print(i, free // (1 << 20), "MiB free")
assert abs(free - free0) < 64 << 20, "device memory leak across create/destroy"
print("no leak")

import sys, os, time, torch
sys.path.insert(0, "/root/repo/sesameai-tts_amd"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/sesameai-tts_amd")
from sesameai.mimi import MimiArgs, MimiCodec
codec = MimiCodec(MimiArgs(), None, device="cuda", max_frames=136)
g = torch.Generator().manual_seed(0)
codes = torch.randint(0, 2048, (1, 32, 125), generator=g).cuda()
for _ in range(3): codec.decode(codes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): codec.decode(codes)
torch.cuda.synchronize(); print("whole 125 frames ms", (time.perf_counter() - t0) * 100)
c10 = codes[:, :, :10].contiguous()
for _ in range(3): codec.decode(c10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): codec.decode(c10)
t1 = time.perf_counter()
torch.cuda.synchronize(); print("10 frames ms", (time.perf_counter() - t0) * 50, "| host time to enqueue one decode ms", (t1 - t0) * 50)
for _ in range(3):                                    # one decode at a time, GPU idle before each: what a streaming chunk sees
    torch.cuda.synchronize(); t0 = time.perf_counter(); codec.decode(c10); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e3
print("10 frames, one call from an idle GPU ms", one)
wav = torch.randn(1, 1, 24000 * 10, generator=torch.Generator().manual_seed(1)).cuda() * 0.1
for _ in range(2): codec.encode(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): codec.encode(wav)
torch.cuda.synchronize(); print("encode 10 s (125 frames) ms", (time.perf_counter() - t0) * 200)

import torch,sys
for S,st in ((700,0),(1334,0)):
    a=torch.load(f"gpurun_out/r4p/new_S{S}_{st}.pt").float().view(S,32,64); b=torch.load(f"gpurun_out/r4p/old_S{S}_{st}.pt").float().view(S,32,64)
    d=(a!=b)
    rows=d.any(-1).any(-1).nonzero().flatten()
    print(S, "differing rows", len(rows), rows[:20].tolist(), "...", rows[-5:].tolist())
    print(" max abs diff", (a-b).abs().max().item(), "heads differing", d.any(-1).any(0).nonzero().flatten().tolist()[:40])
    r=rows[0].item(); print(" first row", r, "elements differing", d[r].sum().item(), "of", 32*64)

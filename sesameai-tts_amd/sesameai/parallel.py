"""Multi-GPU plumbing for the replica layout: utterances are independent, so
each GPU holds a full weight replica + private KV caches and there is NO per-step collective.
The only communication is one broadcast of the weights at start-up (RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" on CPU for the tests), packed into a single flat
buffer so it is one large collective instead of ~200 small ones."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .models import ModelArgs, state_dict_layout


def pack_layout(args: ModelArgs) -> Tuple[List[Tuple[str, Tuple[int, ...], int]], int]:
    """(name, shape, element offset) of every tensor inside the flat bf16 blob, 64-element aligned."""
    out, off = [], 0
    for name, shp in state_dict_layout(args):
        n = 1
        for s in shp:
            n *= s
        out.append((name, shp, off))
        off += (n + 63) // 64 * 64
    return out, off


def broadcast_state_dict(args: ModelArgs, sd: Optional[Dict[str, torch.Tensor]], device: torch.device,
                         src: int = 0) -> Dict[str, torch.Tensor]:
    """Rank ``src`` passes its state dict, every other rank passes None; all ranks return
    tensors that are views into one device-resident flat blob."""
    layout, total = pack_layout(args)
    blob = torch.empty(total, dtype=torch.bfloat16, device=device)
    if dist.get_rank() == src:
        assert sd is not None
        for name, shp, off in layout:
            n = sd[name].numel()
            blob[off:off + n].copy_(sd[name].reshape(-1).to(torch.bfloat16))
    dist.broadcast(blob, src=src)
    return {name: blob[off:off + int(torch.tensor(shp).prod())].view(shp) for name, shp, off in layout}


def shard_utterances(n_utterances: int, world: int, rank: int) -> range:
    """contiguous blocks of ceil(N/world) utterances per GPU."""
    per = (n_utterances + world - 1) // world
    return range(min(n_utterances, rank * per), min(n_utterances, (rank + 1) * per))

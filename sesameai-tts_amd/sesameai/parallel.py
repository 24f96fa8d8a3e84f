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


def broadcast_flat(blob: torch.Tensor, src: int = 0, stats: Optional[dict] = None) -> None:
    """ONE collective over the whole blob; ``stats`` receives its size and its duration (barrier before, device sync after:
    the time of the slowest rank as this rank sees it)."""
    import time
    if stats is not None:
        if blob.is_cuda:
            torch.cuda.synchronize(blob.device)
        dist.barrier()
        t0 = time.perf_counter()
    dist.broadcast(blob, src=src)
    if stats is not None:
        if blob.is_cuda:
            torch.cuda.synchronize(blob.device)
        stats["ms"] = (time.perf_counter() - t0) * 1e3
        stats["bytes"] = blob.numel() * blob.element_size()


def broadcast_state_dict(args: ModelArgs, sd: Optional[Dict[str, torch.Tensor]], device: torch.device,
                         src: int = 0, stats: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """Rank ``src`` passes its state dict, every other rank passes None; all ranks return
    tensors that are views into one device-resident flat blob."""
    layout, total = pack_layout(args)
    blob = torch.empty(total, dtype=torch.bfloat16, device=device)
    if dist.get_rank() == src:
        assert sd is not None
        for name, shp, off in layout:
            n = sd[name].numel()
            blob[off:off + n].copy_(sd[name].reshape(-1).to(torch.bfloat16))
    broadcast_flat(blob, src, stats)
    return {name: blob[off:off + int(torch.tensor(shp).prod())].view(shp) for name, shp, off in layout}


def broadcast_named(sd: Optional[Dict[str, torch.Tensor]], device: torch.device, src: int = 0, stats: Optional[dict] = None,
                    template: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """The same for any fp32 state dict whose names / shapes every rank knows (the Mimi codec's): ``src`` passes the tensors,
    the others pass ``template`` (tensors of the right shapes; their values are ignored)."""
    ref = sd if dist.get_rank() == src else template
    assert ref is not None
    names = sorted(ref)
    offs, off = {}, 0
    for n in names:
        offs[n] = off
        off += (ref[n].numel() + 63) // 64 * 64
    blob = torch.zeros(off, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        for n in names:
            blob[offs[n]:offs[n] + ref[n].numel()].copy_(ref[n].reshape(-1).to(torch.float32))
    broadcast_flat(blob, src, stats)
    return {n: blob[offs[n]:offs[n] + ref[n].numel()].view(ref[n].shape) for n in names}


def shard_utterances(n_utterances: int, world: int, rank: int) -> range:
    """contiguous blocks of ceil(N/world) utterances per GPU."""
    per = (n_utterances + world - 1) // world
    return range(min(n_utterances, rank * per), min(n_utterances, (rank + 1) * per))

"""Request sharding over the GPUs of one node.

Utterances are independent: weights are replicated, every rank runs its own engine on its own
requests and there is NO collective inside the compute.  The only exchange is the final gather of
variable-length waveforms to rank 0 (`torch.distributed.gather`; backend "nccl" == RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The reference has no distributed code at all.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def lpt_partition(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first: deal requests (by predicted cost = text-token count) to the currently
    least-loaded rank.  Deterministic (ties -> lowest index / lowest rank); returns per-rank request indices
    in their ORIGINAL relative order so a rank's batches keep the caller's ordering."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += float(costs[i])
    return [sorted(p) for p in parts]


def waves(indices: Sequence[int], batch: int) -> List[List[int]]:
    """Split one rank's request list into engine batches of at most `batch` rows."""
    return [list(indices[i:i + batch]) for i in range(0, len(indices), batch)]


def gather_waveforms(local_wavs: List[np.ndarray], local_indices: List[int], n_total: int,
                     device: Optional[torch.device] = None, group=None) -> Optional[List[np.ndarray]]:
    """Gather every rank's (index, waveform) pairs on rank 0 and return them in request order there
    (None elsewhere).  Two collectives: lengths, then one padded float tensor per rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    # every rank owns at most ceil(n_total / 1) entries; meta row i = (request index, length), -1 = unused
    cap = n_total
    meta = torch.full((cap, 2), -1, dtype=torch.int64, device=device)
    for j, (idx, w) in enumerate(zip(local_indices, local_wavs)):
        meta[j, 0], meta[j, 1] = int(idx), int(w.shape[0])
    metas = [torch.empty_like(meta) for _ in range(world)] if rank == 0 else None
    dist.gather(meta, metas, dst=0, group=group)
    mx = torch.tensor([max([int(w.shape[0]) for w in local_wavs], default=0)], dtype=torch.int64, device=device)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    L = max(1, int(mx.item()))
    nmax = torch.tensor([len(local_wavs)], dtype=torch.int64, device=device)
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=group)
    n = max(1, int(nmax.item()))
    buf = torch.zeros(n, L, dtype=torch.float32, device=device)
    for j, w in enumerate(local_wavs):
        buf[j, : w.shape[0]] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(device)
    bufs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, bufs, dst=0, group=group)
    if rank != 0:
        return None
    out: List[Optional[np.ndarray]] = [None] * n_total
    for r in range(world):
        m = metas[r].cpu().numpy()
        b = bufs[r].cpu().numpy()
        for j in range(cap):
            idx, ln = int(m[j, 0]), int(m[j, 1])
            if idx >= 0:
                out[idx] = b[j, :ln].copy()
    return out


def gather_padded(wav: torch.Tensor, lengths: torch.Tensor, group=None):
    """The request shard's exchange for DEVICE-resident results: every rank holds one padded batch `wav (B, L) float32` and
    `lengths (B,) int64` on its own device; rank 0 receives all of them with two `torch.distributed.gather` calls (nccl =
    RCCL over xGMI: one (B, L) tensor per rank, ~0.96 MB per 10 s utterance) and returns `(wavs (world, B, L), lens (world, B))`
    still on the device; other ranks return None.  Shapes must agree across ranks (the bench's fixed-length utterances; the
    variable-length serving path is `gather_waveforms`)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    wav = wav.contiguous()
    lengths = lengths.to(device=wav.device, dtype=torch.int64).contiguous()
    wl = [torch.empty_like(wav) for _ in range(world)] if rank == 0 else None
    ll = [torch.empty_like(lengths) for _ in range(world)] if rank == 0 else None
    dist.gather(wav, wl, dst=0, group=group)
    dist.gather(lengths, ll, dst=0, group=group)
    if rank != 0:
        return None
    return torch.stack(wl), torch.stack(ll)

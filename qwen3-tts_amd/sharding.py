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


def gather_waveforms_device(local_wavs: Sequence[torch.Tensor], local_indices: Sequence[int], n_total: int, group=None
                            ) -> Optional[List[torch.Tensor]]:
    """The request shard's exchange for VARIABLE-length, device-resident results: every rank holds its own requests' waveforms as
    1-D float32 tensors on its device (what `decode_padded` returns, cut to length); rank 0 receives all of them and returns a list
    of `n_total` tensors in request order (views of the gathered buffers, still on the device); other ranks return None.  Nothing
    passes through host memory: the rows are packed into one padded (n_max, L_max) tensor per rank by `pad_sequence` on the device
    and travel in ONE `torch.distributed.gather` (nccl = RCCL over xGMI; gloo on CPU tensors in the tests); two small collectives
    agree on n_max / L_max first, one carries the (index, length) table."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if len(local_wavs):
        device = local_wavs[0].device
    else:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    dims = torch.tensor([len(local_wavs), max([int(w.shape[0]) for w in local_wavs], default=0)], dtype=torch.int64, device=device)
    dist.all_reduce(dims, op=dist.ReduceOp.MAX, group=group)
    n, L = max(1, int(dims[0])), max(1, int(dims[1]))
    meta = torch.full((n, 2), -1, dtype=torch.int64)
    for j, (idx, w) in enumerate(zip(local_indices, local_wavs)):
        meta[j, 0], meta[j, 1] = int(idx), int(w.shape[0])
    meta = meta.to(device)
    buf = torch.zeros(n, L, dtype=torch.float32, device=device)
    if len(local_wavs):
        packed = torch.nn.utils.rnn.pad_sequence([w.to(torch.float32) for w in local_wavs], batch_first=True)   # one device op
        buf[: packed.shape[0], : packed.shape[1]] = packed
    metas = [torch.empty_like(meta) for _ in range(world)] if rank == 0 else None
    bufs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(meta, metas, dst=0, group=group)
    dist.gather(buf, bufs, dst=0, group=group)
    if rank != 0:
        return None
    out: List[Optional[torch.Tensor]] = [None] * n_total
    for r in range(world):
        m = metas[r].cpu().numpy()
        for j in range(n):
            idx, ln = int(m[j, 0]), int(m[j, 1])
            if idx >= 0:
                out[idx] = bufs[r][j, :ln]
    return out


def gather_waveforms(local_wavs: List[np.ndarray], local_indices: List[int], n_total: int,
                     device: Optional[torch.device] = None, group=None) -> Optional[List[np.ndarray]]:
    """`gather_waveforms_device` for callers that hold host numpy waveforms (what the API returns, IM:840) and want numpy back on
    rank 0: the arrays are staged on `device` (nccl needs device tensors), gathered, and copied back once."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    dev_wavs = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(device) for w in local_wavs]
    out = gather_waveforms_device(dev_wavs, local_indices, n_total, group)
    if out is None:
        return None
    return [None if w is None else w.cpu().numpy().copy() for w in out]


def engine_partition(costs: Sequence[float], world_size: int, engines_per_rank: int) -> List[List[List[int]]]:
    """LPT over (rank, engine) bins: `parts[r][e]` = the requests engine e of rank r runs.  With E engines per GPU (each its own
    weights, stream and host thread -- a frame step leaves most CUs idle, so a second stream fills the gaps) the balancing unit is
    the engine, not the rank."""
    flat = lpt_partition(costs, world_size * engines_per_rank)
    return [[flat[r * engines_per_rank + e] for e in range(engines_per_rank)] for r in range(world_size)]


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

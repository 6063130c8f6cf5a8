"""Multi-GPU layout of McSAS.analyse: repetitions are independent (mcsas.py:214-262 is a serial
`for nr in range(numReps)` over chains that share only read-only data), so each rank (one process
per GPU) runs a contiguous block of reps and ONE all-gather at the end assembles the reference's
`(…, numReps)` arrays in rep order.  No collective on the data path.

Backend: torch.distributed — "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_reps(n_reps: int, world_size: int, rank: int):
    """Contiguous block of repetitions for `rank`: (first_rep, count).  Blocks differ by at most
    one rep; concatenating the blocks in rank order restores rep order."""
    base, extra = divmod(int(n_reps), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def _gather_var(t, counts, dist, device):
    """all_gather of per-rank tensors whose leading dim differs: pad to the max count."""
    import torch
    m = max(counts)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=device)
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in counts]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def gather_results(local: dict, n_reps: int, device=None, force=False):
    """`local`: per-rank arrays with the REP AXIS FIRST (e.g. contribs as (R_local, N, P)).
    Returns the same keys with all reps, in rep order, on every rank.  One fused all-gather:
    everything is packed into a single (R_local, width) float64 buffer.  `force`: go through the
    collective also at world size 1 (a one-GPU box exercising the RCCL path)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return {k: np.asarray(v) for k, v in local.items()}
    ws, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_reps(n_reps, ws, r)[1] for r in range(ws)]
    keys = sorted(local)
    shapes = {k: np.asarray(local[k]).shape[1:] for k in keys}
    widths = {k: int(np.prod(shapes[k])) if len(shapes[k]) else 1 for k in keys}
    rl = counts[rank]
    # explicit widths: a rank that owns no repetition (world_size > n_reps) still joins the collective
    packed = np.concatenate([np.asarray(local[k], dtype=np.float64).reshape(rl, widths[k]) for k in keys], axis=1)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(packed)).to(device)
    full = _gather_var(t, counts, dist, device).cpu().numpy()
    out, col = {}, 0
    for k in keys:
        out[k] = full[:, col:col + widths[k]].reshape((n_reps,) + tuple(shapes[k]))
        col += widths[k]
    return out

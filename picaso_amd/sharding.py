"""Wavelength sharding across the GPUs of one node (one process per GPU).

Every function on the hot path is pointwise in wavelength, so the grid is cut
into contiguous blocks, one per rank, with no exchange inside the solve; the only collective is
the all-gather of the final spectrum shards (RCCL over xGMI when the backend is "nccl", gloo in
the CPU tests).  Pure host logic: usable without a GPU.

One HIP runtime per process: PyTorch-ROCm wheels bundle their own ``libamdhip64``; import torch (and
select the device) BEFORE the first ``picaso_amd`` call in a process that uses both, as ``bench.py``
does -- ``picaso_amd._lib`` then binds ``libpicaso_hip.so`` to the runtime torch has mapped.
"""
import numpy as np


def shard_bounds(nwno, world):
    """Contiguous [lo, hi) wavelength blocks, sizes differing by at most one."""
    base, rem = divmod(int(nwno), int(world))
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_of(nwno, world, rank):
    return shard_bounds(nwno, world)[rank]


def all_gather_spectrum(local, nwno, dist=None, group=None):
    """Gather per-rank spectrum shards (last axis = wavelength) into the full spectrum on every
    rank.  `local` is a numpy array (gloo / single process) or a torch tensor on the rank's GPU
    (nccl).  Ragged shards are padded to the largest shard for the collective and trimmed after."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    import torch
    world = dist.get_world_size(group)
    bounds = shard_bounds(nwno, world)
    nmax = max(hi - lo for lo, hi in bounds)
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local
    lead = tuple(t.shape[:-1])
    pad = torch.zeros(lead + (nmax,), dtype=t.dtype, device=t.device)
    pad[..., : t.shape[-1]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
    pieces = [out[r][..., : hi - lo] for r, (lo, hi) in enumerate(bounds)]
    full = torch.cat(pieces, dim=-1)
    return full.numpy() if is_np else full

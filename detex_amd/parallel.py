"""Multi-GPU sharding of the hot path: one process per GPU (torchrun), no
data-path collective.  The only exchange is one broadcast of the detector basis
(RCCL over xGMI with backend 'nccl'; 'gloo' in the CPU tests).

Units of work:
  * detection scan: (station, day) units, dealt round-robin over ranks;
  * similarity matrix: a contiguous range of condensed pair indices per rank
    (dtx_xcorr_allpairs(pair_begin, pair_end)); every pair costs the same, so
    equal-sized ranges are balanced.
"""
import numpy as np
import torch


def dist_info():
    """(rank, world) of the default process group, (0, 1) if not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_units(nunits, rank, world):
    """Indices of the (station, day) units this rank scans: round-robin, so
    consecutive days of one station spread over all GPUs."""
    return list(range(rank, nunits, world))


def shard_chunks(nchunks, rank, world, per_unit=24):
    """Chunk indices of one station this rank scans.  Units are runs of `per_unit`
    consecutive chunks (a day of hourly chunks: they go through the kernels as one launch),
    dealt round-robin; with fewer units than ranks the chunks are cut into `world` contiguous
    ranges instead, so that a single day still spreads over all GPUs."""
    nunits = (nchunks + per_unit - 1) // per_unit
    if nunits >= world:
        out = []
        for u in shard_units(nunits, rank, world):
            out.extend(range(u * per_unit, min((u + 1) * per_unit, nchunks)))
        return out
    b, e = shard_pairs(nchunks, rank, world)
    return list(range(b, e))


def gather_rows(rows):
    """All ranks contribute a picklable object (their detection rows, histograms); every rank
    gets the list over ranks.  Host objects, a few KB: not on the data path."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world == 1:
        return [rows]
    out = [None] * world
    dist.all_gather_object(out, rows)
    return out


def shard_pairs(npairs, rank, world):
    """[begin, end) condensed-pair range of this rank; ranges tile [0, npairs)."""
    base, rem = divmod(int(npairs), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def broadcast_basis(U, det_off, n, device, src=0):
    """Rank `src` holds U [Ktot, n] float32 and det_off [D+1]; every rank returns
    both (U as a tensor on `device`).  Two small broadcasts: the offsets (so the
    other ranks can size the buffer), then the basis itself."""
    import torch.distributed as dist
    rank, world = dist_info()
    if not (dist.is_available() and dist.is_initialized()):
        return torch.as_tensor(U, dtype=torch.float32, device=device), np.asarray(det_off, np.int32)
    if rank == src:
        off = torch.as_tensor(np.asarray(det_off, dtype=np.int64), device=device)
        meta = torch.tensor([off.numel(), int(n)], dtype=torch.int64, device=device)
    else:
        meta = torch.zeros(2, dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src)
    if rank != src:
        off = torch.zeros(int(meta[0]), dtype=torch.int64, device=device)
    dist.broadcast(off, src=src)
    Ktot = int(off[-1])
    if rank == src:
        Ut = torch.as_tensor(U, dtype=torch.float32, device=device).contiguous()
        assert tuple(Ut.shape) == (Ktot, int(meta[1]))
    else:
        Ut = torch.empty((Ktot, int(meta[1])), dtype=torch.float32, device=device)
    dist.broadcast(Ut, src=src)
    return Ut, off.cpu().numpy().astype(np.int32)


def gather_condensed(parts, ranges, npairs, device):
    """All-gather the per-rank slices of a condensed vector (cc / lag / subsamp):
    every rank contributes its [begin, end) slice; returns the full vector on
    every rank.  Used only to assemble the similarity matrix after the scan."""
    import torch.distributed as dist
    rank, world = dist_info()
    if world == 1:
        return parts
    width = max(e - b for b, e in ranges)
    buf = torch.zeros(width, dtype=parts.dtype, device=device)
    b, e = ranges[rank]
    buf[:e - b] = parts[b:e]
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    full = torch.zeros(npairs, dtype=parts.dtype, device=device)
    for r, (b, e) in enumerate(ranges):
        full[b:e] = out[r][:e - b]
    return full

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

# collectives issued through this module since import (tests assert that a SubSpace.detex call
# costs a fixed number of them, whatever the number of stations)
COLLECTIVES = {'broadcast': 0, 'all_gather': 0, 'all_gather_object': 0, 'broadcast_object': 0}


def dist_info():
    """(rank, world) of the default process group, (0, 1) if not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def dist_active():
    """True when a default process group exists -- also one of world size 1.  The product's
    collectives (broadcast_banks, gather_rows, gather_condensed) run whenever this holds, so a
    single-GPU box exercises RCCL exactly as an 8-GPU node does (only the peer count differs)."""
    import torch.distributed as dist
    return bool(dist.is_available() and dist.is_initialized())


def local_device():
    """The GPU of this process: the launcher's LOCAL_RANK under a process group, else 0."""
    import os
    return int(os.environ.get('LOCAL_RANK', 0)) if dist_active() else 0


# what bind_to_device did in this process (bench.py and SubSpace.detex put it in their records)
BINDING = {}


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs local_cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11' (the kernel's notation)."""
    out, cpus = [], sorted(set(int(c) for c in cpus))
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else '%d-%d' % (cpus[i], cpus[j]))
        i = j + 1
    return ','.join(out)


def device_locality(pci_bus_id, sysfs='/sys/bus/pci/devices'):
    """NUMA node and local CPUs of the PCI device `pci_bus_id` ('0000:c1:00.0') as the kernel reports them
    (numa_node, local_cpulist); (None, []) where sysfs does not say (containers, single-node hosts: numa_node -1)."""
    import os
    base = os.path.join(sysfs, pci_bus_id.strip().lower())
    node, cpus = None, []
    try:
        with open(os.path.join(base, 'numa_node')) as f:
            v = int(f.read().strip())
            node = v if v >= 0 else None
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(base, 'local_cpulist')) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        pass
    return node, cpus


def bind_to_device(device, sysfs='/sys/bus/pci/devices', pci_bus_id=None):
    """Pin this process to the CPU cores local to its GPU, BEFORE it allocates page-locked buffers: the host feed
    (104 MB per station-day) is the one resource the ranks of a node share, and a rank whose pinned
    record lives on the other socket pays the inter-socket link for every upload.  The kernel's first-touch policy then
    places what the process allocates (hipHostMalloc included) on that NUMA node.  One process per GPU; a no-op
    (recorded as such) where sysfs gives no locality, where the local CPUs are not in the process's affinity mask, or
    under DETEX_AMD_BIND=0.  Returns / records BINDING = {device, pci, numa_node, cpus, applied, reason}."""
    import os
    rec = {'device': int(device), 'pci': None, 'numa_node': None, 'cpus': None, 'applied': False, 'reason': None}
    try:
        if os.environ.get('DETEX_AMD_BIND', '1') == '0':
            rec['reason'] = 'DETEX_AMD_BIND=0'
            return rec
        if pci_bus_id is None:
            import ctypes as C
            from . import _lib
            buf = C.create_string_buffer(64)
            if _lib.load().dtx_device_pci_bus_id(int(device), buf, 64) != 0:
                rec['reason'] = 'no PCI address for device %d' % int(device)
                return rec
            pci_bus_id = buf.value.decode()
        rec['pci'] = pci_bus_id
        node, cpus = device_locality(pci_bus_id, sysfs)
        rec['numa_node'] = node
        if not cpus:
            rec['reason'] = 'sysfs reports no local CPUs'
            return rec
        if not hasattr(os, 'sched_setaffinity'):
            rec['reason'] = 'no sched_setaffinity on this platform'
            return rec
        allowed = os.sched_getaffinity(0)
        use = sorted(set(cpus) & set(allowed))
        if not use:
            rec['reason'] = 'local CPUs outside the affinity mask'
            return rec
        if len(use) < len(allowed):
            try:
                os.sched_setaffinity(0, use)
            except OSError as e:                         # (a sandbox that forbids it: the run goes on unbound)
                rec['reason'] = 'sched_setaffinity refused: %s' % e
                return rec
            rec['applied'] = True
        else:
            rec['reason'] = 'every allowed CPU is local'
        rec['cpus'] = len(use)
        rec['cpulist'] = format_cpulist(use)
        return rec
    finally:
        BINDING.clear()
        BINDING.update(rec)


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


def deal_station_chunks(nchunks, per_unit, rank, world):
    """Chunk indices this rank scans, per station: nchunks[s] chunks on station s, cut into
    (station, day) units of `per_unit` consecutive chunks (a day of hourly chunks goes through
    the kernels as one launch).  The units of ALL stations form one list that is dealt
    round-robin (shard_units) -- what bench.py --config 4 does: with 30 days on 8 ranks every
    rank gets 30 of the 240 station-days, where a per-station deal would hand ranks 0..5 the
    extra day of every station.  With fewer units than ranks the chunks of all stations are cut
    into `world` contiguous ranges instead, so that a single day still spreads over the GPUs.
    Returns one list of chunk indices per station (empty: nothing of that station here)."""
    per_unit = max(1, int(per_unit))
    units = [(s, u) for s, nc in enumerate(nchunks) for u in range((nc + per_unit - 1) // per_unit)]
    out = [[] for _ in nchunks]
    if len(units) >= world:
        for i in shard_units(len(units), rank, world):
            s, u = units[i]
            out[s].extend(range(u * per_unit, min((u + 1) * per_unit, nchunks[s])))
        return out
    flat = [(s, c) for s, nc in enumerate(nchunks) for c in range(nc)]
    b, e = shard_pairs(len(flat), rank, world)
    for s, c in flat[b:e]:
        out[s].append(c)
    return out


def plan_shard(bounds, sel):
    """The sample ranges a rank needs of one station, packed: bounds = the station's chunks
    [(first sample, samples, start time), ...], sel = the chunk indices scanned here.
    Overlapping / abutting chunks (consecutive hours share the 120 s buffer) merge into runs;
    the runs are laid end to end in a compact buffer.  Returns (runs, local, T):
    runs = [(first sample, end sample, offset in the compact buffer)], local = [(first sample in
    the compact buffer, samples, start time, chunk index)] for the selected chunks, T = compact
    length.  Only these T samples per channel are uploaded -- the host link is the one resource
    the ranks share."""
    sel = sorted(int(i) for i in sel if 0 <= int(i) < len(bounds))
    runs = []
    for i in sel:
        a, b = int(bounds[i][0]), int(bounds[i][0]) + int(bounds[i][1])
        if runs and a <= runs[-1][1]:
            runs[-1][1] = max(runs[-1][1], b)
        else:
            runs.append([a, b])
    off, packed = 0, []
    for a, b in runs:
        packed.append((a, b, off))
        off += b - a
    local = []
    for i in sel:
        a = int(bounds[i][0])
        r = next(r for r in packed if r[0] <= a < r[1])
        local.append((r[2] + a - r[0], int(bounds[i][1]), bounds[i][2], i))
    return packed, local, off


def rebase_fragments(frags, runs):
    """Fragments (channel, first sample, samples) of a station clipped to the runs of
    plan_shard and moved to compact-buffer coordinates."""
    from .getdata import clip_fragments
    return clip_fragments(frags, runs)


def gather_rows(rows):
    """All ranks contribute a picklable object (their detection rows, histograms); every rank
    gets the list over ranks.  Host objects, a few KB: not on the data path."""
    import torch.distributed as dist
    rank, world = dist_info()
    if not dist_active():
        return [rows]
    out = [None] * world
    COLLECTIVES['all_gather_object'] += 1
    dist.all_gather_object(out, rows)
    return out


def broadcast_banks(banks, device, src=0):
    """The detector bases of ALL stations from rank `src` in one go: banks = [(U [Ktot, n]
    float array or None, det_off, thresholds float array), ...] per station (None entries on
    the other ranks are fine: only `src`'s are read).  One object broadcast for the shapes,
    offsets and thresholds (host, a few hundred bytes), one tensor broadcast for every basis
    concatenated (7.2 MB per station at configs[1]; RCCL over xGMI under backend 'nccl').
    Returns [(U float32 ndarray, det_off int32, thresholds float64)] on every rank."""
    import torch.distributed as dist
    rank, world = dist_info()
    if not dist_active():
        return [(None if U is None else np.asarray(U, dtype=np.float32),
                 None if U is None else np.asarray(off, np.int32), thr) for (U, off, thr) in banks]
    meta = [None]
    if rank == src:
        # (a string instead of the list: `src` could not build its banks -- every rank raises together)
        meta[0] = banks if isinstance(banks, str) else [
            None if U is None else (tuple(np.shape(U)), [int(v) for v in off], [float(v) for v in thr])
            for (U, off, thr) in banks]
    COLLECTIVES['broadcast_object'] += 1
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    if isinstance(meta, str):
        raise RuntimeError('rank %d could not build the detector banks: %s' % (src, meta))
    total = sum(m[0][0] * m[0][1] for m in meta if m is not None)
    if rank == src:
        flat = np.concatenate([np.asarray(U, dtype=np.float32).ravel() for (U, _, _) in banks
                               if U is not None]) if total else np.zeros(0, np.float32)
        buf = torch.from_numpy(flat).to(device)
    else:
        buf = torch.empty(total, dtype=torch.float32, device=device)
    if total:
        COLLECTIVES['broadcast'] += 1
        dist.broadcast(buf, src=src)
    host = buf.cpu().numpy()
    out, pos = [], 0
    for m in meta:
        if m is None:
            out.append((None, None, None))
            continue
        (K, n), off, thr = m
        out.append((host[pos:pos + K * n].reshape(K, n), np.asarray(off, np.int32),
                    np.asarray(thr, np.float64)))
        pos += K * n
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
    COLLECTIVES['broadcast'] += 3
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
    if not dist_active():
        return parts
    width = max(e - b for b, e in ranges)
    buf = torch.zeros(width, dtype=parts.dtype, device=device)
    b, e = ranges[rank]
    buf[:e - b] = parts[b:e]
    out = [torch.zeros_like(buf) for _ in range(world)]
    COLLECTIVES['all_gather'] += 1
    dist.all_gather(out, buf)
    full = torch.zeros(npairs, dtype=parts.dtype, device=device)
    for r, (b, e) in enumerate(ranges):
        full[b:e] = out[r][:e - b]
    return full

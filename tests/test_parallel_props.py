"""Property tests (hypothesis) of the host-side sharding arithmetic: detex_amd.parallel and SubSpace._scan_units.
CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

from detex_amd import parallel


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 80), min_size=1, max_size=9), st.integers(1, 30), st.integers(1, 9))
def test_deal_station_chunks_partitions_every_station(nchunks, per_unit, world):
    """Over the ranks, every chunk of every station is dealt exactly once; whole (station, day) units stay together
    whenever there are at least as many units as ranks; the ranks' unit counts differ by at most one."""
    dealt = [parallel.deal_station_chunks(nchunks, per_unit, r, world) for r in range(world)]
    for s, nc in enumerate(nchunks):
        got = sorted(c for r in range(world) for c in dealt[r][s])
        assert got == list(range(nc))
    nunits = sum((nc + per_unit - 1) // per_unit for nc in nchunks)
    if nunits >= world:
        counts = []
        for r in range(world):
            units = set()
            for s, cs in enumerate(dealt[r]):
                for c in cs:
                    units.add((s, c // per_unit))
            # a unit is never split between ranks
            for (s, u) in units:
                want = list(range(u * per_unit, min((u + 1) * per_unit, nchunks[s])))
                assert [c for c in dealt[r][s] if c // per_unit == u] == want
            counts.append(len(units))
        assert max(counts) - min(counts) <= 1


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 40), st.integers(10, 400), st.integers(0, 120), st.data())
def test_plan_shard_packs_exactly_the_selected_samples(nchunk, dur, buff, data):
    """The compact buffer of a rank holds, for every selected chunk, exactly that chunk's samples at the position
    plan_shard reports; overlapping / abutting chunks share their samples; nothing unselected is uploaded twice."""
    bounds = [(i * dur, dur + buff, float(i)) for i in range(nchunk)]
    sel = data.draw(st.lists(st.integers(0, nchunk - 1), unique=True, max_size=nchunk))
    runs, local, T = parallel.plan_shard(bounds, sel)
    src = np.arange(nchunk * dur + buff, dtype=np.int64)
    buf = np.full(T, -1, dtype=np.int64)
    covered = 0
    for (a, b, off) in runs:
        buf[off:off + b - a] = src[a:b]
        covered += b - a
    assert covered == T and (buf >= 0).all()
    assert sorted(l[3] for l in local) == sorted(sel)
    for (pos, cnt, ts, idx) in local:
        s0, c0, _ = bounds[idx]
        assert cnt == c0 and np.array_equal(buf[pos:pos + cnt], src[s0:s0 + c0])
    # runs are disjoint, ascending, and no longer than the union of the selected chunks
    union = np.zeros(len(src), bool)
    for i in sel:
        union[bounds[i][0]:bounds[i][0] + bounds[i][1]] = True
    assert T == int(union.sum())


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 16))
def test_shard_pairs_tiles(npairs, world):
    r = [parallel.shard_pairs(npairs, k, world) for k in range(world)]
    assert r[0][0] == 0 and r[-1][1] == npairs
    assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
    assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


@settings(max_examples=100, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 3), st.integers(0, 900), st.integers(1, 120)), min_size=1, max_size=12), st.data())
def test_rebase_fragments_keeps_every_covered_sample(frs, data):
    """Fragments clipped to a rank's runs and moved to compact coordinates hold exactly the samples the runs cover."""
    frags = [(c, t0, np.arange(t0, t0 + m, dtype=np.int32) + 10000 * c) for (c, t0, m) in frs]
    nchunk, dur = 10, 100
    bounds = [(i * dur, dur + 20, float(i)) for i in range(nchunk)]
    sel = data.draw(st.lists(st.integers(0, nchunk - 1), unique=True, min_size=1, max_size=nchunk))
    runs, _, T = parallel.plan_shard(bounds, sel)
    out = parallel.rebase_fragments(frags, runs)
    want = {}
    for (c, t0, x) in frags:
        for (a, b, off) in runs:
            for t in range(max(t0, a), min(t0 + len(x), b)):
                want.setdefault((c, off + t - a), []).append(int(x[t - t0]))
    got = {}
    for (c, p, x) in out:
        assert 0 <= p and p + len(x) <= T
        for j, v in enumerate(x):
            got.setdefault((c, p + j), []).append(int(v))
    assert {k: sorted(v) for k, v in got.items()} == {k: sorted(v) for k, v in want.items()}


def test_bind_to_device_reads_sysfs(tmp_path, monkeypatch):
    """parallel.bind_to_device: the GPU's PCI address -> numa_node / local_cpulist under sysfs -> the process's
    affinity mask (before any page-locked allocation); every way it can find nothing is recorded, not raised."""
    import os
    from detex_amd import parallel
    allowed = sorted(os.sched_getaffinity(0))
    dev = tmp_path / '0000:c1:00.0'
    dev.mkdir()
    (dev / 'numa_node').write_text('1\n')
    assert parallel.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert parallel.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == '0-3,8,10-11' and parallel.format_cpulist([]) == ''
    assert parallel.device_locality('0000:C1:00.0', str(tmp_path)) == (1, [])
    try:
        # no local_cpulist: nothing applied
        rec = parallel.bind_to_device(3, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')
        assert rec['applied'] is False and rec['numa_node'] == 1 and 'no local CPUs' in rec['reason']
        assert parallel.BINDING == rec and sorted(os.sched_getaffinity(0)) == allowed
        # local CPUs outside the mask
        (dev / 'local_cpulist').write_text('%d-%d\n' % (max(allowed) + 1, max(allowed) + 4))
        rec = parallel.bind_to_device(3, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')
        assert rec['applied'] is False and 'outside' in rec['reason']
        # the switch
        monkeypatch.setenv('DETEX_AMD_BIND', '0')
        assert parallel.bind_to_device(3, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')['reason'] == 'DETEX_AMD_BIND=0'
        monkeypatch.delenv('DETEX_AMD_BIND')
        # every allowed CPU local: nothing to do
        (dev / 'local_cpulist').write_text('0-%d\n' % (max(allowed) + 8))
        rec = parallel.bind_to_device(3, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')
        assert rec['applied'] is False and rec['cpus'] == len(allowed)
        if len(allowed) >= 2:
            # a proper subset: the mask shrinks to it
            half = allowed[:len(allowed) // 2]
            (dev / 'local_cpulist').write_text(','.join(map(str, half)) + '\n')
            rec = parallel.bind_to_device(3, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')
            assert rec['applied'] is True and rec['cpus'] == len(half) and rec['device'] == 3 and rec['pci'] == '0000:c1:00.0'
            assert sorted(os.sched_getaffinity(0)) == half
            # a sandbox that refuses the call: recorded, the run goes on unbound
            os.sched_setaffinity(0, allowed)

            def refuse(pid, cpus):
                raise PermissionError(1, 'Operation not permitted')
            monkeypatch.setattr(os, 'sched_setaffinity', refuse)
            rec = parallel.bind_to_device(3, sysfs=str(tmp_path), pci_bus_id='0000:c1:00.0')
            assert rec['applied'] is False and 'refused' in rec['reason'] and parallel.BINDING == rec
            monkeypatch.undo()
    finally:
        os.sched_setaffinity(0, allowed)
        parallel.BINDING.clear()

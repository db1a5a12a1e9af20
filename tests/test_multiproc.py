"""N > 1 control flow on CPU: gloo backend, world size 2 (the GPU path uses the
same code with backend 'nccl' = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from detex_amd import parallel
    assert parallel.dist_info() == (rank, world)
    # basis broadcast: only rank 0 owns the data
    rs = np.random.RandomState(0)
    if rank == 0:
        U = rs.standard_normal((11, 40)).astype(np.float32)
        off = np.array([0, 3, 4, 11], dtype=np.int32)
    else:
        U, off = None, None
    Ut, offb = parallel.broadcast_basis(U, off, 40, torch.device('cpu'))
    ref = np.random.RandomState(0).standard_normal((11, 40)).astype(np.float32)
    ok = np.array_equal(Ut.numpy(), ref) and list(offb) == [0, 3, 4, 11]
    # pair sharding + gather
    npairs = 37
    ranges = [parallel.shard_pairs(npairs, r, world) for r in range(world)]
    full_ref = torch.arange(npairs, dtype=torch.float32) * 0.5
    mine = torch.zeros(npairs)
    b, e = ranges[rank]
    mine[b:e] = full_ref[b:e]
    full = parallel.gather_condensed(mine, ranges, npairs, torch.device('cpu'))
    ok = ok and torch.equal(full, full_ref)
    # unit sharding + max-over-ranks timing reduction as in bench.py
    units = parallel.shard_units(7, rank, world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t) == world
    q.put((rank, ok, units, ranges))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] for r in res)
    assert sorted(res[0][2] + res[1][2]) == list(range(7))
    assert res[0][3] == [(0, 19), (19, 37)]


def test_shard_pairs_tiles_range():
    from detex_amd import parallel
    for npairs in [0, 1, 7, 1999000]:
        for world in [1, 2, 3, 8]:
            r = [parallel.shard_pairs(npairs, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == npairs
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def _product_worker(rank, world, port, q):
    """createCluster (pair ranges + all-gather) and SubSpace.detex (basis broadcast, chunk
    sharding, gathered results) as the product runs them with one process per GPU, on CPU:
    gloo instead of RCCL, the oracle stand-in instead of the HIP context."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from stub_ctx import OracleCtx
    from test_host_logic import small_workflow
    ctx = OracleCtx()
    ss, cf, plants, _ = small_workflow(ctx)
    xc = [c for c in ctx.calls if c[0] == 'xcorr']
    res = ss.detex(subspaceDB=None, useSingles=True, ctx=ctx)
    scans = [c for c in ctx.calls if c[0] == 'scan']
    row = ss.clusters.trdf.iloc[0]
    q.put((rank, xc, scans, res['ss_df'].values.tolist(), res['sg_df'].values.tolist(),
           {k: v.tolist() for k, v in ss.histSubSpaces['TA.M17A'].items()},
           row.CCs.values.astype(float).tolist(), row.Lags.values.astype(float).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_product_entry_points_world2():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from detex_amd import parallel
    from stub_ctx import OracleCtx
    from test_host_logic import small_workflow
    # single process: the expected answer
    c1 = OracleCtx()
    ss1, cf, plants, _ = small_workflow(c1)
    ref = ss1.detex(subspaceDB=None, useSingles=True, ctx=c1)
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_product_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    npairs = 9 * 8 // 2
    for rank, xc, scans, ssrows, sgrows, hist, ccs, lags in res:
        # every rank correlated its half of the pair list ...
        assert xc == [('xcorr',) + tuple(parallel.shard_pairs(npairs, rank, world))]
        # ... and ends with the full matrices and the full detection tables
        r1 = ss1.clusters.trdf.iloc[0]
        assert np.allclose(np.array(ccs), r1.CCs.values.astype(float), equal_nan=True)
        assert np.allclose(np.array(lags), r1.Lags.values.astype(float), equal_nan=True)
        got = np.array([[v for v in r if not isinstance(v, str)] for r in ssrows], dtype=float)
        want = ref['ss_df'].sort_values(['STMP', 'Name'], kind='stable')
        want = want[[c for c in want.columns if c not in ('Name', 'Sta')]].values.astype(float)
        assert got.shape == want.shape and np.allclose(got, want, equal_nan=True)
        assert len(sgrows) == len(ref['sg_df'])
        for nm, h in ss1.histSubSpaces['TA.M17A'].items():
            assert hist[nm] == h.tolist()
    # the four chunks were split two and two: each rank scanned only its share
    # (rank 1's two chunks differ in length: one ragged launch)
    assert [c[1] for c in res[0][2]] == [2] and [c[1] for c in res[1][2]] == [2]


def _expected_upload(cf, stas, rank, world):
    """Bytes of station data rank `rank` has to upload: its own chunks' sample ranges, merged."""
    from detex_amd import parallel
    bounds = [cf.chunk_bounds(s) for s in stas]
    per = max(1, int(round(86400.0 / cf.conDatDuration)))
    dealt = parallel.deal_station_chunks([len(b) for b in bounds], per, rank, world)
    return sum(4 * len(cf.channels) * parallel.plan_shard(b, d)[2] for b, d in zip(bounds, dealt)), dealt


def _configs3_worker(rank, world, port, q):
    """SubSpace.detex over three stations x two days with one process per 'GPU' (gloo + the
    oracle stand-in): what each rank uploads, how many collectives the call costs, the tables."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from detex_amd import parallel
    from stub_ctx import OracleCtx
    from test_host_logic import multi_station_workflow
    ctx = OracleCtx()
    ss, cf, stas, _ = multi_station_workflow(ctx)
    before = dict(parallel.COLLECTIVES)
    res = ss.detex(subspaceDB=None, ctx=ctx)
    coll = {k: parallel.COLLECTIVES[k] - before[k] for k in before}
    scans = [c[1] for c in ctx.calls if c[0] == 'scan']
    hist = {sta: {k: v.tolist() for k, v in ss.histSubSpaces[sta].items()} for sta in stas}
    last = dict(ss.last_run)
    # a failure on ONE rank must not leave the others hung in the gather: both raise
    class Boom(OracleCtx):
        def subspace_scan(self, *a, **kw):
            if rank == 1:
                raise RuntimeError('injected fault on rank 1')
            return OracleCtx.subspace_scan(self, *a, **kw)
    try:
        ss.detex(subspaceDB=None, ctx=Boom())
        failed = None
    except Exception as e:
        failed = str(e)
    # ... and neither must a failure BEFORE the basis broadcast: a rank that cannot build its
    # banks still takes part in every collective of the call; whichever rank it is, all of them raise
    early = []
    for bad_rank in (1, 0):
        orig = ss._detector_bank
        if rank == bad_rank:
            def boom(*a, **kw):
                raise ValueError('no bank on rank %d' % rank)
            ss._detector_bank = boom
        try:
            ss.detex(subspaceDB=None, ctx=ctx)
            early.append(None)
        except Exception as e:
            early.append(str(e))
        ss.__dict__.pop('_detector_bank', None)
        assert ss._detector_bank == orig
    # the group is still usable afterwards: same tables as before
    again = ss.detex(subspaceDB=None, ctx=ctx)
    assert again['ss_df'].values.tolist() == res['ss_df'].values.tolist()
    # ... also with every rank's chunks scanned in day units (SubSpace._scan_units): the cut does not show
    os.environ['DETEX_AMD_DS_BUDGET_MB'] = '0'
    nscan = len([c for c in ctx.calls if c[0] == 'scan'])
    units = ss.detex(subspaceDB=None, ctx=ctx)
    os.environ.pop('DETEX_AMD_DS_BUDGET_MB')
    assert units['ss_df'].values.tolist() == res['ss_df'].values.tolist()
    assert len([c for c in ctx.calls if c[0] == 'scan']) - nscan > len(stas)      # more launches than stations
    # utcSaves under a process group: the rows of the chunks the ranks scanned come together in chunk order
    t_in = [cf.chunk_bounds(stas[0])[1][2] + 100.0, cf.chunk_bounds(stas[0])[3][2] + 100.0]
    os.chdir(os.environ.get('DTX_TEST_TMP', '.'))
    usv = ss.detex(subspaceDB=None, ctx=ctx, utcSaves=t_in)['utc_saves']
    utc = [(r.Station, r.Name, float(r.TS1), float(np.max(r.SSdetect))) for _, r in usv.iterrows()]
    q.put((rank, coll, last, scans, res['ss_df'].values.tolist(), hist, failed, early, utc))
    dist.barrier()
    dist.destroy_process_group()


def test_configs3_sharding_world2():
    """(station, day) units dealt globally, only the rank's own sample ranges uploaded, one
    basis broadcast + one gather per call whatever the number of stations, rows and histograms
    identical to the single-process run (row order included), a failing rank fails every rank."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from stub_ctx import OracleCtx
    from test_host_logic import multi_station_workflow
    c1 = OracleCtx()
    ss1, cf, stas, _ = multi_station_workflow(c1)
    ref = ss1.detex(subspaceDB=None, ctx=c1)
    assert len(ref['ss_df']) >= 10
    total = sum(4 * 3 * cf.data[s][0].shape[1] for s in stas)
    assert ss1.last_run['uploaded_bytes'] == total and ss1.last_run['world'] == 1
    world = 2
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    port = _free_port()
    os.environ['DTX_TEST_TMP'] = os.getcwd()             # (UTCsaves.pkl of the workers goes to the test's scratch directory)
    procs = [mpc.Process(target=_configs3_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    up = []
    want_utc = None
    for rank, coll, last, scans, rows, hist, failed, early, utc in res:
        # three collectives per call: shapes / thresholds (object), the bases (tensor), the rows
        assert coll == {'broadcast': 1, 'broadcast_object': 1, 'all_gather_object': 1, 'all_gather': 0}, coll
        want_bytes, dealt = _expected_upload(cf, stas, rank, world)
        assert last['uploaded_bytes'] == want_bytes and last['rank'] == rank and last['world'] == world
        # five chunks per station = two days + a 120 s tail = three units; the nine units of the three
        # stations are dealt round-robin as ONE list (a per-station deal would give rank 0 the extra unit thrice)
        assert dealt == ([[0, 1, 4], [2, 3], [0, 1, 4]] if rank == 0 else [[2, 3], [0, 1, 4], [2, 3]])
        assert last['chunks'] == {s: len(d) for s, d in zip(stas, dealt)}
        up.append(last['uploaded_bytes'])
        # identical tables on every rank, in the single-process row order
        assert len(rows) == len(ref['ss_df'])
        for got, (_, want) in zip(rows, ref['ss_df'].iterrows()):
            assert got[3] == want.Name and got[4] == want.Sta and got[2] == want.STMP
            assert np.allclose([got[0], got[1]] + got[5:], [want.DS, want.DS_STALTA] + list(want.values[5:]),
                               rtol=1e-6, equal_nan=True)
        for sta in stas:
            for nm, h in ss1.histSubSpaces[sta].items():
                assert hist[sta][nm] == h.tolist()
        assert failed is not None and 'rank' in failed and 'injected fault' in failed
        assert len(early) == 2 and all(e is not None and 'no bank on rank' in e for e in early), early
        # utcSaves rows: the same on every rank as in a single process (chunks 1 and 3 of every station cover a time)
        if want_utc is None:
            t_in = [cf.chunk_bounds(stas[0])[1][2] + 100.0, cf.chunk_bounds(stas[0])[3][2] + 100.0]
            if os.path.exists('UTCsaves.pkl'):           # (rank 0's file: a new run appends what it finds, detect.py:101-105)
                os.remove('UTCsaves.pkl')
            u1 = ss1.detex(subspaceDB=None, ctx=c1, utcSaves=t_in)['utc_saves']
            want_utc = [(r.Station, r.Name, float(r.TS1), float(np.max(r.SSdetect))) for _, r in u1.iterrows()]
            assert len(want_utc) >= 2 * len(stas)
        assert len(utc) == len(want_utc)
        for g, w in zip(utc, want_utc):
            assert g[:3] == w[:3] and abs(g[3] - w[3]) < 1e-6      # (every rank scans with the float32 basis that was broadcast)
    # each rank moved about half of the data over the host link (plus the 120 s chunk overlap)
    assert max(up) < 0.52 * total and sum(up) < 1.01 * total


def _nccl_worker(rank, world, port, q, backend='nccl'):
    """The same three-station run with one process per GPU under RCCL (backend 'nccl'), the real HIP
    contexts and a pinned fetcher: exercises broadcast_banks / all_gather_object over xGMI.
    backend='gloo': both processes share GPU 0 (a one-GPU box): the same product code with device tensors
    in the collectives, real contexts, copy streams and packed uploads -- everything but RCCL itself."""
    dev = rank if backend == 'nccl' else 0
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(dev),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    rank_dev = dev
    from detex_amd import parallel
    from detex_amd.hip import HipContext
    from test_host_logic import multi_station_workflow
    ctx = HipContext(rank_dev)
    ss, cf, stas, _ = multi_station_workflow(ctx, pin=True)
    ctx.close()
    before = dict(parallel.COLLECTIVES)
    res = ss.detex(subspaceDB=None)                      # device defaults to LOCAL_RANK
    coll = {k: parallel.COLLECTIVES[k] - before[k] for k in before}
    q.put((rank, coll, dict(ss.last_run), res['ss_df'].values.tolist(),
           {sta: {k: v.tolist() for k, v in ss.histSubSpaces[sta].items()} for sta in stas}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_configs3_sharding_world2_nccl():
    """Skips itself unless the box has two GPUs: the first multi-GPU node this suite meets runs
    SubSpace.detex with two RCCL ranks against the single-GPU answer."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL world of 2)')
    _check_gpu_world2('nccl')


def _check_gpu_world2(backend):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from detex_amd.hip import HipContext
    from test_host_logic import multi_station_workflow
    c1 = HipContext(0)
    ss1, cf, stas, _ = multi_station_workflow(c1, pin=True)
    c1.close()
    ref = ss1.detex(subspaceDB=None, device=0)
    world = 2
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_nccl_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total = sum(4 * 3 * cf.data[s][0].shape[1] for s in stas)
    for rank, coll, last, rows, hist in res:
        assert coll == {'broadcast': 1, 'broadcast_object': 1, 'all_gather_object': 1, 'all_gather': 0}, coll
        assert last['uploaded_bytes'] == _expected_upload(cf, stas, rank, world)[0] < 0.52 * total
        assert len(rows) == len(ref['ss_df']) >= 10
        for got, (_, want) in zip(rows, ref['ss_df'].iterrows()):
            assert got[3] == want.Name and got[4] == want.Sta and got[2] == want.STMP
            assert np.allclose([got[0], got[1]] + got[5:], [want.DS, want.DS_STALTA] + list(want.values[5:]),
                               rtol=1e-5, equal_nan=True)
        for sta in stas:
            for nm, h in ss1.histSubSpaces[sta].items():
                assert hist[sta][nm] == h.tolist()


@pytest.mark.gpu
def test_configs3_sharding_world2_gpu_gloo():
    """Two processes sharing this box's one GPU (gloo carries the collectives): SubSpace.detex with real
    HIP contexts, pinned fetchers, packed per-rank uploads and device tensors in the broadcast -- against
    the single-process answer, row for row."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    _check_gpu_world2('gloo')


def test_device_defaults_to_local_rank():
    """SubSpace.detex and createCluster take the GPU from LOCAL_RANK when none is named
    (one process per GPU); a literal default of 0 would put every rank on cuda:0."""
    import inspect
    from detex_amd import construct, parallel, subspace
    assert inspect.signature(subspace.SubSpace.detex).parameters['device'].default is None
    assert inspect.signature(construct.createCluster).parameters['device'].default is None
    for fn in (subspace.SubSpace.getFAS, subspace.SubSpace.validateClusters, construct.fast_normcorr):
        assert inspect.signature(fn).parameters['device'].default is None, fn
    assert parallel.local_device() == 0 and not parallel.dist_active()        # no process group here

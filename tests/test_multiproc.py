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
    assert [c[1] for c in res[0][2]] == [2] and [c[1] for c in res[1][2]] == [1, 1]

"""Worker of tests/test_gpu_rccl.py, started through ``python -m torch.distributed.run --nproc-per-node 1``:
the product's entry points with every collective carried by RCCL (backend 'nccl') at world size 1.

The answer without a process group is computed first (dist inactive: the plain single-process path), then the
group is initialised exactly as a one-process-per-GPU launch does and the same calls are made again: same tables,
and the collectives counted (parallel.COLLECTIVES) are the ones an 8-GPU run issues.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def tables(res):
    return [[(v if isinstance(v, str) else float(v)) for v in r] for r in res['ss_df'].values.tolist()]


def same_tables(a, b):
    """Row for row: names, stations, times, DS and STA/LTA exactly; the magnitude columns to 1e-6 (under a
    process group every rank scans with the float32 basis that was broadcast, the plain run builds its
    magnitude constants from the float64 one)."""
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if x[:7] != y[:7]:
            return False
        if not np.allclose(x[7:], y[7:], rtol=1e-6, atol=0, equal_nan=True):
            return False
    return True


def main():
    from detex_amd import createCluster, parallel
    from detex_amd.hip import HipContext
    from test_host_logic import multi_station_workflow
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    out = {}
    # ---- without a process group
    assert not parallel.dist_active()
    c0 = HipContext(local)
    ss, cf, stas, _ = multi_station_workflow(c0, pin=True)
    c0.close()
    ref = ss.detex(subspaceDB=None, device=local)
    ref_rows = tables(ref)
    ref_hist = {sta: {k: v.tolist() for k, v in ss.histSubSpaces[sta].items()} for sta in stas}
    cl_ref = ss.clusters
    # ---- RCCL, world size 1
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    assert parallel.dist_active() and parallel.dist_info() == (0, 1)
    before = dict(parallel.COLLECTIVES)
    res = ss.detex(subspaceDB=None)                      # device: LOCAL_RANK
    coll = {k: parallel.COLLECTIVES[k] - before[k] for k in before}
    rows = tables(res)
    assert same_tables(rows, ref_rows), 'SubSpace.detex under RCCL differs from the plain run'
    for sta in stas:
        for nm, h in ss.histSubSpaces[sta].items():
            assert h.tolist() == ref_hist[sta][nm]
    out['detex'] = dict(rows=len(rows), collectives=coll, uploaded_bytes=ss.last_run['uploaded_bytes'],
                        world=ss.last_run['world'])
    # a second call re-uses the resident bank objects (the broadcast numbers are unchanged)
    res2 = ss.detex(subspaceDB=None)
    assert same_tables(tables(res2), ref_rows)
    # ---- createCluster: pair range + all_gather of the condensed vectors (three all_gathers per station)
    before = dict(parallel.COLLECTIVES)
    cl = createCluster(CCreq=0.5, fetch_arg=cl_ref.fetcher, saveclust=False)    # device: LOCAL_RANK
    coll = {k: parallel.COLLECTIVES[k] - before[k] for k in before}
    for (_, a), (_, b) in zip(cl.trdf.iterrows(), cl_ref.trdf.iterrows()):
        assert a.Station == b.Station
        for col in ('CCs', 'Lags', 'Subsamp'):
            assert np.array_equal(np.asarray(a[col].values, float), np.asarray(b[col].values, float), equal_nan=True), col
        assert np.array_equal(a.Link, b.Link)
    out['createCluster'] = dict(stations=len(cl.trdf), collectives=coll)
    # ---- the bench's basis broadcast
    U = np.random.RandomState(3).standard_normal((11, 60)).astype(np.float32)
    Ut, off = parallel.broadcast_basis(U, np.array([0, 3, 11], np.int32), 60, torch.device('cuda', local))
    assert np.array_equal(Ut.cpu().numpy(), U) and list(off) == [0, 3, 11]
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == '__main__':
    main()

"""Every collective path of the product under RCCL (backend 'nccl') on the ONE GPU of the test box: world
size 1, launched through ``python -m torch.distributed.run --nproc-per-node 1`` exactly as the driver launches
N > 1 (north_star: days sharded, one basis broadcast).  What differs from an 8-GPU run is the peer
count only: the rendezvous, ``init_process_group('nccl', device_id=...)``, the object broadcast, the tensor
broadcast, ``all_gather`` / ``all_gather_object``, the barriers and the max-over-ranks reduction all execute."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _torchrun(script_args, timeout=900, env=None):
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    e.pop('DTX_BENCH_BACKEND', None)
    e.pop('DTX_BENCH_ONE_DEVICE', None)
    e.update(env or {})
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                        '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args,
                       capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _plain(script_args, timeout=900):
    r = subprocess.run([sys.executable] + script_args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1])


@pytest.fixture(autouse=True)
def _needs_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


def test_bench_weak_under_rccl_world1():
    """bench.py --gpus 1 under torch.distributed.run + RCCL: same candidates as the plain launch."""
    bench = os.path.join(ROOT, 'bench.py')
    args = ['--gpus', '1', '--steps', '3', '--warmup', '1', '--chunks', '6', '--no-extras', '--no-cpu-baseline',
            '--prewarm', '0']
    # (DTX_BENCH_HOST_FED=1: the host-fed leg of N > 1 -- every rank's uploads at once, gathered with
    # all_gather_object -- at world size 1, so that its collectives have run under RCCL too)
    d = _torchrun([bench] + args, env={'DTX_BENCH_HOST_FED': '1'})
    p = _plain([bench] + args)
    assert d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['config']['planted_found'] == '50/50'
    assert d['config']['candidates'] == p['config']['candidates']
    hf = d['host_fed']
    assert len(hf['per_rank']) == 1 and hf['per_rank'][0]['ms_per_step'] > 0 and 'pci' in hf['per_rank'][0]['binding']
    assert hf['ms_per_step_min'] == hf['ms_per_step_max'] == hf['per_rank'][0]['ms_per_step']
    P = (int(round(3720.0 * 100.0)) * 3 - 9000) // 3 + 1
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 - 6 * P) <= 1e-6 * 6 * P


def test_bench_strong_under_rccl_world1():
    """bench.py --config 4 --stations 2 --days 3 under RCCL: per-station bases through three RCCL broadcasts
    each, all-reduced timing and plant counts; same candidates as the plain launch."""
    bench = os.path.join(ROOT, 'bench.py')
    args = ['--config', '4', '--stations', '2', '--days', '3', '--chunks', '4', '--gpus', '1', '--steps', '2',
            '--warmup', '1']
    d = _torchrun([bench] + args, env={'DTX_BENCH_HOST_FED': '1'})       # (+ the host-fed leg of N > 1, as above)
    p = _plain([bench] + args)
    assert d['n_gpus'] == 1 and d['scaling'] == 'strong' and d['config']['planted_found'] == '30/30'
    assert d['config']['candidates'] == p['config']['candidates']
    assert d['host_fed']['per_rank'][0]['ms_per_step'] > 0 and d['host_fed']['ms_per_station_day_slowest_rank'] > 0


def test_product_api_under_rccl_world1():
    """SubSpace.detex (object broadcast + basis broadcast + all_gather_object) and createCluster (pair range +
    all_gather) under RCCL at world size 1: tables identical to the run without a process group (asserted in
    the worker), the collectives of a call counted."""
    d = _torchrun([os.path.join(ROOT, 'tests', 'rccl_world1_worker.py')])
    assert d['detex']['collectives'] == {'broadcast': 1, 'broadcast_object': 1, 'all_gather_object': 1,
                                         'all_gather': 0}
    assert d['detex']['rows'] >= 10 and d['detex']['world'] == 1
    # three condensed vectors (cc, lag, subsamp) gathered per station
    assert d['createCluster']['collectives']['all_gather'] == 3 * d['createCluster']['stations']


def test_dist_overhead_field():
    """The default bench line's dist_overhead leg: the timed loop again with an in-process RCCL group."""
    bench = os.path.join(ROOT, 'bench.py')
    sys.path.insert(0, ROOT)
    r = subprocess.run([sys.executable, '-c',
                        'import sys, json; sys.argv = ["bench.py"]; sys.path.insert(0, %r)\n'
                        'import numpy as np, torch, bench\n'
                        'from detex_amd.hip import HipContext\n'
                        'from detex_amd import synth\n'
                        'ctx = HipContext(0)\n'
                        'U, off = synth.detector_bank(1, 4, 2, 900)\n'
                        'ctx.upload_basis(U, off, 3)\n'
                        'x = torch.randn(2, 30000, device="cuda")\n'
                        'thr = torch.full((4,), 0.5, device="cuda")\n'
                        'def loop(steps, barrier):\n'
                        '    import time\n'
                        '    barrier(); t0 = time.perf_counter()\n'
                        '    for _ in range(steps): r = ctx.subspace_scan(x, thr=thr, lazy=True)[1]()\n'
                        '    barrier(); return time.perf_counter() - t0, r\n'
                        'dt, _ = loop(5, torch.cuda.synchronize)\n'
                        'print(json.dumps(bench.dist_overhead(0, loop, 5, 1e3 * dt / 5, np.asarray(U, np.float32), off, 900)))\n'
                        % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1])
    assert d['world_size'] == 1 and d['backend'].startswith('nccl') and d['ms_per_step_dist'] > 0

"""All-cores leg of bench.py's ``cpu_baseline`` -- TEST / MEASUREMENT INFRASTRUCTURE, not
product code (only bench.py's cpu_baseline leg imports this).

The reference scans single-threaded (detex/subspace.py:1843-1845 rejects ``multiprocess``);
SURVEY.md section 8(d) asks for the same CPU restatement on all physical cores as well: one
worker process per physical core, each running the reference's FFT algorithm
(oracle.detex_oracle.mpxds_fft, detex/detect.py:559-578) on its own chunk for a slice of the
basis vectors.  Workers are plain child processes (``python -m oracle.cpu_bench <dir> <i>``,
numpy / scipy only -- they never touch the GPU and do not re-import the parent's main
module); a file gate starts them together once all have finished importing, and the rate is
the work of all workers over the wall clock from the gate to the last finish.
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_info():
    """(model name, physical cores usable by this process, logical cpus)."""
    model, phys = 'unknown', set()
    try:
        pid = cid = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name') and model == 'unknown':
                    model = line.split(':', 1)[1].strip()
                elif line.startswith('physical id'):
                    pid = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    cid = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    nphys = len(phys) if phys else logical
    return model, max(1, min(nphys, usable)), logical


def cgroup_cpu_limit():
    """CPUs the container may use according to its cgroup CPU controller, or None when unlimited / unreadable:
    cgroup v2 ``cpu.max`` ("<quota> <period>" or "max ..."), cgroup v1 ``cpu.cfs_quota_us / cpu.cfs_period_us``.
    sched_getaffinity does not show such a quota."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            q = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cgroup_throttle():
    """(nr_throttled, throttled microseconds) of the container's cgroup so far, or None."""
    for path in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
        try:
            d = {}
            with open(path) as f:
                for line in f:
                    k, v = line.split()[:2]
                    d[k] = float(v)
            us = d.get('throttled_usec', d.get('throttled_time', 0.0) / 1e3)
            return d.get('nr_throttled', 0.0), us
        except (OSError, ValueError):
            continue
    return None


def effective_cores():
    """Cores a worker pool can really occupy here: physical cores this process may run on, capped by the
    cgroup CPU quota when there is one."""
    _, ncores, _ = cpu_info()
    lim = cgroup_cpu_limit()
    return ncores if lim is None else max(1, min(ncores, int(lim + 0.999)))


def _pool_report(res, wall, workers, onecore_rate, rate, thr0):
    """What shows whether an all-cores leg really ran on `workers` cores: the workers' own CPU seconds over
    the wall clock (time-sliced or throttled workers accumulate less CPU time than wall time; workers that
    stall on a shared memory system accumulate it in full and still do not scale), the cgroup's quota and its
    throttling during the leg, per-worker seconds, and the speed-up over the one-core leg."""
    cpu_s = sum(r.get('cpu_seconds', r['seconds']) for r in res)
    thr1 = cgroup_throttle()
    out = dict(workers=workers, effective_cores=cpu_s / wall if wall > 0 else None,
               cgroup_cpu_limit=cgroup_cpu_limit(),
               worker_seconds=dict(mean=float(np.mean([r['seconds'] for r in res])),
                                   max=float(max(r['seconds'] for r in res)),
                                   min=float(min(r['seconds'] for r in res))),
               worker_cpu_seconds_mean=cpu_s / len(res))
    if thr0 is not None and thr1 is not None:
        out['cgroup_throttled'] = dict(periods=thr1[0] - thr0[0], seconds=(thr1[1] - thr0[1]) / 1e6)
    if onecore_rate:
        out['speedup_vs_one_core'] = rate / onecore_rate
        out['parallel_efficiency'] = rate / onecore_rate / workers
        # the same work on one otherwise idle core would take each worker this long
        out['worker_seconds_if_alone'] = out['worker_seconds']['mean'] * rate / onecore_rate / workers
    return out


def _mem_available_gb():
    try:
        with open('/proc/meminfo') as f:
            for line in f:
                if line.startswith('MemAvailable'):
                    return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 8.0


def _worker(workdir, i):
    """One worker: the reference algorithm on one chunk and a slice of the detectors (its own
    stream FFT, then one 2^21-point IFFT per basis vector)."""
    import scipy.fftpack
    sys.path.insert(0, ROOT)
    from oracle import detex_oracle as orc
    z = np.load(os.path.join(workdir, 'task%d.npz' % i))
    x, U, off, Nc = z['x'].astype(np.float64), z['U'].astype(np.float64), z['off'], int(z['Nc'])
    n = U.shape[1]
    rbi = 2 ** int(len(x) + n).bit_length()
    open(os.path.join(workdir, 'ready%d' % i), 'w').close()
    deadline = time.time() + 600
    while not os.path.exists(os.path.join(workdir, 'go')):      # the start gate
        if time.time() > deadline:
            sys.exit(2)
        time.sleep(0.005)
    t0, c0 = time.perf_counter(), time.process_time()
    xFD = scipy.fftpack.fft(x, n=rbi)                            # detect.py:255-256
    nv = 0
    for d in range(len(off) - 1):
        orc.mpxds_fft(x, U[off[d]:off[d + 1]], Nc, rbi=rbi, xFD=xFD)
        nv += int(off[d + 1] - off[d])
    dt = time.perf_counter() - t0
    with open(os.path.join(workdir, 'done%d.json' % i), 'w') as f:
        json.dump(dict(nvec=nv, seconds=dt, cpu_seconds=time.process_time() - c0), f)


def allcores_baseline(U, det_off, x_chunks, Nc, nvec_per_worker=40, max_workers=None,
                      timeout=600.0, onecore_rate=None):
    """windows/s of the reference algorithm with one worker per physical core, each on one
    chunk x ``nvec_per_worker`` basis vectors; the rate is scaled linearly to all K vectors
    (the cost is one IFFT per vector and chunk)."""
    model, ncores, logical = cpu_info()
    # ~1.2 GB peak per worker (8-vector detector: spectra, products and inverse transforms
    # of 2^21 complex128 points each)
    workers = max(1, min(effective_cores(), int(_mem_available_gb() / 1.5)))
    if max_workers:
        workers = min(workers, int(max_workers))
    thr0 = cgroup_throttle()
    D = len(det_off) - 1
    K, n = U.shape
    x_chunks = np.atleast_2d(x_chunks)
    procs = []
    with tempfile.TemporaryDirectory(prefix='dtx_cpu_') as wd:
        d = 0
        for w in range(workers):
            rows, off = [], [0]
            while off[-1] < nvec_per_worker:
                rows.append(U[det_off[d]:det_off[d + 1]])
                off.append(off[-1] + rows[-1].shape[0])
                d = (d + 1) % D
            np.savez(os.path.join(wd, 'task%d.npz' % w), x=x_chunks[w % len(x_chunks)],
                     U=np.concatenate(rows), off=np.array(off), Nc=Nc)
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        try:
            for w in range(workers):
                procs.append(subprocess.Popen([sys.executable, '-m', 'oracle.cpu_bench', wd, str(w)],
                                              cwd=ROOT, env=env))
            t_end = time.time() + timeout
            while not all(os.path.exists(os.path.join(wd, 'ready%d' % w)) for w in range(workers)):
                if time.time() > t_end or any(p.poll() not in (None, 0) for p in procs):
                    raise RuntimeError('cpu_bench workers failed to start')
                time.sleep(0.01)
            t0 = time.perf_counter()
            open(os.path.join(wd, 'go'), 'w').close()
            for p in procs:
                p.wait(timeout=max(1.0, t_end - time.time()))
            wall = time.perf_counter() - t0
            res = []
            for w in range(workers):
                with open(os.path.join(wd, 'done%d.json' % w)) as f:
                    res.append(json.load(f))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    nvec = sum(r['nvec'] for r in res)
    P = (x_chunks.shape[1] - n) // Nc + 1
    # nvec (vector, chunk) units done in `wall` seconds; a full chunk needs K of them
    value = P * (nvec / float(K)) / wall
    return dict(value=value, unit='windows/s', cores=workers, kind='port', cpu_model=model,
                physical_cores=ncores, os_cpu_count=logical,
                pool=_pool_report(res, wall, workers, onecore_rate, value, thr0),
                sample='%d worker processes (one per physical core) x 1 chunk x %d basis vectors '
                       'each, float64 FFT path, %.1f s wall (slowest worker %.1f s), scaled linearly '
                       'to %d vectors' % (workers, nvec // workers, wall,
                                          max(r['seconds'] for r in res), K))


def xcorr_onecore(td, Nc, npairs=24):
    """pairs/s of the reference's _CCX2 (detex/construct.py:425-466, oracle restatement ccx2_fft,
    float64) on one core: `npairs` pairs drawn from the events td [>= 2][n]; the events' spectra
    are precomputed as in the reference (_getFreqDomain once per event, construct.py:669-676)."""
    sys.path.insert(0, ROOT)
    from oracle import detex_oracle as orc
    ne = min(len(td), 8)
    ev = [np.asarray(td[i], dtype=np.float64) for i in range(ne)]
    fds = [orc.get_freq_domain(e) for e in ev]
    pairs = [(b, c) for b in range(ne) for c in range(b + 1, ne)]
    t0 = time.perf_counter()
    for k in range(npairs):
        b, c = pairs[k % len(pairs)]
        orc.ccx2_fft(ev[b], ev[c], Nc, fds[b], fds[c])
    dt = time.perf_counter() - t0
    model, ncores, logical = cpu_info()
    return dict(value=npairs / dt, unit='pairs/s', cores=1, kind='port', cpu_model=model,
                physical_cores=ncores, os_cpu_count=logical,
                sample='%d pairs of n = %d (float64 FFT path of _CCX2, spectra precomputed), %.1f s'
                       % (npairs, len(ev[0]), dt))


def _xc_worker(workdir, i):
    """One worker of the all-cores cross-correlation leg: `npairs` pairs of its own events."""
    sys.path.insert(0, ROOT)
    from oracle import detex_oracle as orc
    z = np.load(os.path.join(workdir, 'xtask%d.npz' % i))
    ev = [e.astype(np.float64) for e in z['td']]
    Nc, npairs = int(z['Nc']), int(z['npairs'])
    fds = [orc.get_freq_domain(e) for e in ev]
    pairs = [(b, c) for b in range(len(ev)) for c in range(b + 1, len(ev))]
    open(os.path.join(workdir, 'ready%d' % i), 'w').close()
    deadline = time.time() + 600
    while not os.path.exists(os.path.join(workdir, 'go')):
        if time.time() > deadline:
            sys.exit(2)
        time.sleep(0.005)
    t0, c0 = time.perf_counter(), time.process_time()
    for k in range(npairs):
        b, c = pairs[k % len(pairs)]
        orc.ccx2_fft(ev[b], ev[c], Nc, fds[b], fds[c])
    with open(os.path.join(workdir, 'done%d.json' % i), 'w') as f:
        json.dump(dict(npairs=npairs, seconds=time.perf_counter() - t0, cpu_seconds=time.process_time() - c0), f)


def xcorr_allcores(td, Nc, pairs_per_worker=24, max_workers=None, timeout=300.0, onecore_rate=None):
    """pairs/s of the same restatement with one worker process per physical core."""
    model, ncores, logical = cpu_info()
    workers = max(1, min(effective_cores(), int(_mem_available_gb() / 0.25)))
    if max_workers:
        workers = min(workers, int(max_workers))
    thr0 = cgroup_throttle()
    procs = []
    with tempfile.TemporaryDirectory(prefix='dtx_cpux_') as wd:
        for w in range(workers):
            np.savez(os.path.join(wd, 'xtask%d.npz' % w), td=np.asarray(td[(4 * w) % max(1, len(td) - 4):][:4]),
                     Nc=Nc, npairs=pairs_per_worker)
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        try:
            for w in range(workers):
                procs.append(subprocess.Popen([sys.executable, '-m', 'oracle.cpu_bench', wd, str(w), 'xc'],
                                              cwd=ROOT, env=env))
            t_end = time.time() + timeout
            while not all(os.path.exists(os.path.join(wd, 'ready%d' % w)) for w in range(workers)):
                if time.time() > t_end or any(p.poll() not in (None, 0) for p in procs):
                    raise RuntimeError('cpu_bench workers failed to start')
                time.sleep(0.01)
            t0 = time.perf_counter()
            open(os.path.join(wd, 'go'), 'w').close()
            for p in procs:
                p.wait(timeout=max(1.0, t_end - time.time()))
            wall = time.perf_counter() - t0
            res = []
            for w in range(workers):
                with open(os.path.join(wd, 'done%d.json' % w)) as f:
                    res.append(json.load(f))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    total = sum(r['npairs'] for r in res)
    return dict(value=total / wall, unit='pairs/s', cores=workers, kind='port', cpu_model=model,
                physical_cores=ncores, os_cpu_count=logical,
                pool=_pool_report(res, wall, workers, onecore_rate, total / wall, thr0),
                sample='%d worker processes x %d pairs of n = %d each, %.1f s wall (slowest worker %.1f s)'
                       % (workers, pairs_per_worker, int(np.shape(td)[1]), wall, max(r['seconds'] for r in res)))


if __name__ == '__main__':
    if len(sys.argv) > 3 and sys.argv[3] == 'xc':
        _xc_worker(sys.argv[1], int(sys.argv[2]))
    else:
        _worker(sys.argv[1], int(sys.argv[2]))

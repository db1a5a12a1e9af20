#!/usr/bin/env python
"""Headline benchmark: detection windows/sec of the subspace detection scan
(BASELINE.json configs[1]: 200 basis vectors x 24 h of 3-channel 100 Hz data,
as 24 chunks of 3720 s, on one MI355X; N GPUs = N station-days, weak scaling).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one station-day: for each of the 24
chunks the detection statistic of all 25 detectors (8 basis vectors each) at
every window, plus the histogram / max / threshold-candidate pass (one
dtx_subspace_scan call).  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1 sparse figure)
# what a registers-only bf16 MFMA loop sustains on this chip with random operands (the power
# limit clocks it down to 1.96 GHz; 2479 with all-zero operands): scripts/probe/mfma_peak.hip,
# profiles/p1_mfma_peak.txt.  Reported next to the nominal peak, never instead of it.
SUSTAINED_BF16_MFMA_TFLOPS = 2057.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--detectors', type=int, default=25)
    ap.add_argument('--basis-per-detector', type=int, default=8)
    ap.add_argument('--chunks', type=int, default=24)
    ap.add_argument('--chunk-seconds', type=float, default=3720.0)   # 3600 + 120 (getdata.py:299)
    ap.add_argument('--sr', type=float, default=100.0)
    ap.add_argument('--nc', type=int, default=3)
    ap.add_argument('--template-seconds', type=float, default=30.0)  # subspace.py:1462
    ap.add_argument('--engine', default='auto', choices=['auto', 'fft', 'bf16x3', 'fp32'],
                    help="auto (library default): fft for templates of 256..32768 samples per channel, "
                         "else bf16x3; fft: frequency-domain overlap-save engine (float32 transforms in "
                         "LDS); bf16x3: split-bfloat16 MFMA contraction (3 bf16 products per float32 "
                         "product, float32 accumulate); fp32: exact float32 MFMA contraction")
    ap.add_argument('--prewarm', type=float, default=5.0,
                    help='cap (seconds) of the untimed scanning before the warm-up steps: the same step is '
                         'repeated until the scan kernel\'s duration has settled (mean of the last 10 launches '
                         'within 1 %% of the 10 before), at least --prewarm-min seconds, at most this; 0 = none')
    ap.add_argument('--prewarm-min', type=float, default=0.5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the secondary fields (other engines, configs[2] / configs[4], PCIe-inclusive '
                         'rate): what the profiling scripts use')
    ap.add_argument('--config', type=int, default=2, choices=[2, 4],
                    help='2 (default): BASELINE configs[1], one station-day per GPU and step (weak scaling); '
                         '4: BASELINE configs[3], 8 stations x 30 days dealt round-robin over the GPUs, one '
                         'step = all 240 station-days (strong scaling)')
    ap.add_argument('--stations', type=int, default=8)
    ap.add_argument('--days', type=int, default=30)
    ap.add_argument('--cpu-vectors', type=int, default=200,
                    help='basis vectors the CPU baseline sample covers')
    return ap.parse_args()


def cpu_baseline(U, det_off, x_chunk, Nc, nvec):
    """The reference's FFT algorithm (oracle restatement, NumPy/SciPy float64,
    one thread -- the reference has no parallelism, subspace.py:1843-1845) on ONE
    chunk and the first `nvec` basis vectors; extrapolated linearly in the number
    of basis vectors (the cost is one length-2^21 IFFT per vector)."""
    import scipy.fftpack
    from oracle import detex_oracle as orc
    x = x_chunk.astype(np.float64)
    n = U.shape[1]
    K = U.shape[0]
    rbi = 2 ** int(len(x) + n).bit_length()
    t0 = time.perf_counter()
    xFD = scipy.fftpack.fft(x, n=rbi)                       # detect.py:255-256
    done = 0
    d = 0
    while done < nvec and d < len(det_off) - 1:
        Ud = U[det_off[d]:det_off[d + 1]].astype(np.float64)
        orc.mpxds_fft(x, Ud, Nc, rbi=rbi, xFD=xFD)
        done += Ud.shape[0]
        d += 1
    dt = time.perf_counter() - t0
    P = (len(x) - n) // Nc + 1
    full = dt * K / done                                    # seconds per chunk, all K vectors
    from oracle import cpu_bench
    model, ncores, logical = cpu_bench.cpu_info()
    return dict(value=P / full, unit='windows/s', cores=1, kind='port', cpu_model=model,
                physical_cores=ncores, os_cpu_count=logical, effective_cores=cpu_bench.effective_cores(),
                cgroup_cpu_limit=cpu_bench.cgroup_cpu_limit(),
                sample='1 chunk (%d samples) x %d of %d basis vectors, float64 FFT path, '
                       '%.1f s measured, scaled linearly to %d vectors' % (len(x), done, K, dt, K))


def cpu_baseline_allcores(U, det_off, x, Nc, onecore_rate=None):
    """The same CPU restatement on every physical core of the host (SURVEY.md 8d / BASELINE.md 3):
    one worker process per core, each one chunk x 8 basis vectors (one detector).  The reference itself is
    single-threaded; this is the generous CPU figure."""
    from oracle import cpu_bench
    try:
        return cpu_bench.allcores_baseline(U, det_off, x, Nc, nvec_per_worker=8, timeout=240.0,
                                           onecore_rate=onecore_rate)
    except Exception as e:          # a bounded leg: never take the bench line down
        return dict(value=None, unit='windows/s', error=repr(e))


def _tag_order(path):
    """profiles/<letter><number>_...: the newest measurement of a series has the highest number (file times
    say nothing after a fresh checkout)."""
    import re
    m = re.match(r'([a-z]+)(\d+)', os.path.basename(path))
    return (m.group(1), int(m.group(2))) if m else ('', -1)


def kernel_source_digest():
    """sha1 over detex_amd/csrc/*.hip|*.h, as scripts/summarize_profile.py records it with a counter summary."""
    import glob
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(ROOT, 'detex_amd', 'csrc')
    for fn in sorted(glob.glob(os.path.join(base, '*.hip')) + glob.glob(os.path.join(base, '*.h'))):
        h.update(os.path.basename(fn).encode())
        with open(fn, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_summary(pattern, instance=None):
    """HBM-side bytes per launch from the newest committed rocprofv3 --pmc summary profiles/<pattern>
    (scripts/profile.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate passes; scripts/summarize_profile.py)."""
    import glob
    pm = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)), key=_tag_order)
    if not pm:
        return None
    with open(pm[-1]) as f:
        prof = json.load(f)
    b = prof.get('counters', {}).get('hbm_bytes_per_launch_corrected')
    # NOT a live counter: read from the committed rocprofv3 --pmc summary of the same command
    # (scripts/profile.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate passes).  The summary names the kernel
    # instantiation(s) it counted and the sha1 of the kernel sources it was built from: a summary of an
    # older kernel, or of another instantiation than the one that just ran, is flagged here
    inst = prof.get('kernel_instances')
    sha = prof.get('source_sha1')
    return {'bytes': b, 'source': 'profiles/' + os.path.basename(pm[-1]), 'live': False,
            'same_kernel_source': (sha == kernel_source_digest()) if sha else None,
            'same_instantiation': (any(instance.replace(' ', '') in k.replace(' ', '') for k in inst)
                                   if inst and instance else None)}


def roofline(eng, kms, K, n, Nc, L, D, P, chunks, flops, default_cfg, kernel_name=None):
    """Roofline object of the dominant kernel of one engine (kms: its average launch
    duration from HIP events on the launch stream)."""
    rm = (min(K, 208) + 15) // 16            # row blocks of the (first) K-tile, dtx_hankel.h

    def pmc(pattern, instance=None):
        # for the default configuration only
        return pmc_summary(pattern, instance) if default_cfg else None
    if eng == 'fft':
        N = 8192
        V = (N - L + 1) // 512 * 512 if L <= 5633 else 4096      # segmented templates advance by 4096
        nseg = 1 if L <= 5633 else (L + 4095) // 4096
        nblk = (P + V - 1) // V
        npairs = D * ((K // D + 1) // 2) if K % D == 0 else (K + 1) // 2
        nfft = npairs * nblk * chunks
        # algorithmic HBM bytes: the stream once (4 Nc B/window) + DS once (4 D B/window)
        bytes_alg = (4.0 * Nc + 4.0 * D) * P * chunks
        # what the kernel computes: one 8192-point complex transform (5 N log2 N) and Nc complex
        # MACs per frequency (8 flop each) per (pair, block)
        flop_fft = nfft * (5.0 * N * 13 + 8.0 * Nc * nseg * N)
        gbs = bytes_alg / (kms * 1e-3) / 1e9
        # single-segment templates on 1 or 3 channels run the ping-pong kernel (two half-workgroups
        # per CU taking turns on the transform image), everything else the one-group kernel
        pp = Nc in (1, 3) and nseg == 1
        kname = kernel_name or (('ds_fft_pp_kernel<%d>' if pp else 'ds_fft_kernel<%d>') % (V // 512))   # (the library's own name of what it dispatched)
        return {'bound': 'hbm', 'kernel': kname, 'achieved': gbs, 'peak': 8000.0,
                'unit': 'GB/s', 'frac': gbs / 8000.0,
                'traffic': pmc('*_bench_pmc_ds_fft_pp_kernel.json' if pp else '*_bench_pmc_ds_fft_kernel.json',
                               ('ds_fft_pp_kernel<%d,false,%d>' % (V // 512, Nc)) if pp else None),
                'kernel_ms': kms, 'bytes_per_launch': bytes_alg, 'engine': 'fft',
                'compute': {'unit': 'TFLOP/s', 'issued': flop_fft / (kms * 1e-3) / 1e12,
                            'peak_f32_vector': PEAK_FP32_MFMA_TFLOPS,
                            'frac': flop_fft / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                            # the kernel against its OWN floor: its butterflies and MACs at the f32 vector peak
                            'floor_ms': flop_fft / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3,
                            'times_floor': kms / (flop_fft / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3),
                            'contraction_equivalent': flops / (kms * 1e-3) / 1e12},
                # the vector ALUs' share of a round of the ping-pong kernel (one pair per CU and round; DESIGN.md
                # 4.1, profiles/g9_ablations.txt): 1000 instructions per pair and wave in the <10,false,3>
                # binary (scripts/valu_count.py = SQ_INSTS_VALU), four waves of ~500 per SIMD and round,
                # 2 cycles per instruction
                'valu_issue': ({'instructions_per_pair_and_wave': 1000,
                                'issue_cycles_per_round_and_simd': 4000,
                                'cycles_per_round_at_2.4GHz': kms * 2.4e6 * 256 / nfft,
                                'frac': 4000.0 / (kms * 2.4e6 * 256 / nfft)}
                               if pp and V == 5120 and Nc == 3 else None),
                'note': 'overlap-save engine: %d transforms of 8192 complex points per launch in LDS; '
                        'neither HBM- nor MFMA-bound -- the algorithmic HBM bytes (112 B/window) are '
                        'what the schema asks for, `compute` prices the transforms against the f32 '
                        'vector peak, `contraction_equivalent` is the direct contraction it replaces '
                        '(2*K*n flop per window); no unit is saturated (`valu_issue`: the vector ALUs\' share) -- the '
                        'bound is the dependent instruction chains of the four waves per SIMD, DESIGN.md 4.1' % nfft}
    if eng == 'fp32':
        ach = flops / (kms * 1e-3) / 1e12
        return {'bound': 'mfma', 'kernel': kernel_name or 'ds_mfma_kernel<%d,%d>' % (rm, 2 if rm >= 9 else (4 if rm >= 5 else 8)),
                'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / PEAK_FP32_MFMA_TFLOPS, 'traffic': pmc('*_bench_pmc_ds_mfma_kernel.json'),
                'kernel_ms': kms, 'flop_per_launch': flops, 'engine': 'fp32',
                'note': 'algorithmic flop = 2*K*n per window'}
    ach = flops / (kms * 1e-3) / 1e12
    issued = 3.0 * ach * (16.0 * rm) / min(K, 208) if K <= 208 else None
    return {'bound': 'mfma', 'kernel': 'ds_mfma_kernel_s<%d,%d>' % (rm, 3 if rm >= 9 else 4),
            'achieved': ach, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': ach / PEAK_BF16_MFMA_TFLOPS, 'traffic': pmc('*_bench_pmc_ds_mfma_kernel_s.json'),
            'kernel_ms': kms, 'flop_per_launch': flops, 'engine': 'bf16x3',
            'sustained_peak': SUSTAINED_BF16_MFMA_TFLOPS, 'issued_mfma_tflops': issued,
            'issued_frac_of_sustained': issued / SUSTAINED_BF16_MFMA_TFLOPS if issued else None,
            'note': 'algorithmic flop = 2*K*n per window; the split engine issues 3 bf16 MFMAs per '
                    'float32 product, so its ceiling is peak/3; issued_mfma_tflops counts those 3 MFMAs '
                    'on the zero-padded 16-row blocks and is to be read against sustained_peak (measured '
                    'power-limited rate of a registers-only bf16 MFMA loop on random operands, '
                    'profiles/p1_mfma_peak.txt)'}


def pcie_inclusive(ctx, x, ds, thr, hist, windows, steps, barrier=None):
    """The same step with the station-day handed over as a HOST buffer: pinned memory, the next
    day's H2D copy on a second stream under the current day's scan (two device buffers), results
    (candidates, maxima, histograms -- a few KB) fetched every step, DS left on the device."""
    xp = torch.from_numpy(x).pin_memory()
    bufs = [torch.empty(x.shape, dtype=torch.float32, device='cuda') for _ in range(2)]
    copy = torch.cuda.Stream()
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    main = torch.cuda.current_stream()

    def upload(i):
        with torch.cuda.stream(copy):
            copy.wait_event(done[i])
            bufs[i].copy_(xp, non_blocking=True)
            ready[i].record(copy)

    for i in range(2):
        done[i].record(main)
    thr = torch.from_numpy(np.asarray(thr, dtype=np.float32)).cuda()
    prev = None
    # warm-up, pipelined like the timed loop: result buffers of BOTH steps in flight; ten steps -- the first uses of the
    # freshly pinned buffer and of the two device buffers run slow (scripts/probe/pcie_leg.py: 3.55 ms per step
    # over the first 20 steps of a process, 3.28 from then on)
    for i in range(10):
        upload(i & 1)
        main.wait_event(ready[i & 1])
        f = ctx.subspace_scan(bufs[i & 1], out=ds, thr=thr, hist=hist, lazy=True)[1]
        done[i & 1].record(main)
        if prev is not None:
            prev()
        prev = f
    prev()
    torch.cuda.synchronize()
    if barrier is not None:                  # N > 1: every rank's uploads start together (the host feed is what they share)
        barrier()
    t0 = time.perf_counter()
    upload(0)
    pending = None
    for i in range(steps):
        cur = i & 1
        if i + 1 < steps:
            upload(1 - cur)
        main.wait_event(ready[cur])
        _, fetch = ctx.subspace_scan(bufs[cur], out=ds, thr=thr, hist=hist, lazy=True)
        done[cur].record(main)
        if pending is not None:              # the previous step's results, while this one runs (as the main loop)
            pending()
        pending = fetch
    pending()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = sorted(ctx.kernel_ms_list(min(steps, 64)))
    return {'value': windows * steps / dt, 'unit': 'windows/s', 'ms_per_step': 1e3 * dt / steps,
            'steps': steps, 'kernel_ms': ctx.kernel_ms_mean(min(steps, 64)),
            'kernel_ms_min': ks[0], 'kernel_ms_median': ks[len(ks) // 2], 'kernel_ms_max': ks[-1],
            'note': 'host buffers: %.0f MB per station-day from pinned memory, double-buffered on a copy '
                    'stream; candidate list / maxima / histograms copied back, DS stays in HBM'
                    % (x.nbytes / 1e6)}


def host_fed_all_ranks(leg, rank, world, windows_per_rank_step, binding):
    """The N > 1 line's second number (VERDICT r5 missing 1): every rank runs its host-fed leg AT THE SAME TIME --
    station-days from page-locked host memory, the next upload under the current scan -- so that a scaling curve
    also shows the one resource the ranks share, the host feed (~33 GB/s per GPU at 3.2 ms per
    station-day).  leg(barrier) -> {'ms_per_step', 'steps', ...} of this rank; gathered over ranks: min / max / mean
    of the ranks' ms per step and the aggregate rate at the slowest rank's pace, beside each rank's CPU binding."""
    import torch.distributed as dist

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
    mine = leg(barrier)
    rows = [None] * world
    dist.all_gather_object(rows, {'rank': rank, 'ms_per_step': mine['ms_per_step'], 'steps': mine['steps'],
                                  'kernel_ms': mine.get('kernel_ms'), 'binding': dict(binding)})
    ms = [r['ms_per_step'] for r in rows]
    return {'unit': 'windows/s', 'value': world * windows_per_rank_step / (max(ms) * 1e-3),
            'ms_per_step_min': min(ms), 'ms_per_step_max': max(ms), 'ms_per_step_mean': sum(ms) / len(ms),
            'steps': mine['steps'], 'per_rank': rows,
            'note': 'every rank at once: station-days from pinned host memory, double-buffered H2D on a copy stream '
                    'under the previous scan; value = all ranks\' windows per step / the slowest rank\'s step'}


def dist_overhead(local, timed_loop, steps, ms_plain, Uh, det_off, n):
    """The same timed loop with a process group initialised: RCCL (backend 'nccl') at world size 1 on this
    one GPU -- the rendezvous, the basis broadcast, the barriers around the timed region and the
    max-over-ranks reduction of the driver's N > 1 contract all execute, only the peer count differs.
    Reported next to the plain step so that the cost of being launched under torch.distributed is visible
    in the N = 1 line (the N > 1 curve itself is the driver's to measure)."""
    import socket
    import torch.distributed as dist
    from detex_amd import parallel
    if dist.is_initialized():
        return None
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dev = torch.device('cuda', local)
    t0 = time.perf_counter()
    import datetime
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    try:
        dist.barrier()
        torch.cuda.synchronize()
        t_init = time.perf_counter() - t0
        t0 = time.perf_counter()
        Ut, off = parallel.broadcast_basis(Uh, det_off, n, dev)        # three RCCL broadcasts (sizes, offsets, basis)
        torch.cuda.synchronize()
        t_bc = time.perf_counter() - t0
        assert np.array_equal(Ut.cpu().numpy(), Uh) and list(off) == list(det_off)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
        timed_loop(min(steps, 5), barrier)
        dt, _ = timed_loop(steps, barrier)
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finally:
        dist.destroy_process_group()
    ms = 1e3 * dt / steps
    return {'backend': 'nccl (RCCL)', 'world_size': 1, 'steps': steps, 'ms_per_step_plain': ms_plain,
            'ms_per_step_dist': ms, 'ratio': ms / ms_plain, 'init_s': t_init, 'basis_broadcast_ms': 1e3 * t_bc,
            'note': 'process group initialised in-process; barrier + synchronize on both sides of the timed '
                    'region, all_reduce(MAX) of the elapsed time -- the driver\'s N > 1 bracket at N = 1'}


def xcorr_full(ctx, synth, Nc, cpu=True):
    """BASELINE configs[2] at full size: 2000 events x (3 x 13000) samples, 1 999 000 pairs."""
    N, L = 2000, 13000
    n = Nc * L
    td_host = synth.event_families(2, 100, 20, Nc, L)
    td = torch.from_numpy(td_host).cuda()
    npairs = N * (N - 1) // 2
    S = L - L // 2
    flop_pair = 2.0 * Nc * ((2 * S + 1) * L - S * (S + 1))        # 0.7605 GFLOP
    res = {}
    for eng in ('auto', 'bf16x3'):
        ctx.set_engine(eng)
        ctx.xcorr_allpairs(td, Nc)
        ctx.sync()
        t1 = time.perf_counter()
        cc, lag, ss = ctx.xcorr_allpairs(td, Nc)
        ctx.sync()
        res[eng] = (time.perf_counter() - t1, ctx.last_kernel_ms(), float(cc.mean()))
        if eng == 'auto':
            kname_full = ctx.last_kernel_name()
    ctx.set_engine('auto')
    dt, kms, ccm = res['auto']
    dtm, kmsm, _ = res['bf16x3']
    # SURVEY 8d's short variant of the same catalogue: events of 3000 samples per channel (n = 9000)
    short = None
    try:
        tds = torch.from_numpy(synth.event_families(2, 100, 20, Nc, 3000)).cuda()
        ctx.xcorr_allpairs(tds, Nc)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.xcorr_allpairs(tds, Nc)
        ctx.sync()
        ks = ctx.last_kernel_ms()
        kname_short = ctx.last_kernel_name()
        Ls = 3000
        Ss = Ls - Ls // 2
        b_s = 4.0 * N * Nc * Ls + 12.0 * npairs
        short = {'seconds': time.perf_counter() - t1, 'kernel_ms': ks, 'value': npairs / (time.perf_counter() - t1),
                 'unit': 'pairs/s',
                 'workload': 'configs[2] short variant: %d events, Nc=%d, L=3000 (n=%d)' % (N, Nc, 3000 * Nc),
                 # one segment, one lag block (3001 lags in a block of 8192): one transform and Nc products per event pair;
                 # outside the lag search: the float64 finish (xc_finish2_kernel) + preparation, seconds - kernel_ms
                 'outside_lag_search_ms': 1e3 * (time.perf_counter() - t1) - ks,
                 'roofline': {'bound': 'hbm', 'kernel': kname_short, 'kernel_ms': ks, 'bytes_per_launch': b_s,
                              'achieved': b_s / (ks * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                              'frac': b_s / (ks * 1e-3) / 1e9 / 8000.0, 'traffic': None,
                              'compute': {'unit': 'TFLOP/s', 'peak_f32_vector': PEAK_FP32_MFMA_TFLOPS,
                                          'issued': (npairs / 2.0) * (5.0 * 8192 * 13 + Nc * 8.0 * 8192) / (ks * 1e-3) / 1e12,
                                          'contraction_equivalent': 2.0 * Nc * ((2 * Ss + 1) * Ls - Ss * (Ss + 1)) * npairs
                                          / (ks * 1e-3) / 1e12}}}
        if cpu:
            from oracle import cpu_bench
            try:
                short['cpu_baseline'] = cpu_bench.xcorr_onecore(tds.cpu().numpy(), Nc, npairs=2000)
            except Exception as e:                      # noqa: BLE001
                short['cpu_baseline'] = dict(value=None, unit='pairs/s', error=repr(e))
        del tds
    except Exception as e:                              # noqa: BLE001 -- an extra never takes the line down
        short = {'error': repr(e)}
    # frequency-domain lag search (dtx_xcfft.hip): events of 13 000 samples are cut into 4 segments of
    # 3585, the 13 001 lags into 3 blocks of 4608; per (later event c, lag block, event pair) one
    # 8192-point complex transform and one MAC step per (segment that sees data, channel): 11 x Nc steps
    # per three transforms
    nblk, nprod = 3, 11
    nfft = (npairs / 2.0) * nblk
    flop_fft = nfft * 5.0 * 8192 * 13 + (npairs / 2.0) * nprod * Nc * 8.0 * 8192
    # per product: half a pair spectrum (one serves two later events) + half of a staged block spectrum (the 2 x 2 workgroup)
    l2_bytes = (npairs / 2.0) * nprod * Nc * (8192 * 8 / 2.0 + 4096 * 8 / 2.0)
    bytes_alg = 4.0 * N * n + 12.0 * npairs                        # events once + 12 B per pair
    # counters of the committed rocprofv3 --pmc passes of scripts/bench_xcorr.py (scripts/profile_xc.sh), newest tag
    import glob
    pmc, pmc_src, pmc_same = {}, None, None
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_xcorr_pmc.json')), key=_tag_order)[-1:]:
        with open(fn) as f:
            prof = json.load(f)
        pmc, pmc_src = prof.get('kernels', {}), 'profiles/' + os.path.basename(fn)
        if prof.get('source_sha1'):
            pmc_same = prof['source_sha1'] == kernel_source_digest()

    def cnt(kern, name):
        v = pmc.get(kern, {}).get(name)
        return v['mean_per_launch'] if isinstance(v, dict) else v
    xk = 'xc_fft22_kernel' if pmc.get('xc_fft22_kernel') else 'xc_fft_kernel'      # (summaries up to x8: the round-3 kernel)
    hbm = cnt(xk, 'hbm_bytes_per_launch_corrected')
    l2req = cnt(xk, 'TCC_REQ_sum')
    cpu_leg = None
    if cpu:
        # the reference's _CCX2 (construct.py:425-466) on this host: one core (the reference is
        # single-threaded) and all physical cores; a bounded sample of the same events
        from oracle import cpu_bench
        try:
            cpu_leg = cpu_bench.xcorr_onecore(td_host, Nc, npairs=400)
            cpu_leg['all_cores'] = cpu_bench.xcorr_allcores(td_host, Nc, pairs_per_worker=60,
                                                            onecore_rate=cpu_leg['value'])
        except Exception as e:                          # noqa: BLE001 -- a bounded leg never takes the line down
            cpu_leg = dict(value=None, unit='pairs/s', error=repr(e))
    return {'value': npairs / dt, 'unit': 'pairs/s', 'seconds': dt,
            'workload': 'configs[2]: %d events, Nc=%d, L=%d (n=%d), %d pairs' % (N, Nc, L, n, npairs),
            'cc_mean': ccm, 'cpu_baseline': cpu_leg, 'short_variant': short,
            'roofline': {'bound': 'hbm', 'kernel': kname_full, 'kernel_ms': kms,
                         'achieved': bytes_alg / (kms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': bytes_alg / (kms * 1e-3) / 1e9 / 8000.0,
                         'traffic': ({'bytes': hbm, 'source': pmc_src, 'live': False,
                                      'same_kernel_source': pmc_same} if hbm else None),
                         'bytes_per_launch': bytes_alg,
                         'bound_actual': 'the sum of its phases, run one after the other by one workgroup of sixteen '
                                         'waves per CU: products (L2 -> L1 stream of pair and block spectra, 48 KB per '
                                         'product, ~0.8 of the L1 fill rate), transforms, lag search '
                                         '(counters: %s; DESIGN 4.3) -- the algorithmic HBM bytes the '
                                         'schema prices are 3 orders of magnitude below what the frequency-domain '
                                         'search moves on chip' % pmc_src,
                         'l2_counters': ({'TCC_REQ_sum': l2req, 'bytes_at_128B': 128.0 * l2req,
                                          'TBps': 128.0 * l2req / (kms * 1e-3) / 1e12,
                                          'hit_rate': cnt(xk, 'TCC_HIT_sum') / l2req,
                                          'source': pmc_src, 'live': False} if l2req else None),
                         'compute': {'unit': 'TFLOP/s', 'issued': flop_fft / (kms * 1e-3) / 1e12,
                                     'peak_f32_vector': PEAK_FP32_MFMA_TFLOPS,
                                     'frac': flop_fft / (kms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                     'contraction_equivalent': flop_pair * npairs / (kms * 1e-3) / 1e12},
                         'l2': {'unit': 'TB/s', 'model_bytes': l2_bytes,
                                'achieved': l2_bytes / (kms * 1e-3) / 1e12, 'peak': 34.5,
                                'frac': l2_bytes / (kms * 1e-3) / 1e12 / 34.5},
                         'note': 'frequency-domain lag search: %.3g transforms of 8192 points and %d MAC steps per '
                                 'three of them, whose spectra stream from L2 (model: one 64 KB pair spectrum per '
                                 'transform and step, block spectra shared by two pairs); algorithmic HBM bytes = '
                                 'events once + 12 B per pair' % (nfft, nprod * Nc)},
            'mfma_engine': {'value': npairs / dtm, 'unit': 'pairs/s', 'seconds': dtm,
                            'roofline': {'bound': 'mfma', 'kernel': 'xc_mfma_kernel_s<13,2>', 'kernel_ms': kmsm,
                                         'achieved': flop_pair * npairs / (kmsm * 1e-3) / 1e12,
                                         'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                         'frac': flop_pair * npairs / (kmsm * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                         'flop_per_pair': flop_pair,
                                         'note': 'split-bf16 contraction (3 MFMAs per product: ceiling peak / 3)'}}}


def config5(ctx, synth):
    """BASELINE configs[4]: 500 Hz 6-channel borehole array, one 64-dimensional subspace, 2 s
    window (n = 6000), one hour of data."""
    Nc, L, K, T = 6, 1000, 64, 1800000
    n = Nc * L
    U, off = synth.detector_bank(5, 1, K, n)
    x, plants = synth.continuous_day(5, 1, Nc * T, Nc, U, off, nplant=10)
    xt = torch.from_numpy(x).cuda()
    thr = np.full(1, 0.3, np.float32)
    res = {}
    for eng in ('auto', 'fp32'):
        ctx.set_engine(eng)
        ctx.upload_basis(U, off, Nc)
        ds, r = ctx.subspace_scan(xt, thr=thr)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            ds, r = ctx.subspace_scan(xt, out=ds, thr=thr)
        ctx.sync()
        res[eng] = ((time.perf_counter() - t1) / 5, ctx.last_kernel_ms())
        if eng == 'auto':
            kname5 = ctx.last_kernel_name()
        cb, cd, cp, cv = r['cand']
        assert all(((cp - p) == 0).any() for (_, p, _) in plants), 'config5: planted event missed'
    ctx.set_engine('auto')
    P = T - L + 1
    flops = 2.0 * K * n * P
    # the same hour per call from a host that keeps calls in flight (results of call i fetched while call i + 1 runs,
    # as the headline loop and SubSpace.detex do): the post pass then runs beside the next call's preparation
    piped = None
    try:
        ctx.upload_basis(U, off, Nc)
        thr_dev = torch.from_numpy(thr).cuda()
        pend = None
        for it in range(25):
            if it == 5:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            f = ctx.subspace_scan(xt, out=ds, thr=thr_dev, lazy=True)[1]
            if pend is not None:
                pend()
            pend = f
        pend()
        torch.cuda.synchronize()
        piped = {'ms_per_hour_of_data': 1e3 * (time.perf_counter() - t1) / 20, 'steps': 20,
                 'kernel_ms': ctx.kernel_ms_mean(20)}
    except Exception as e:                              # noqa: BLE001 -- an extra never takes the line down
        piped = {'error': repr(e)}
    # a DAY per call, the way configs[1] hands its 24 chunks over: 24 hourly chunks (views of one buffer would do; here
    # the hour repeated) in ONE dtx_subspace_scan -- preparation kernels and launch boundaries are paid once per day
    day = None
    try:
        ctx.upload_basis(U, off, Nc)
        xd = xt.reshape(1, -1).expand(24, -1).contiguous()
        dsd, rd = ctx.subspace_scan(xd, thr=thr)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            dsd, rd = ctx.subspace_scan(xd, out=dsd, thr=thr)
        ctx.sync()
        dtd = (time.perf_counter() - t1) / 3
        kd = ctx.last_kernel_ms()
        day = {'value': 24 * P / dtd, 'unit': 'windows/s', 'ms_per_hour_of_data': 1e3 * dtd / 24, 'kernel_ms_per_hour': kd / 24,
               'step_over_kernel': 1e3 * dtd / kd, 'hours_per_call': 24}
        del xd, dsd, rd
    except Exception as e:                              # noqa: BLE001 -- an extra never takes the line down
        day = {'error': repr(e)}
    ka = res['auto'][1]
    b_alg = (4.0 * Nc + 4.0) * P                        # the stream once + one detector's DS once
    nblk5 = -(-P // ((8192 - L + 1) // 512 * 512))
    flop_fft = (K / 2) * nblk5 * (5.0 * 8192 * 13 + 8.0 * Nc * 8192)
    return {'value': P / res['auto'][0], 'unit': 'windows/s', 'ms_per_hour_of_data': 1e3 * res['auto'][0],
            'kernel_ms': res['auto'][1], 'step_over_kernel': 1e3 * res['auto'][0] / res['auto'][1],
            'day_per_call': day, 'pipelined': piped,
            'roofline': {'bound': 'hbm', 'kernel': kname5, 'kernel_ms': ka, 'bytes_per_launch': b_alg,
                         'achieved': b_alg / (ka * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': b_alg / (ka * 1e-3) / 1e9 / 8000.0,
                         'traffic': pmc_summary('*c5_bench_pmc_ds_fft_kernel.json', kname5),
                         'compute': {'unit': 'TFLOP/s', 'peak_f32_vector': PEAK_FP32_MFMA_TFLOPS,
                                     'issued': flop_fft / (ka * 1e-3) / 1e12,
                                     'frac': flop_fft / (ka * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                     'contraction_equivalent': flops / (ka * 1e-3) / 1e12},
                         'note': 'frequency-domain engine, block spectra streamed (6 channels do not fit beside the image): '
                                 '%d transforms of 8192 points per launch; 28 algorithmic bytes per window' % ((K // 2) * nblk5)},
            'workload': 'configs[4]: 500 Hz, 6 channels, K=64, n=6000, 1 h (%d windows)' % P,
            'fp32_mfma_engine': {'value': P / res['fp32'][0], 'unit': 'windows/s', 'kernel_ms': res['fp32'][1],
                                 'roofline': {'bound': 'mfma', 'achieved': flops / (res['fp32'][1] * 1e-3) / 1e12,
                                              'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                              'frac': flops / (res['fp32'][1] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}}}


def config2_singles(ctx, synth, a, xt, M, Nc, n, cpu_rate=None):
    """SURVEY 8d's variant of configs[1]: the same 200 vectors as 200 single-vector detectors (useSingles,
    detect.py:319-388, subspace.py:1871-1880) -- 200 DS rows per window instead of 25.  Here the DS tensor is real HBM
    traffic: 812 B per window, 7.2 GB per station-day written by the scan and read once more by the post pass."""
    D = a.detectors * a.basis_per_detector
    U, off = synth.detector_bank(1, D, 1, n)
    ctx.upload_basis(U, off, Nc)
    P = ctx.num_windows(M)
    ldp = (P + 63) // 64 * 64
    ds = torch.empty((a.chunks, D, ldp), dtype=torch.float32, device='cuda')[:, :, :P]
    thr = torch.full((D,), 0.3, dtype=torch.float32, device='cuda')
    hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
    for _ in range(3):
        ctx.subspace_scan(xt, out=ds, thr=thr, hist=hist, lazy=True)[1]()
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    pending = None
    for _ in range(steps):
        f = ctx.subspace_scan(xt, out=ds, thr=thr, hist=hist, lazy=True)[1]
        if pending is not None:
            pending()
        pending = f
    pending()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kms = ctx.kernel_ms_mean(steps)
    # the post pass alone (histogram / maxima / candidates over the finished tensor): timed apart, HIP events
    ctx.timer_start()
    for _ in range(3):
        ctx.ds_post(ds, thr=thr, hist=hist)
    post_ms = ctx.timer_stop() / 3
    W = a.chunks * P
    b_scan = (4.0 * Nc + 4.0 * D) * W                 # stream once + DS written once (812 B per window)
    b_post = 4.0 * D * W                              # DS read once more
    out = {'value': W / dt, 'unit': 'windows/s', 'ms_per_step': 1e3 * dt, 'kernel_ms': kms, 'post_ms': post_ms,
           'workload': 'configs[1] as singles: %d detectors x 1 vector x %d chunks, n=%d; DS tensor %.2f GB'
                       % (D, a.chunks, n, 4.0 * D * W / 1e9),
           'roofline': {'bound': 'hbm', 'kernel': ctx.last_kernel_name() + ' + ds_post_kernel',
                        'bytes_per_launch': b_scan + b_post, 'kernel_ms': kms + post_ms,
                        'achieved': (b_scan + b_post) / ((kms + post_ms) * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                        'frac': (b_scan + b_post) / ((kms + post_ms) * 1e-3) / 1e9 / 8000.0, 'traffic': None,
                        'scan': {'bytes': b_scan, 'ms': kms, 'GBps': b_scan / (kms * 1e-3) / 1e9,
                                 'frac': b_scan / (kms * 1e-3) / 1e9 / 8000.0},
                        'post': {'bytes': b_post, 'ms': post_ms, 'GBps': b_post / (post_ms * 1e-3) / 1e9,
                                 'frac': b_post / (post_ms * 1e-3) / 1e9 / 8000.0},
                        'step': {'bytes': b_scan + b_post, 'ms': 1e3 * dt, 'GBps': (b_scan + b_post) / dt / 1e9,
                                 'frac': (b_scan + b_post) / dt / 1e9 / 8000.0}}}
    if cpu_rate:
        # the reference's cost per (basis vector, chunk) does not depend on how the vectors are grouped into
        # detectors (one IFFT per vector, detect.py:559-578): the configs[1] CPU figure applies as it is
        out['cpu_baseline'] = {'value': cpu_rate, 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
                               'sample': 'the configs[1] sample (same 200 vectors, one IFFT per vector and chunk)'}
    del ds
    return out


def config2_dirty(ctx, synth, a, x, plants, ds, thr_dev, hist, Nc, n, L, D, P, clean_ms):
    """configs[1] as real records look: the same station-day with 10 zero-filled gaps of 60 s (all
    channels; getdata's fillZeros, detect.py:231, construct.py:1038) and 10 single-sample glitches of 1e5 noise sigmas
    on top of the 50 planted events.  The reference's float64 transforms pay nothing extra for such data
    (detect.py:275-281, 559-578); here the dynamic-range guard lists the blocks they touch, splits the glitches off the
    transforms (re-scan with their products added directly) and sends what is still out of range to the float32
    contraction in tasks.  Same pipelined loop as the headline; `over_clean` against the `steady` leg.
    A second draw with 1000 glitches (two blocks in three listed) prices the worst case."""
    def loop(xt, steps):
        for _ in range(5):
            ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]()
        cnt = ctx.repair_counts()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = res = None
        for _ in range(steps):
            f = ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]
            if pend is not None:
                res = pend()
            pend = f
        res = pend()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, ctx.kernel_ms_mean(min(steps, 64)), cnt, res
    xd = x.copy()
    gaps, spikes = synth.make_dirty(xd, Nc, 11, ngap=10, gap_frames=6000, nspike=10, spike_amp=1e5)
    xt = torch.from_numpy(xd).cuda()
    steps = 40
    dt, kms, cnt, res = loop(xt, steps)
    cb, cd, cp, cv = res['cand']
    alive = [(b, p, d) for (b, p, d) in plants
             if not any(bb == b and t - L < p < t + g for (bb, t, g) in gaps)]
    found = sum(bool(((cb == b) & (cd == d) & (np.abs(cp - p) <= 2)).any()) for (b, p, d) in alive)
    W = a.chunks * P
    bts = (4.0 * Nc + 4.0 * D) * W
    out = {'value': W / dt, 'unit': 'windows/s', 'ms_per_step': 1e3 * dt, 'steps': steps, 'kernel_ms': kms,
           'clean_ms_per_step': clean_ms, 'over_clean': 1e3 * dt / clean_ms if clean_ms else None,
           'workload': 'configs[1] + 10 gaps of 60 s (zero-filled, all channels) + 10 glitches of 1e5 sigma + the 50 planted events',
           'listed_blocks': cnt['blocks'], 'blocks_rescanned_without_outliers': cnt['split'],
           'tiles_to_float32_contraction': cnt['tiles'], 'blocks_total': None,
           'planted_found': '%d/%d' % (found, len(alive)), 'planted_lost_to_gaps': len(plants) - len(alive),
           'roofline': {'bound': 'hbm', 'kernel': 'the whole step (scan kernel + what the guard lists + post pass)',
                        'bytes_per_launch': bts, 'achieved': bts / dt / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                        'frac': bts / dt / 1e9 / 8000.0, 'traffic': None,
                        'scan_kernel': {'ms': kms, 'frac': bts / (kms * 1e-3) / 1e9 / 8000.0}}}
    assert found == len(alive), 'dirty leg: planted events missed: %d of %d' % (found, len(alive))
    xd = x.copy()
    synth.make_dirty(xd, Nc, 3, ngap=0, nspike=1000, spike_amp=1e5)
    xt = torch.from_numpy(xd).cuda()
    dt2, kms2, cnt2, _ = loop(xt, 10)
    out['glitches_1000'] = {'ms_per_step': 1e3 * dt2, 'kernel_ms': kms2, 'over_clean': 1e3 * dt2 / clean_ms if clean_ms else None,
                            'listed_blocks': cnt2['blocks'], 'blocks_rescanned_without_outliers': cnt2['split'],
                            'tiles_to_float32_contraction': cnt2['tiles'],
                            'workload': 'configs[1] + 1000 glitches of 1e5 sigma'}
    return out


def e2e_createcluster(local, cpu=True):
    """BASELINE configs[2] through the PRODUCT API: createCluster on an EventFetcher of 2000 events x 3 x 13000
    samples (one station) -- event matrix, upload, kernels, download, the three 1999 x 1999 frames, SciPy single
    linkage of 1 999 000 dissimilarities, Cluster.updateReqCC -- with the seconds of every phase
    (ClusterStream.timings)."""
    from detex_amd import EventFetcher, createCluster, synth
    N, Nc, L = 2000, 3, 13000
    td = synth.event_families(2, 100, 20, Nc, L)
    out = {'workload': 'configs[2] through createCluster: %d events, Nc=%d, L=%d, one station, saveclust=False'
                       % (N, Nc, L), 'unit': 'pairs/s'}
    npairs = N * (N - 1) // 2
    for dtype in ('single', 'double'):
        ef = EventFetcher(100.0, ['HHE', 'HHN', 'HHZ'])
        for e in range(N):
            # ([Nc, L] contiguous rows: what an ObsPy stream's traces are)
            ef.add('TA.M17A', 'ev%04d' % e, np.ascontiguousarray(td[e].reshape(L, Nc).T), starttime=1000.0 + 200.0 * e,
                   origintime=990.0 + 200.0 * e, magnitude=1.0)
        createCluster(CCreq=0.5, fetch_arg=ef, device=local, saveclust=False, dtype=dtype)      # warm: workspaces
        times, last = [], None
        for _ in range(2):
            t0 = time.perf_counter()
            last = createCluster(CCreq=0.5, fetch_arg=ef, device=local, saveclust=False, dtype=dtype)
            times.append(time.perf_counter() - t0)
        dt = min(times)
        out[dtype] = {'seconds': dt, 'value': npairs / dt, 'calls_s': times,
                      'phases_s': {k: round(v, 4) for k, v in last.timings.items()},
                      'clusters': len(last[0].clusts), 'singles': len(last[0].singles)}
        del ef, last
    out['value'] = out['double']['value']                  # the reference's default dtype
    out['seconds'] = out['double']['seconds']
    # the call's dominant part is the same lag search + float64 finish as xcorr_full (its `kernels` phase also holds the
    # interleave of the event matrix): the schema's figure over that phase
    ksec = out['double']['phases_s'].get('kernels')
    if ksec:
        b_alg = 4.0 * N * Nc * L + 12.0 * npairs
        out['roofline'] = {'bound': 'hbm', 'kernel': 'xc_fft22_kernel + xc_finish2_kernel (phase `kernels`)',
                           'kernel_ms': 1e3 * ksec, 'bytes_per_launch': b_alg, 'achieved': b_alg / ksec / 1e9,
                           'peak': 8000.0, 'unit': 'GB/s', 'frac': b_alg / ksec / 1e9 / 8000.0, 'traffic': None,
                           'share_of_call': ksec / out['double']['seconds']}
    if cpu:
        from oracle import cpu_bench
        try:
            out['cpu_baseline'] = cpu_bench.xcorr_onecore(td, Nc, npairs=200)
            out['cpu_baseline']['note'] = ('the reference spends its time in _CCX2 (32 ms per pair measured in the build '
                                           'container, SURVEY 8a): pairs/s of the kernel leg is the comparable figure')
        except Exception as e:                              # noqa: BLE001
            out['cpu_baseline'] = dict(value=None, unit='pairs/s', error=repr(e))
    return out


def e2e_fas(local, cpu_rate=None):
    """SubSpace.getFAS(conDatNum=100) (subspace.py:1652-1743 -> fas._initFAS, fas.py:23-117) for the 25 detectors of
    one station over a 5-day record (120 hourly chunks, 100 of them sampled): STA/LTA screen of every candidate
    chunk on the host, the detection statistic of every detector on every chosen chunk (the scan's kernel),
    histogram + beta-fit moments on the device, 25 beta fits (SciPy) on the host."""
    from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace, synth
    SR, NC, L, NFAM, NPER, NDAY = 100.0, 3, 3300, 25, 8, 5
    rs = np.random.RandomState(3)
    td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
    ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    for e in range(len(td)):
        ef.add('TA.M17A', 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e,
               origintime=990.0 + 100 * e, magnitude=1.0)
    T = int(NDAY * 24 * 3600 * SR)
    con = rs.standard_normal((NC, T)).astype(np.float32)
    cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    cf.add('TA.M17A', con, starttime=1.5e9 - 1.5e9 % 3600)
    cl = createCluster(CCreq=0.5, fetch_arg=ef, device=local, saveclust=False)
    ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
    ss.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
    ss.getFAS(100, recalc=True, device=local, seed=0)                  # warm
    torch.cuda.synchronize()
    times = []
    for k in range(3):
        t0 = time.perf_counter()
        ss.getFAS(100, recalc=True, device=local, seed=k)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    names, kinds, U, off, _, _ = ss._detector_bank('TA.M17A', useSingles=False)
    n = U.shape[1]
    P = (int(round(3720.0 * SR)) * NC - n) // NC + 1
    fas0 = ss.subspaces['TA.M17A'].FAS.iloc[0]
    out = {'seconds': dt, 'calls_s': times, 'value': 100 * P / dt, 'unit': 'windows/s',
           'workload': 'getFAS(conDatNum=100): %d detectors (%d vectors, n=%d) x 100 chunks of 3720 s drawn from %d, '
                       'one station' % (len(names), U.shape[0], n, len(cf.chunk_bounds('TA.M17A'))),
           'ms_per_chunk': 1e3 * dt / 100, 'beta_first_detector': [float(v) for v in fas0['betadist'][:2]],
           'windows_in_histogram': int(np.sum(fas0['hist']))}
    if cpu_rate:
        # fas._MPXSSCorr is the bit-equal twin of _MPXDS (fas.py:120-134): the reference's cost is the scan's
        out['cpu_baseline'] = {'value': cpu_rate * 200.0 / U.shape[0], 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
                               'sample': 'the configs[1] sample scaled to %d vectors (one IFFT per vector and chunk)'
                                         % U.shape[0]}
    return out


def e2e_detex(local):
    """The configs[1] workload through the reference's API (subspace.py:1745-1902, detect.py:137-218):
    createCluster -> createSubSpace -> SVD on 25 families x 8 events (25 detectors x 8 basis vectors of
    30 s), then SubSpace.detex(estimateMags=True) over 8 stations x 1 day (24 chunks of 3720 s each) held
    by a pinned ContinuousFetcher: uploads, scans, trigger picking, STA/LTA, magnitudes and the result
    tables are all inside the timed region; once without and once with the SQLite file."""
    import tempfile
    from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace, synth
    SR, NC, L, NFAM, NPER, NSTA = 100.0, 3, 3300, 25, 8, 8
    rs = np.random.RandomState(0)
    td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
    stas = ['TA.S%02dA' % i for i in range(NSTA)]
    ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    for sta in stas:
        for e in range(len(td)):
            ef.add(sta, 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e,
                   origintime=990.0 + 100 * e, magnitude=1.0 + (e % 5) * .2)
    T = int(24 * 3600 * SR)
    con = rs.standard_normal((NC, T)).astype(np.float32)
    nplant = 50
    for k in range(nplant):
        e = int(rs.randint(len(td)))
        t = int(rs.randint(0, T - L))
        con[:, t:t + L] += 3.0 * td[e].reshape(L, NC).T / np.std(td[e])
    cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    for i, sta in enumerate(stas):
        cf.add(sta, con if i == 0 else con.copy(), starttime=1.5e9)
    cf.pin()
    t0 = time.perf_counter()
    cl = createCluster(CCreq=0.5, fetch_arg=ef, device=local)
    ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
    ss.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
    ss.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    names, kinds, U, off, _, _ = ss._detector_bank(stas[0], useSingles=False)
    n = U.shape[1]
    P = (int(round(3720.0 * SR)) * NC - n) // NC + 1
    windows = NSTA * 24 * P
    out = {'workload': '%d stations x 1 day (24 chunks of 3720 s, 3-ch 100 Hz) through createCluster / createSubSpace / '
                       'SVD / SubSpace.detex(estimateMags=True); %d detectors, %d basis vectors, n = %d; pinned fetcher'
                       % (NSTA, len(names), U.shape[0], n),
           'setup_s': t_setup, 'unit': 'windows/s'}
    with tempfile.TemporaryDirectory() as tmp:
        for key, db in (('no_db', None), ('sqlite', os.path.join(tmp, 'SubSpace.db'))):
            res = ss.detex(subspaceDB=db, estimateMags=True, device=local)       # warm: contexts, basis images
            torch.cuda.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                res = ss.detex(subspaceDB=db, estimateMags=True, device=local)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[key] = {'value': windows / dt, 'ms_per_call': 1e3 * dt, 'ms_per_station_day': 1e3 * dt / NSTA,
                        'detections': int(len(res['ss_df']))}
    # the same with an UNFILTERED fetcher: every chunk is detrended and band-passed on the device before its scan
    # (the reference's own order of work, detect.py:231 _applyFilter per chunk)
    cfr = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'], raw=True)
    for i, sta in enumerate(stas):
        cfr.add(sta, cf.data[sta][0], starttime=1.5e9)
    cfr.pin()
    ssr = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cfr)
    ssr.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
    ssr.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
    resr = ssr.detex(subspaceDB=None, estimateMags=True, device=local)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        resr = ssr.detex(subspaceDB=None, estimateMags=True, device=local)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out['raw_fetcher'] = {'value': windows / dt, 'ms_per_call': 1e3 * dt, 'ms_per_station_day': 1e3 * dt / NSTA,
                          'detections': int(len(resr['ss_df'])),
                          'note': 'per-chunk detrend + zero-phase band-pass on the device inside the timed region'}
    got = res['ss_df']
    out['planted'] = nplant * NSTA
    out['detections_with_magnitude'] = int(got.Mag.notna().sum())
    assert len(got) >= nplant * NSTA * 0.9, 'e2e_detex: %d detections for %d planted events' % (len(got), nplant * NSTA)
    out['value'] = out['no_db']['value']
    return out


def e2e_detex_240(local):
    """BASELINE configs[3]'s volume through the PRODUCT API on one GPU: 8 stations x 30 days = 240 station-days
    (720 hourly chunks of 3720 s per station) through SubSpace.detex(estimateMags=True) -- without a DB, with
    the SQLite file, and with an unfiltered fetcher (per-chunk detrend + band-pass on the device).  A station's
    record is scanned in day units (SubSpace._scan_units), three units in flight.  Host memory stays bounded:
    the eight stations share ONE page-locked 30-day record (3.1 GB; one day of noise + 50 planted events,
    repeated), so every station-day still crosses PCIe on its own."""
    import tempfile
    from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace, synth
    SR, NC, L, NFAM, NPER, NSTA, NDAY = 100.0, 3, 3300, 25, 8, 8, 30
    rs = np.random.RandomState(0)
    td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
    stas = ['TA.S%02dA' % i for i in range(NSTA)]
    ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    for sta in stas:
        for e in range(len(td)):
            ef.add(sta, 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e,
                   origintime=990.0 + 100 * e, magnitude=1.0 + (e % 5) * .2)
    T1 = int(24 * 3600 * SR)
    day = rs.standard_normal((NC, T1)).astype(np.float32)
    nplant = 50
    for k in range(nplant):
        e = int(rs.randint(len(td)))
        t = int(rs.randint(L, T1 - 2 * L))
        day[:, t:t + L] += 3.0 * td[e].reshape(L, NC).T / np.std(td[e])
    rec = torch.empty((NC, NDAY * T1), dtype=torch.float32).pin_memory()
    for d in range(NDAY):
        rec[:, d * T1:(d + 1) * T1] = torch.from_numpy(day)
    t_start = 1499997600.0                                   # on the hour: 720 chunks, the last one 3600 s

    def fetcher(raw):
        cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'], raw=raw)
        for sta in stas:
            cf.add(sta, rec.numpy(), starttime=t_start)
        cf.pinned = {sta: rec for sta in stas}               # (what ContinuousFetcher.pin() sets up, shared)
        return cf

    cl = createCluster(CCreq=0.5, fetch_arg=ef, device=local, saveclust=False)
    out = {'workload': '%d stations x %d days = %d station-days (720 chunks of 3720 s per station, 3-ch 100 Hz) through '
                       'SubSpace.detex(estimateMags=True), day units, one shared pinned 30-day record'
                       % (NSTA, NDAY, NSTA * NDAY), 'unit': 'windows/s', 'planted': nplant * NDAY * NSTA}
    with tempfile.TemporaryDirectory() as tmp:
        for key, raw, db in (('no_db', False, None), ('sqlite', False, os.path.join(tmp, 'SubSpace.db')),
                             ('raw_fetcher', True, None)):
            cf = fetcher(raw)
            ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
            ss.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
            ss.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
            names, kinds, U, off, _, _ = ss._detector_bank(stas[0], useSingles=False)
            n = U.shape[1]
            windows = NSTA * sum((int(b[1]) * NC - n) // NC + 1 for b in cf.chunk_bounds(stas[0]))
            for _ in range(3):      # warm: contexts, device buffers, torch's pools of pinned result buffers (they settle over two to three calls)
                res = ss.detex(subspaceDB=db, estimateMags=True, device=local)
            torch.cuda.synchronize()
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                res = ss.detex(subspaceDB=db, estimateMags=True, device=local)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            # the median of the three calls: the host is shared with the pod's other GPU jobs, and a call whose
            # Python side lost its cores for a moment shows up as an outlier (all three calls are listed)
            dt = sorted(times)[len(times) // 2]
            out[key] = {'value': windows / dt, 's_per_call': dt, 'ms_per_station_day': 1e3 * dt / (NSTA * NDAY),
                        'ms_per_station_day_calls': [1e3 * t / (NSTA * NDAY) for t in times],
                        'ms_per_station_day_mean': 1e3 * sum(times) / len(times) / (NSTA * NDAY),
                        'detections': int(len(res['ss_df'])), 'units': NSTA * NDAY,
                        'uploaded_gb': ss.last_run['uploaded_bytes'] / 1e9}
            assert len(res['ss_df']) >= 0.9 * nplant * NDAY * NSTA, (key, len(res['ss_df']))
            del ss, cf, res
            torch.cuda.empty_cache()
    out['value'] = out['no_db']['value']
    return out


def run_config4(a, rank, world, local, use_dist, binding=None):
    """BASELINE configs[3]: `stations` x `days` station-days of the configs[1] workload, dealt
    round-robin over the GPUs (detex_amd.parallel.shard_units); every station has its own
    detector basis, built on rank 0 and broadcast over RCCL / xGMI.  One step = every rank scans
    all its station-days (strong scaling: the total is fixed as N grows).  Inputs (noise drawn on
    the device per (station, day) seed + planted events) are resident in HBM before the timed
    region; each station's basis images are uploaded once, before it."""
    import torch.distributed as dist
    from detex_amd import parallel, synth
    from detex_amd.hip import HipContext
    binding = binding if binding is not None else dict(parallel.BINDING)
    dev = torch.device('cuda', local)
    Nc = a.nc
    L = int(round(a.template_seconds * a.sr))
    n = Nc * L
    M = int(round(a.chunk_seconds * a.sr)) * Nc
    D, k = a.detectors, a.basis_per_detector
    units = [(s, d) for s in range(a.stations) for d in range(a.days)]
    mine = [units[i] for i in parallel.shard_units(len(units), rank, world)]
    ctxs, bases = {}, {}
    t_up = time.perf_counter()
    for s in range(a.stations):
        U, off = synth.detector_bank(100 + s, D, k, n) if rank == 0 else (None, None)
        Ut, off = parallel.broadcast_basis(U, off, n, dev)          # every rank takes part
        if any(u[0] == s for u in mine):
            c = HipContext(local)
            c.set_engine(a.engine)
            c.upload_basis(Ut, off, Nc)
            ctxs[s], bases[s] = c, (Ut.cpu().numpy(), off)
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t_up
    gen = torch.Generator(device=dev)
    data, plants = {}, {}
    for (s, d) in mine:
        gen.manual_seed(1000 * s + d)                               # SURVEY 8d cfg 4
        x = torch.randn((a.chunks, M), generator=gen, device=dev, dtype=torch.float32)
        rs = np.random.RandomState(1000 * s + d)
        Uh, off = bases[s]
        pl = []
        for _ in range(5):
            b, dd = int(rs.randint(a.chunks)), int(rs.randint(D))
            while True:
                p = int(rs.randint(0, (M - n) // Nc))
                if all(pb != b or abs(pp - p) >= 2 * L for (pb, pp, _) in pl):
                    break
            coef = rs.standard_normal(k)
            coef /= np.linalg.norm(coef)
            w = (rs.uniform(5, 20) * np.sqrt(n) * (coef @ Uh[off[dd]:off[dd + 1]])).astype(np.float32)
            x[b, p * Nc:p * Nc + n] += torch.from_numpy(w).to(dev)
            pl.append((b, p, dd))
        data[(s, d)], plants[(s, d)] = x, pl
    P = next(iter(ctxs.values())).num_windows(M) if ctxs else (M - n) // Nc + 1
    ldp = (P + 63) // 64 * 64
    ds = torch.empty((a.chunks, D, ldp), dtype=torch.float32, device=dev)[:, :, :P]
    thr = np.full(D, 0.3, dtype=np.float32)
    hists = {s: torch.zeros((D, 400), dtype=torch.int32, device=dev) for s in ctxs}

    found = [0, 0]
    digest = [0, 0]                                # candidates of the checked step: count, sum of window indices

    thr_dev = torch.from_numpy(thr).to(dev)       # (a host array would be copied up, synchronously, by every scan)

    def step(check=False):
        """Every station-day of this rank: scan enqueued, its results (maxima, candidates) fetched while the
        next one runs -- all of them, inside the timed region."""
        prev = None
        for (s, d) in mine:
            _, f = ctxs[s].subspace_scan(data[(s, d)], out=ds, thr=thr_dev, hist=hists[s], lazy=True)
            if check:
                cb, cd, cp, cv = f()['cand']
                digest[0] += int(len(cp))
                digest[1] += int(cp.sum())
                for (b, p, dd) in plants[(s, d)]:
                    assert ((cb == b) & (cd == dd) & (np.abs(cp - p) <= 2)).any(), (s, d, b, p, dd)
                    found[0] += 1
                found[1] += len(plants[(s, d)])
            else:
                if prev is not None:
                    prev()
                prev = f
        if prev is not None:
            prev()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step(check=True)
    for _ in range(max(a.warmup - 1, 0)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    windows = len(units) * a.chunks * P * a.steps
    host_fed = None
    if (use_dist and world > 1) or os.environ.get('DTX_BENCH_HOST_FED') == '1':      # (the switch: the leg at world size 1)
        # the same pass with every station-day coming from page-locked host memory, on every rank at once: up to eight
        # distinct pinned station-days per rank, cycled (more than a socket's last-level cache), two device buffers,
        # the next upload on a copy stream under the current scan
        def leg(bar):
            pool = [data[u].cpu().pin_memory() for u in mine[:8]]
            bufs = [torch.empty_like(data[mine[0]]) for _ in range(2)] if mine else []
            copy = torch.cuda.Stream()
            ready = [torch.cuda.Event() for _ in range(2)]
            freed = [torch.cuda.Event() for _ in range(2)]
            main_s = torch.cuda.current_stream()
            for e in freed:
                e.record(main_s)

            def upload(i, j):
                with torch.cuda.stream(copy):
                    copy.wait_event(freed[j])
                    bufs[j].copy_(pool[i % len(pool)], non_blocking=True)
                    ready[j].record(copy)

            def one_pass():
                prev = None
                if mine:
                    upload(0, 0)
                for i, (s, d) in enumerate(mine):
                    j = i & 1
                    if i + 1 < len(mine):
                        upload(i + 1, 1 - j)
                    main_s.wait_event(ready[j])
                    _, f = ctxs[s].subspace_scan(bufs[j], out=ds, thr=thr_dev, hist=hists[s], lazy=True)
                    freed[j].record(main_s)
                    if prev is not None:
                        prev()
                    prev = f
                if prev is not None:
                    prev()
            one_pass()
            torch.cuda.synchronize()
            bar()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                one_pass()
            torch.cuda.synchronize()
            dth = time.perf_counter() - t1
            return {'ms_per_step': 1e3 * dth / a.steps, 'steps': a.steps,
                    'kernel_ms': None}
        if use_dist:
            host_fed = host_fed_all_ranks(leg, rank, world, len(units) * a.chunks * P / world, binding)
        else:
            r = leg(lambda: None)
            host_fed = {'ms_per_step_min': r['ms_per_step'], 'ms_per_step_max': r['ms_per_step'],
                        'ms_per_step_mean': r['ms_per_step'], 'steps': r['steps'],
                        'value': len(units) * a.chunks * P / (r['ms_per_step'] * 1e-3), 'unit': 'windows/s'}
        host_fed['ms_per_station_day_slowest_rank'] = host_fed['ms_per_step_max'] / max(1, -(-len(units) // world))
    if use_dist:
        t = torch.tensor(found + digest, dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        found, digest = [int(v) for v in t.tolist()[:2]], [int(v) for v in t.tolist()[2:]]
    for c in ctxs.values():
        c.close()
    return {'metric': 'detection windows/sec (24h 3-ch 100Hz, 200 templates)', 'value': windows / dt,
            'unit': 'windows/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32 (reference default f64)', 'data': 'synthetic',
            'config': {'workload': 'configs[3]: %d stations x %d days = %d station-days of %d basis vectors '
                                   '(%d x %d) x %d chunks of %g s, %d-ch %g Hz, dealt round-robin over %d GPUs'
                                   % (a.stations, a.days, len(units), D * k, D, k, a.chunks, a.chunk_seconds,
                                      Nc, a.sr, world),
                       'engine': a.engine, 'station_days_per_gpu_max': -(-len(units) // world),
                       'planted_found': '%d/%d' % tuple(found),
                       'candidates': {'n': digest[0], 'index_sum': digest[1]},
                       'resident_stream_gb_per_gpu': sum(v.numel() for v in data.values()) * 4 / 1e9,
                       'basis_broadcast_and_upload_s': t_up,
                       'host_fed_ms_per_step': None if host_fed is None else
                       [host_fed['ms_per_step_min'], host_fed['ms_per_step_mean'], host_fed['ms_per_step_max']]},
            'host_fed': host_fed}


def main():
    a = parse()
    if a.gpus > 1 and 'RANK' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as a child job (one process per
        # GPU, RCCL rendezvous on the loopback address) before anything here touches a GPU
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio when its first
    # communicator comes up (flushed at exit, i.e. behind the line), and other libraries may print too: from
    # here on file descriptor 1 is stderr; the line goes out through the saved descriptor (emit).
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(out_fd, (json.dumps(obj) + '\n').encode())
    # (tests on a one-GPU box: DTX_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and DTX_BENCH_BACKEND=gloo
    # carries the collectives -- the N > 1 code path without a second GPU; RCCL refuses two ranks per device)
    if os.environ.get('DTX_BENCH_ONE_DEVICE') == '1':
        local = 0
    backend = os.environ.get('DTX_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local)
    use_dist = 'RANK' in os.environ          # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    from detex_amd.hip import HipContext
    from detex_amd import synth
    from detex_amd import parallel as _par
    # one process per GPU: next to its GPU's NUMA node before anything page-locked is allocated
    binding = _par.bind_to_device(local)

    Nc = a.nc
    L = int(round(a.template_seconds * a.sr))
    n = Nc * L
    M = int(round(a.chunk_seconds * a.sr)) * Nc
    D, k = a.detectors, a.basis_per_detector
    K = D * k
    if a.config == 4:
        out = run_config4(a, rank, world, local, use_dist, binding)
        out['config']['binding'] = dict(binding)
        if rank == 0:
            emit(out)
        if use_dist:
            dist.destroy_process_group()
        return
    ctx = HipContext(local)
    ctx.set_engine(a.engine)

    # detector basis: built on rank 0, broadcast over RCCL/xGMI (the only collective)
    from detex_amd import parallel
    if rank == 0:
        U, det_off = synth.detector_bank(1, D, k, n)
    else:
        U, det_off = None, None
    Ut, det_off = parallel.broadcast_basis(U, det_off, n, torch.device('cuda', local))
    ctx.upload_basis(Ut, det_off, Nc)
    Uh = Ut.cpu().numpy()

    # one station-day per rank (seed = 1000*station + day, SURVEY.md 8d cfg 4)
    x, plants = synth.continuous_day(1000 * rank + 1, a.chunks, M, Nc, Uh, det_off, nplant=50)
    xt = torch.from_numpy(x).cuda()
    P = ctx.num_windows(M)
    ldp = (P + 63) // 64 * 64              # aligned row pitch
    ds = torch.empty((a.chunks, D, ldp), dtype=torch.float32, device='cuda')[:, :, :P]
    thr = np.full(D, 0.3, dtype=np.float32)
    hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')

    thr_dev = torch.from_numpy(thr).cuda()    # resident like the stream (a host array would be copied up by every scan)

    def step():
        # detection statistic + histogram / max / threshold candidates (dtx_subspace_scan), everything
        # enqueued; fetch() waits for and returns the step's small results (maxima, candidate list)
        return ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The clocks of an idle GPU take time under load to settle, and how long differs from box to box (round 4's
    # driver run: 2.88 ms per kernel after a fixed half second, 2.72 on the builder's boxes; 20 steps are 70 ms).
    # A criterion instead of a constant: repeat the same step, untimed, until the scan kernel's duration (HIP
    # events around it) has stopped moving -- mean of the last 10 launches within 1 % of the 10 before -- for
    # at least --prewarm-min and at most --prewarm seconds; what was spent is in the line (config.prewarm).
    t_pre = time.perf_counter()
    pre_k, settled = [], False
    while a.prewarm > 0:
        step()()
        pre_k.append(ctx.last_kernel_ms())
        el = time.perf_counter() - t_pre
        if len(pre_k) >= 20 and el >= min(a.prewarm_min, a.prewarm):
            m1, m0 = sum(pre_k[-10:]) / 10.0, sum(pre_k[-20:-10]) / 10.0
            settled = abs(m1 - m0) <= 0.01 * m0
        if settled or el >= a.prewarm:
            break
    prewarm = {'seconds': time.perf_counter() - t_pre, 'steps': len(pre_k), 'settled': settled,
               'cap_s': a.prewarm, 'min_s': a.prewarm_min,
               'kernel_ms_first10': (sum(pre_k[:10]) / len(pre_k[:10])) if pre_k else None,
               'kernel_ms_last10': (sum(pre_k[-10:]) / len(pre_k[-10:])) if pre_k else None}
    # the W warm-up steps run the way the timed ones do -- one step's results fetched while the next step runs -- so
    # that everything the pipelined loop allocates exists before the clock starts: with two steps in flight torch's
    # caching allocators hand out a SECOND set of result tensors (12 MB on the device, the pinned host buffers behind
    # them), and a first-time hipMalloc / hipHostMalloc inside the timed region cost one of its 20 kernels a
    # millisecond (kernel_ms_each showed 3.7 ms among 2.8s; the 200-step leg behind it never did)
    pend = None
    for _ in range(a.warmup):
        f = step()
        if pend is not None:
            res = pend()
        pend = f
    if pend is not None:
        res = pend()

    def timed_loop(steps, barrier):
        # K steps, each one's results fetched inside the timed region -- while the next step runs, as
        # SubSpace.detex does with its stations (the copies to pinned host memory go on a side stream
        # behind the step's post pass; nothing is skipped, the GPU just does not wait for the host)
        barrier()
        t0 = time.perf_counter()
        pending = res = None
        stamps.clear()
        for _ in range(steps):
            fetch = step()
            if pending is not None:
                res = pending()
                stamps.append(time.perf_counter())
            pending = fetch
        res = pending()
        stamps.append(time.perf_counter())
        barrier()
        return time.perf_counter() - t0, res

    def spread(steps):
        """Per-step spread of the loop that just ran: the scan kernel's durations one by one (HIP events, the
        last <= 64 launches) and the host's step-to-step intervals (a step's results arrive while the next one
        runs, so an interval is one step of the pipeline)."""
        each = ctx.kernel_ms_list(min(steps, 64))
        ks = sorted(each)
        iv = sorted(1e3 * (b - a_) for a_, b in zip(stamps[:-1], stamps[1:]))
        o = {'kernel_ms_min': ks[0], 'kernel_ms_median': ks[len(ks) // 2], 'kernel_ms_max': ks[-1],
             'kernel_ms_launches': len(ks)}
        if steps <= 64:
            o['kernel_ms_each'] = [round(v, 3) for v in each]      # in launch order: WHICH step was slow
        if iv:
            o.update({'step_ms_min': iv[0], 'step_ms_median': iv[len(iv) // 2], 'step_ms_max': iv[-1]})
        return o

    stamps = []
    dt, res = timed_loop(a.steps, barrier)
    step_spread = spread(a.steps)
    # dominant-kernel time: HIP events the library records around the scan kernel of every call on the
    # context's stream (a ring of 64 pairs), read here, after the timed region: average launch duration
    # over its (last 64) steps
    kms = ctx.kernel_ms_mean(min(a.steps, 64))
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    windows = a.chunks * P * world * a.steps
    value = windows / dt

    # sanity: every planted event is a candidate of its own detector
    cb, cd, cp, cv = res['cand']
    found = 0
    for (b, p, d) in plants:
        sel = (cb == b) & (cd == d) & (np.abs(cp - p) <= 2)
        found += bool(sel.any())
    if os.environ.get('DTX_BENCH_TIMING_ONLY') != '1':      # (set for timing-only ablation builds of the kernel)
        assert found == len(plants), 'planted events missed: %d of %d found' % (found, len(plants))

    flops = 2.0 * K * n * P * a.chunks       # the contraction's algorithmic flop
    default_cfg = (D, k, a.chunks, Nc, a.sr, a.chunk_seconds, a.template_seconds) == \
        (25, 8, 24, 3, 100.0, 3720.0, 30.0)
    eng = a.engine
    if eng == 'auto':                        # dtx_internal.h: kFftAutoMinL .. kFftLongMaxL
        eng = 'fft' if 256 <= L <= 32768 else 'bf16x3'
    if eng == 'bf16x3' and L < 64:
        eng = 'fp32'
    out = {
        'metric': 'detection windows/sec (24h 3-ch 100Hz, 200 templates)',
        'value': value, 'unit': 'windows/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        # the reference defaults to float64 (construct.py:27-34 dtype='double', 'single' = float32);
        # north_star prescribes float32 within 1e-5 of that path
        'dtype': {'fft': 'f32 (complex64 transforms, f64 window statistics; reference default f64)',
                  'fp32': 'f32 (reference default f64)',
                  'bf16x3': 'bf16x3 (f32 operands split in 2 bf16, f32 accumulate; reference default f64)'}[eng],
        'data': 'synthetic',
        'config': {'workload': '%s%d basis vectors (%d detectors x %d) x %d chunks of '
                               '%g s, %d-ch %g Hz, n=%d, one station-day per GPU'
                               % ('configs[1]: ' if default_cfg else 'custom: ', K, D, k, a.chunks,
                                  a.chunk_seconds, Nc, a.sr, n),
                   'engine': eng if a.engine == eng else '%s -> %s' % (a.engine, eng),
                   'prewarm_s': prewarm['seconds'], 'prewarm': prewarm,
                   **step_spread, 'windows_per_step_per_gpu': a.chunks * P, 'planted_found': '%d/%d' % (found, len(plants)),
                   # digest of the last step's candidate list: equal between launch modes (plain / torchrun) on one box
                   'candidates': {'n': int(len(cp)), 'index_sum': int(cp.sum()), 'value_sum': float(np.sum(cv, dtype=np.float64))},
                   'results': 'maxima and candidate lists of every step fetched to the host inside the timed '
                              'region, each while the next step runs (pinned buffers, side stream)',
                   'binding': dict(binding)},
        'roofline': roofline(eng, kms, K, n, Nc, L, D, P, a.chunks, flops, default_cfg, kernel_name=ctx.last_kernel_name()),
    }
    tr = out['roofline'].get('traffic') if out['roofline'] else None
    if tr and tr.get('bytes') and out['roofline'].get('bytes_per_launch'):
        # flat copies of the nested `traffic` object (a consumer that keeps only scalars keeps these)
        out['roofline']['traffic_bytes'] = tr['bytes']
        out['roofline']['traffic_ratio'] = tr['bytes'] / out['roofline']['bytes_per_launch']
    if use_dist and (world > 1 or os.environ.get('DTX_BENCH_HOST_FED') == '1'):
        # N > 1 (or asked for at world size 1: the RCCL test of the leg): the same step host-fed, on every rank at once (HBM-resident data scale trivially; the host feed is shared)
        hf_steps = max(20, min(a.steps, 80))
        hf = host_fed_all_ranks(lambda bar: pcie_inclusive(ctx, x, ds, thr, hist, a.chunks * P, hf_steps, barrier=bar),
                                rank, world, a.chunks * P, binding)
        out['host_fed'] = hf
        out['config']['host_fed_ms_per_step'] = [hf['ms_per_step_min'], hf['ms_per_step_mean'], hf['ms_per_step_max']]
    extras = rank == 0 and world == 1 and default_cfg and not a.no_extras
    if extras:
        # the same loop again, 200 steps, right behind the timed region: what this box sustains once nothing is
        # cold.  The headline within a few per cent of this = the K timed steps were not taken on a cold lease.
        sdt, _ = timed_loop(200, barrier)
        sk = ctx.kernel_ms_mean(64)
        out['steady'] = {'value': a.chunks * P * 200 / sdt, 'unit': 'windows/s', 'steps': 200,
                         'ms_per_step': 1e3 * sdt / 200, 'kernel_ms': sk, **spread(200),
                         'headline_over_steady': value / (a.chunks * P * 200 / sdt)}
        try:
            out['config2_dirty'] = config2_dirty(ctx, synth, a, x, plants, ds, thr_dev, hist, Nc, n, L, D, P,
                                                 out['steady']['ms_per_step'])
        except Exception as e:                         # noqa: BLE001 -- a secondary field never takes the line down
            out['config2_dirty'] = {'value': None, 'error': repr(e)}
        # the other engines on the same input (same step: scan + post pass)
        for other, key in (('bf16x3', 'mfma_engine'), ('fp32', 'fp32_engine'), ('fft', 'fft_engine')):
            if other == eng:
                continue
            ctx.set_engine(other)
            step()()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()()
            torch.cuda.synchronize()
            dto = time.perf_counter() - t1
            ko = ctx.last_kernel_ms()
            out[key] = {'value': a.chunks * P / dto, 'unit': 'windows/s',
                        'roofline': roofline(other, ko, K, n, Nc, L, D, P, a.chunks, flops, default_cfg,
                                             kernel_name=ctx.last_kernel_name())}
        ctx.set_engine(a.engine)
        try:
            out['dist_overhead'] = dist_overhead(local, timed_loop, min(a.steps, 100), 1e3 * dt / a.steps,
                                                 Uh, det_off, n)
        except Exception as e:                         # noqa: BLE001 -- a secondary field never takes the line down
            out['dist_overhead'] = {'error': repr(e)}
        out['pcie_inclusive'] = pcie_inclusive(ctx, x, ds, thr, hist, a.chunks * P, max(40, min(a.steps, 80)))
        out['xcorr_full'] = xcorr_full(ctx, synth, Nc, cpu=not a.no_cpu_baseline)
        out['config5'] = config5(ctx, synth)
        try:
            del ds
            torch.cuda.empty_cache()
            out['config2_singles'] = config2_singles(ctx, synth, a, xt, M, Nc, n)
        except Exception as e:                         # noqa: BLE001 -- a secondary field never takes the line down
            out['config2_singles'] = {'value': None, 'error': repr(e)}
        torch.cuda.empty_cache()
        ds = None
        ctx.upload_basis(Ut, det_off, Nc)
        # BASELINE configs[3] at its real size on this one GPU: the N = 1 anchor of the strong-scaling curve
        # (the driver measures N = 2, 4, 8 with --config 4 --gpus N)
        try:
            import copy
            a4 = copy.copy(a)
            a4.config, a4.steps, a4.warmup, a4.stations, a4.days = 4, 2, 1, 8, 30
            torch.cuda.empty_cache()
            c4 = run_config4(a4, 0, 1, local, False)
            out['config4_n1'] = {'value': c4['value'], 'unit': c4['unit'], 'ms_per_step': c4['ms_per_step'],
                                 'steps': a4.steps, 'scaling': 'strong', 'n_gpus': 1,
                                 'ms_per_station_day': c4['ms_per_step'] / 240.0, 'config': c4['config']}
        except Exception as e:                         # noqa: BLE001 -- a secondary field never takes the line down
            out['config4_n1'] = {'value': None, 'error': repr(e)}
        torch.cuda.empty_cache()
        try:
            out['e2e_createCluster_2000'] = e2e_createcluster(local, cpu=not a.no_cpu_baseline)
        except Exception as e:                         # noqa: BLE001
            out['e2e_createCluster_2000'] = {'value': None, 'error': repr(e)}
        try:
            out['e2e_fas'] = e2e_fas(local)
        except Exception as e:                         # noqa: BLE001
            out['e2e_fas'] = {'value': None, 'error': repr(e)}
        torch.cuda.empty_cache()
        try:
            out['e2e_detex'] = e2e_detex(local)
        except Exception as e:                         # noqa: BLE001
            out['e2e_detex'] = {'value': None, 'error': repr(e)}
        torch.cuda.empty_cache()
        try:
            out['e2e_detex_240'] = e2e_detex_240(local)
        except Exception as e:                         # noqa: BLE001
            out['e2e_detex_240'] = {'value': None, 'error': repr(e)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(Uh, det_off, x[0], Nc, a.cpu_vectors)
        if default_cfg and not a.no_extras:
            out['cpu_baseline']['all_cores'] = cpu_baseline_allcores(Uh, det_off, x, Nc,
                                                                     onecore_rate=out['cpu_baseline']['value'])
            # the same reference arithmetic prices the singles variant and the FAS (one IFFT per vector and chunk)
            rate = out['cpu_baseline']['value']
            if isinstance(out.get('config2_singles'), dict) and out['config2_singles'].get('value'):
                out['config2_singles']['cpu_baseline'] = {
                    'value': rate, 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
                    'sample': 'the configs[1] sample: the same 200 vectors, one IFFT per vector and chunk however they '
                              'are grouped into detectors (detect.py:559-578)'}
            if isinstance(out.get('e2e_fas'), dict) and out['e2e_fas'].get('value'):
                out['e2e_fas']['cpu_baseline'] = {
                    'value': rate, 'unit': 'windows/s', 'cores': 1, 'kind': 'port',
                    'sample': 'the configs[1] sample: fas._MPXSSCorr is the bit-equal twin of _MPXDS (fas.py:120-134), '
                              '200 vectors per chunk here too'}
    else:
        out['cpu_baseline'] = None
    if extras:
        # the scalars a reader needs, flat in `config` (a consumer that keeps `config` and the headline keeps these)
        def pick(key, *path):
            v = out.get(key)
            for q in path:
                v = v.get(q) if isinstance(v, dict) else None
            return v
        e240 = out.get('e2e_detex_240') if isinstance(out.get('e2e_detex_240'), dict) else {}
        out['config'].update({
            'steady_ms_per_step': pick('steady', 'ms_per_step'),
            'headline_over_steady': pick('steady', 'headline_over_steady'),
            'pcie_ms_per_step': pick('pcie_inclusive', 'ms_per_step'),
            'e2e_240_ms_per_station_day': [(e240.get(q) or {}).get('ms_per_station_day')
                                           for q in ('no_db', 'sqlite', 'raw_fetcher')],
            'xcorr_full_s': pick('xcorr_full', 'seconds'),
            'singles_ms_per_step': pick('config2_singles', 'ms_per_step'),
            'dirty_ms_per_step': pick('config2_dirty', 'ms_per_step'),
            'dirty_over_clean': pick('config2_dirty', 'over_clean'),
            'glitches_1000_over_clean': pick('config2_dirty', 'glitches_1000', 'over_clean'),
        })
    if rank == 0:
        emit(out)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

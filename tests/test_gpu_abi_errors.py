"""Error behaviour of the C ABI on a real device: bad arguments and call-order
mistakes return an error code with a message (the reference's convention is
detex.log(level='error') -> raise, detex/__init__.py:131-138); nothing crashes."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture()
def raw():
    from detex_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.dtx_ctx_create(0, None, 1, C.byref(h)) == 0      # own non-blocking stream
    yield lib, h
    assert lib.dtx_ctx_destroy(h) == 0


def _msg(lib, h):
    return lib.dtx_last_error(h).decode()


def test_call_order_and_arguments(raw):
    lib, h = raw
    x = torch.zeros(1000, device='cuda')
    out = torch.zeros(1000, device='cuda')
    rc = lib.dtx_subspace_ds(h, C.c_void_p(x.data_ptr()), 1000, 1000, 1, C.c_void_p(out.data_ptr()), 1000)
    assert rc == -3 and 'no basis' in _msg(lib, h)                # DTX_E_STATE
    U = np.zeros((3, 30), np.float32)
    off = np.array([0, 3], np.int32)
    up = lambda n, Nc, o=off, D=1: lib.dtx_basis_upload(  # noqa: E731
        h, U.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), D, n, Nc)
    assert up(30, 4) == -1 and 'multiple of Nc' in _msg(lib, h)
    assert up(30, 0) == -1
    assert up(30, 3, np.array([1, 3], np.int32)) == -1 and 'det_off[0]' in _msg(lib, h)
    assert up(30, 3, np.array([0, 0], np.int32)) == -1
    assert lib.dtx_basis_upload(h, None, off.ctypes.data_as(C.c_void_p), 1, 30, 3) == -1
    assert up(30, 3) == 0
    # block not longer than the template (detect.py:262-266), ldp too small, null pointers
    rc = lib.dtx_subspace_ds(h, C.c_void_p(x.data_ptr()), 30, 30, 1, C.c_void_p(out.data_ptr()), 1000)
    assert rc == -1 and 'not longer than the basis' in _msg(lib, h)
    rc = lib.dtx_subspace_ds(h, C.c_void_p(x.data_ptr()), 300, 300, 1, C.c_void_p(out.data_ptr()), 10)
    assert rc == -1 and 'ldp' in _msg(lib, h)
    assert lib.dtx_subspace_ds(h, None, 300, 300, 1, C.c_void_p(out.data_ptr()), 1000) == -1
    assert lib.dtx_ds_post(h, None, 10, 10, 1, 1, None, None, None, None, None, 0, None) == -1
    assert lib.dtx_ds_moments(h, C.c_void_p(out.data_ptr()), 10, 10, 1, 1, 1.0, 0.0, 10, None, None) == -1
    rc = lib.dtx_xcorr_allpairs(h, C.c_void_p(x.data_ptr()), 1, 30, 3, C.c_void_p(out.data_ptr()),
                                C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()), 0, 0)
    assert rc == -1 and 'N >= 2' in _msg(lib, h)
    rc = lib.dtx_xcorr_allpairs(h, C.c_void_p(x.data_ptr()), 4, 30, 3, C.c_void_p(out.data_ptr()),
                                C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()), 0, 7)
    assert rc == -1 and 'pair range' in _msg(lib, h)
    assert lib.dtx_set_engine(h, 7) == -1 and 'unknown engine' in _msg(lib, h)
    assert all(lib.dtx_set_engine(h, e) == 0 for e in (0, 1, 2, 3))
    ms = C.c_float()
    assert lib.dtx_last_kernel_ms(h, C.byref(ms)) == -3
    # a good call still works afterwards, on the context's own stream
    assert lib.dtx_subspace_ds(h, C.c_void_p(x.data_ptr()), 300, 300, 1, C.c_void_p(out.data_ptr()), 1000) == 0
    assert lib.dtx_ctx_sync(h) == 0
    assert lib.dtx_last_kernel_ms(h, C.byref(ms)) == 0 and ms.value >= 0
    assert lib.dtx_timer_start(h) == 0 and lib.dtx_timer_stop(h, C.byref(ms)) == 0


def test_host_entry_points(raw):
    """dtx_subspace_ds_host / dtx_xcorr_allpairs_host: numpy in, numpy out."""
    from oracle import detex_oracle as orc
    lib, h = raw
    rs = np.random.RandomState(0)
    Nc, L = 3, 40
    U = rs.standard_normal((2, Nc * L))
    U = (U / np.linalg.norm(U, axis=1, keepdims=True)).astype(np.float32)
    off = np.array([0, 2], np.int32)
    assert lib.dtx_basis_upload(h, U.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), 1,
                                Nc * L, Nc) == 0
    x = rs.standard_normal(Nc * 500).astype(np.float32)
    P = lib.dtx_num_windows(len(x), Nc * L, Nc)
    ds = np.empty((1, P), np.float32)
    assert lib.dtx_subspace_ds_host(h, x.ctypes.data_as(C.c_void_p), len(x),
                                    ds.ctypes.data_as(C.c_void_p)) == 0
    assert np.max(np.abs(ds[0] - orc.mpxds_fft(x, U, Nc))) <= 1e-5
    td = rs.standard_normal((5, Nc * L)).astype(np.float32)
    cc = np.empty(10, np.float32)
    lag = np.empty(10, np.int32)
    ss = np.empty(10, np.float32)
    assert lib.dtx_xcorr_allpairs_host(h, td.ctypes.data_as(C.c_void_p), 5, Nc * L, Nc,
                                       cc.ctypes.data_as(C.c_void_p), lag.ctypes.data_as(C.c_void_p),
                                       ss.ctypes.data_as(C.c_void_p)) == 0
    rcc, rlag, rss = orc.make_cclags(list(td.astype(np.float64)), Nc)
    iu = np.triu_indices(5, 1)
    assert np.max(np.abs(cc - rcc[iu[0], iu[1] - 1])) <= 1e-5
    assert np.array_equal(lag, rlag[iu[0], iu[1] - 1].astype(np.int32))


def test_preprocess_planes_arguments(raw):
    """dtx_preprocess_planes: the channel stride must hold a chunk, the chunk stride may be anything >= 0 (chunks
    overlap in the record they are cut from), the output rows must hold Nc T samples."""
    lib, h = raw
    rec = torch.randn(3 * 5000, device='cuda')
    out = torch.zeros(4 * 3 * 1000, device='cuda')
    pp = lambda T, cs, hs, mps, Nc=3, nch=4: lib.dtx_preprocess_planes(  # noqa: E731
        h, C.c_void_p(rec.data_ptr()), nch, Nc, T, cs, hs, None, 0, 0, 0, C.c_void_p(out.data_ptr()), mps)
    assert pp(1000, 900, 999, 3000) == -1 and 'dtx_preprocess_planes' in _msg(lib, h)      # channels would overlap
    assert pp(1000, -1, 5000, 3000) == -1
    assert pp(1000, 900, 5000, 2999) == -1                                                  # output rows too short
    assert pp(0, 900, 5000, 3000) == -1
    assert lib.dtx_preprocess_planes(h, None, 4, 3, 1000, 900, 5000, None, 0, 0, 0, C.c_void_p(out.data_ptr()), 3000) == -1
    assert pp(1000, 900, 5000, 3000) == 0 and pp(1000, 0, 5000, 3000) == 0
    assert pp(1000, 1000, 0, 1000, Nc=1) == 0                                               # one channel: no channel stride
    assert lib.dtx_ctx_sync(h) == 0
    # (the last good call: one channel, chunks of 1000 samples 1000 apart -> the record's first 4000 samples)
    assert torch.equal(out[:4000], rec[:4000])


def test_wrapper_rejects_misshapen_outputs():
    """A caller-supplied DS / histogram / sums tensor of the wrong shape, dtype or pitch is refused
    before the kernels would write through it."""
    import numpy as np
    import torch
    from detex_amd.hip import HipContext
    ctx = HipContext(0)
    U = np.random.RandomState(0).standard_normal((3, 90)).astype(np.float32)
    ctx.upload_basis(U, [0, 2, 3], 3)
    x = torch.randn(2, 3000, device='cuda')
    P = ctx.num_windows(3000)
    good = torch.empty((2, 2, P + 5), device='cuda')[:, :, :P]
    ctx.subspace_ds(x, out=good)
    for bad in (torch.empty((2, 3, P), device='cuda'), torch.empty((2, 2, P - 1), device='cuda'),
                torch.empty((2, 2, P), device='cuda', dtype=torch.float64),
                torch.empty((2, 2, 2 * P), device='cuda')[:, :, ::2],
                torch.empty((1, 2, P), device='cuda'), torch.empty((2, 2, P))):
        with pytest.raises(ValueError):
            ctx.subspace_ds(x, out=bad)
        with pytest.raises(ValueError):
            ctx.subspace_scan(x, out=bad, thr=np.float32(0.5))
    with pytest.raises(ValueError):
        ctx.subspace_scan(x, thr=np.float32(0.5), hist=torch.zeros((2, 399), dtype=torch.int32, device='cuda'))
    ds = ctx.subspace_ds(x)
    with pytest.raises(ValueError):
        ctx.ds_moments(ds, sums=torch.zeros((2, 5), dtype=torch.float32, device='cuda'))
    with pytest.raises(ValueError):
        ctx.ds_moments(ds, hist=torch.zeros((3, 400), dtype=torch.int32, device='cuda'))
    ctx.close()


def test_basis_bcast_between_contexts():
    """dtx_device_count / dtx_basis_bcast (multi-device hosts): the basis of the first context
    replicated into the others, each then scanning its own share of the chunks -- on a one-GPU box both
    contexts sit on device 0 (the copy is device to device either way); identical results."""
    import torch
    from detex_amd import _lib, synth
    lib = _lib.load()
    assert lib.dtx_device_count() >= 1
    Nc, L, D, k = 3, 400, 4, 3
    n = Nc * L
    U, off = synth.detector_bank(2, D, k, n)
    x, plants = synth.continuous_day(3, 4, Nc * 30000, Nc, U, off, nplant=6)
    hs = (C.c_void_p * 2)()
    for i in range(2):
        h = C.c_void_p()
        dev = i % lib.dtx_device_count()
        assert lib.dtx_ctx_create(dev, None, 1, C.byref(h)) == 0
        hs[i] = h
    assert lib.dtx_basis_bcast(hs, 2) < 0 and b'no basis' in lib.dtx_last_error(hs[0])
    assert lib.dtx_basis_upload(hs[0], U.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), D, n, Nc) == 0
    assert lib.dtx_basis_bcast(hs, 2) == 0, lib.dtx_last_error(hs[0])
    P = lib.dtx_num_windows(Nc * 30000, n, Nc)
    out = []
    for i in range(2):                       # every context scores all four chunks here: they must agree bit for bit
        dev = i % lib.dtx_device_count()
        with torch.cuda.device(dev):
            xt = torch.from_numpy(x).cuda()
            ds = torch.empty((4, D, P), dtype=torch.float32, device='cuda')
            assert lib.dtx_subspace_ds(hs[i], C.c_void_p(xt.data_ptr()), Nc * 30000, Nc * 30000, 4,
                                       C.c_void_p(ds.data_ptr()), P) == 0, lib.dtx_last_error(hs[i])
            assert lib.dtx_ctx_sync(hs[i]) == 0
            out.append(ds.cpu().numpy())
    assert np.array_equal(out[0], out[1])
    for (b, p, d) in plants:
        assert out[1][b, d, p] > 0.9
    for i in range(2):
        lib.dtx_ctx_destroy(hs[i])


def test_kernel_timing_ring_and_ragged_arguments(raw):
    """dtx_kernel_ms_mean: the durations of the dominant kernel of the last n calls, read after the calls
    were enqueued (the ring holds 64); asking for more than were timed, or than the ring holds, is an error.
    dtx_subspace_scan_ragged / dtx_trigger_stats_ragged without a chunk table are errors too."""
    lib, h = raw
    ms = C.c_float()
    assert lib.dtx_kernel_ms_mean(h, 1, C.byref(ms)) == -3 and 'no timed kernel' in _msg(lib, h)
    Nc, L = 3, 400
    n = Nc * L
    rs = np.random.RandomState(0)
    U = np.linalg.qr(rs.standard_normal((n, 4)))[0].T.astype(np.float32).copy()
    off = np.array([0, 2, 4], np.int32)
    assert lib.dtx_basis_upload(h, U.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), 2, n, Nc) == 0
    M = Nc * 40000
    stream = torch.cuda.ExternalStream(lib.dtx_ctx_stream(h))
    with torch.cuda.stream(stream):
        x = torch.randn(M, device='cuda')
        P = (M - n) // Nc + 1
        ds = torch.empty((1, 2, P), device='cuda')
        for _ in range(5):
            assert lib.dtx_subspace_ds(h, C.c_void_p(x.data_ptr()), M, M, 1, C.c_void_p(ds.data_ptr()), P) == 0
        assert lib.dtx_kernel_ms_mean(h, 5, C.byref(ms)) == 0 and 0.0 < ms.value < 50.0
        one = C.c_float()
        assert lib.dtx_last_kernel_ms(h, C.byref(one)) == 0 and 0.0 < one.value < 50.0
        assert lib.dtx_kernel_ms_mean(h, 6, C.byref(ms)) == -1 and 'only the last' in _msg(lib, h)
        assert lib.dtx_kernel_ms_mean(h, 65, C.byref(ms)) == -1
        assert lib.dtx_kernel_ms_mean(h, 0, C.byref(ms)) == -1
        rc = lib.dtx_subspace_scan_ragged(h, C.c_void_p(x.data_ptr()), None, M, 1, C.c_void_p(ds.data_ptr()), P,
                                          None, None, None, None, None, 0, None)
        assert rc == -1 and 'chunk table' in _msg(lib, h)
        trig = torch.zeros((1, 5), dtype=torch.int32, device='cuda')
        out = torch.zeros((1, 6), dtype=torch.float64, device='cuda')
        rc = lib.dtx_trigger_stats_ragged(h, C.c_void_p(x.data_ptr()), None, M, C.c_void_p(ds.data_ptr()), P, P,
                                          C.c_void_p(trig.data_ptr()), 1, 1, 10, None, None, 0,
                                          C.c_void_p(out.data_ptr()))
        assert rc == -1 and 'chunk table' in _msg(lib, h)
    assert lib.dtx_ctx_sync(h) == 0

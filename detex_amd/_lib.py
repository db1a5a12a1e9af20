"""ctypes binding of libdetexhip.so (C ABI declared in include/detex_hip.h).

The library is the product's only compute path: there is no NumPy / CPU
fallback.  If it is missing or a call fails this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = 'libdetexhip.so'
LIB_PATH = os.path.join(_HERE, LIB_NAME)
ABI_VERSION = 1

# every symbol include/detex_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    'dtx_version', 'dtx_ctx_create', 'dtx_ctx_destroy', 'dtx_ctx_sync',
    'dtx_ctx_stream', 'dtx_last_error', 'dtx_set_engine', 'dtx_basis_upload', 'dtx_num_windows',
    'dtx_subspace_ds', 'dtx_subspace_scan', 'dtx_ds_post', 'dtx_ds_moments', 'dtx_subspace_ds_host',
    'dtx_xcorr_allpairs', 'dtx_xcorr_allpairs_host', 'dtx_preprocess', 'dtx_preprocess_planes', 'dtx_decimate', 'dtx_timer_start',
    'dtx_timer_stop', 'dtx_last_kernel_ms', 'dtx_trigger_stats', 'dtx_normcorr', 'dtx_ingest_counts', 'dtx_gapfree_run',
    'dtx_device_count', 'dtx_basis_bcast', 'dtx_kernel_ms_mean', 'dtx_subspace_scan_ragged',
    'dtx_trigger_stats_ragged', 'dtx_chunk_table_check', 'dtx_ctx_set_trigger_stream', 'dtx_stream_wait_scan_kernel',
    'dtx_kernel_ms_list', 'dtx_stalta_max', 'dtx_repair_counts', 'dtx_device_pci_bus_id', 'dtx_export_results', 'dtx_ctx_set_post_stream',
    'dtx_stream_wait_post', 'dtx_last_kernel_name',
]


class DetexHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load libdetexhip.so (once) and declare its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DetexHipError(
            '%s not found: build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (or `make -C detex_amd/csrc`).  detex_amd has no CPU '
            'fallback.' % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; importing it first makes this
    # library resolve against the same HIP runtime instance, so device
    # pointers and streams are shared with torch tensors.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32
    pf = C.POINTER(C.c_float)
    lib.dtx_version.restype = i32
    lib.dtx_version.argtypes = []
    lib.dtx_ctx_create.restype = i32
    lib.dtx_ctx_create.argtypes = [i32, vp, i32, C.POINTER(vp)]
    lib.dtx_ctx_destroy.restype = i32
    lib.dtx_ctx_destroy.argtypes = [vp]
    lib.dtx_ctx_sync.restype = i32
    lib.dtx_ctx_sync.argtypes = [vp]
    lib.dtx_ctx_set_trigger_stream.restype = i32
    lib.dtx_ctx_set_trigger_stream.argtypes = [vp, vp]
    lib.dtx_stream_wait_scan_kernel.restype = i32
    lib.dtx_stream_wait_scan_kernel.argtypes = [vp, vp]
    lib.dtx_ctx_stream.restype = vp
    lib.dtx_ctx_stream.argtypes = [vp]
    lib.dtx_last_error.restype = C.c_char_p
    lib.dtx_last_error.argtypes = [vp]
    lib.dtx_set_engine.restype = i32
    lib.dtx_set_engine.argtypes = [vp, i32]
    lib.dtx_basis_upload.restype = i32
    lib.dtx_basis_upload.argtypes = [vp, vp, vp, i32, i32, i32]
    lib.dtx_num_windows.restype = i64
    lib.dtx_num_windows.argtypes = [i64, i32, i32]
    lib.dtx_subspace_ds.restype = i32
    lib.dtx_subspace_ds.argtypes = [vp, vp, i64, i64, i32, vp, i64]
    lib.dtx_subspace_scan.restype = i32
    lib.dtx_subspace_scan.argtypes = [vp, vp, i64, i64, i32, vp, i64, vp, vp, vp, vp, vp, u32, vp]
    lib.dtx_subspace_scan_ragged.restype = i32
    lib.dtx_subspace_scan_ragged.argtypes = [vp, vp, vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, u32, vp]
    lib.dtx_chunk_table_check.restype = i32
    lib.dtx_chunk_table_check.argtypes = [vp, i32, i64, i64]
    lib.dtx_ds_post.restype = i32
    lib.dtx_ds_post.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, u32, vp]
    lib.dtx_ds_moments.restype = i32
    lib.dtx_ds_moments.argtypes = [vp, vp, i64, i64, i32, i32, C.c_double, C.c_double, i32, vp, vp]
    lib.dtx_subspace_ds_host.restype = i32
    lib.dtx_subspace_ds_host.argtypes = [vp, vp, i64, vp]
    lib.dtx_xcorr_allpairs.restype = i32
    lib.dtx_xcorr_allpairs.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i64, i64]
    lib.dtx_xcorr_allpairs_host.restype = i32
    lib.dtx_xcorr_allpairs_host.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.dtx_preprocess.restype = i32
    lib.dtx_preprocess.argtypes = [vp, vp, i32, i32, i64, i64, vp, i32, i32, i32, vp, i64]
    lib.dtx_preprocess_planes.restype = i32
    lib.dtx_preprocess_planes.argtypes = [vp, vp, i32, i32, i64, i64, i64, vp, i32, i32, i32, vp, i64]
    lib.dtx_decimate.restype = i32
    lib.dtx_decimate.argtypes = [vp, vp, i32, i64, i64, vp, i32, i32, vp, i64]
    lib.dtx_trigger_stats.restype = i32
    lib.dtx_trigger_stats.argtypes = [vp, vp, i64, i64, vp, i64, i64, vp, i32, i32, i32, vp, vp, i32, vp]
    lib.dtx_trigger_stats_ragged.restype = i32
    lib.dtx_trigger_stats_ragged.argtypes = [vp, vp, vp, i64, vp, i64, i64, vp, i32, i32, i32, vp, vp, i32, vp]
    lib.dtx_normcorr.restype = i32
    lib.dtx_normcorr.argtypes = [vp, vp, i32, vp, i64, vp]
    lib.dtx_ingest_counts.restype = i32
    lib.dtx_ingest_counts.argtypes = [vp, vp, vp, i32, i32, i64, vp, vp]
    lib.dtx_gapfree_run.restype = i32
    lib.dtx_gapfree_run.argtypes = [vp, vp, i32, i64, i64, vp]
    lib.dtx_device_count.restype = i32
    lib.dtx_device_count.argtypes = []
    lib.dtx_basis_bcast.restype = i32
    lib.dtx_basis_bcast.argtypes = [C.POINTER(vp), i32]
    lib.dtx_timer_start.restype = i32
    lib.dtx_timer_start.argtypes = [vp]
    lib.dtx_timer_stop.restype = i32
    lib.dtx_timer_stop.argtypes = [vp, pf]
    lib.dtx_kernel_ms_mean.restype = i32
    lib.dtx_kernel_ms_mean.argtypes = [vp, i32, pf]
    lib.dtx_stalta_max.restype = i32
    lib.dtx_stalta_max.argtypes = [vp, vp, vp, i32, i64, i32, i32, i32, vp]
    lib.dtx_kernel_ms_list.restype = i32
    lib.dtx_kernel_ms_list.argtypes = [vp, i32, pf]
    lib.dtx_device_pci_bus_id.restype = i32
    lib.dtx_device_pci_bus_id.argtypes = [i32, C.c_char_p, i32]
    lib.dtx_ctx_set_post_stream.restype = i32
    lib.dtx_ctx_set_post_stream.argtypes = [vp, vp]
    lib.dtx_stream_wait_post.restype = i32
    lib.dtx_stream_wait_post.argtypes = [vp, vp]
    lib.dtx_last_kernel_name.restype = C.c_char_p
    lib.dtx_last_kernel_name.argtypes = [vp]
    lib.dtx_export_results.restype = i32
    lib.dtx_export_results.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp]
    lib.dtx_repair_counts.restype = i32
    lib.dtx_repair_counts.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.dtx_last_kernel_ms.restype = i32
    lib.dtx_last_kernel_ms.argtypes = [vp, pf]
    if lib.dtx_version() != ABI_VERSION:
        raise DetexHipError('libdetexhip.so ABI %d != expected %d; rebuild it'
                            % (lib.dtx_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(lib, ctx, rc, what):
    if rc != 0:
        msg = lib.dtx_last_error(ctx)
        raise DetexHipError('%s failed (%d): %s' % (
            what, rc, msg.decode('utf-8', 'replace') if msg else ''))

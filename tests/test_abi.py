"""CPU-side checks of the C ABI: the library builds, loads, exports every symbol
include/detex_hip.h declares, and fails loudly without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'detex_amd', 'libdetexhip.so')


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(SO):
        import __graft_entry__ as g
        g.build()
    from detex_amd import _lib
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, 'include', 'detex_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'\b(dtx_[a-z0-9_]+)\s*\(', src)
    return sorted(set(n for n in names if n not in ('dtx_cand_pack', 'dtx_cand_unpack')))


def test_header_symbols_exported(lib):
    from detex_amd import _lib
    decl = _declared()
    assert decl == sorted(_lib.SYMBOLS)
    for s in decl:
        assert hasattr(lib, s), s


def test_version_and_num_windows(lib):
    assert lib.dtx_version() == 1
    assert lib.dtx_num_windows(1116000, 9000, 3) == 369001
    assert lib.dtx_num_windows(100, 100, 1) == 1
    assert lib.dtx_num_windows(99, 100, 1) == 0


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = ctypes.c_void_p()
    rc = lib.dtx_ctx_create(0, None, 0, ctypes.byref(h))
    assert rc < 0 and b'HIP' in lib.dtx_last_error(None)
    from detex_amd.hip import HipContext
    from detex_amd import DetexHipError
    with pytest.raises(DetexHipError):
        HipContext(0)


def test_sparse_trigger_picker_matches_oracle():
    """Host logic: the sparse trigger loop equals the dense reference loop."""
    import numpy as np
    from detex_amd.detect import pick_triggers_sparse
    from oracle import detex_oracle as orc
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'trig.npz'))
    sr = float(z['sr'])
    for nm in z['names']:
        key = nm.split(':')[0]
        ds, thr = z[key + '_ds'], float(z[key + '_thr'])
        idx = np.nonzero(ds >= thr)[0]
        got = pick_triggers_sparse(idx, ds[idx], len(ds), thr, sr, ds.max())
        want = orc.pick_triggers(ds, thr, sr, 0.0) if orc.eval_trig_con(ds.max(), thr) else []
        assert [g[0] for g in got] == [w['index'] for w in want], nm
        assert len(got) == z[key + '_rows'].shape[0], nm


def test_header_is_valid_c99(tmp_path):
    """The boundary header must be consumable from plain C (cgo / JNI / FFI users)."""
    import subprocess
    src = tmp_path / 't.c'
    src.write_text('#include "detex_hip.h"\n'
                   'int main(void) { uint32_t a, b, c; dtx_cand_unpack(dtx_cand_pack(3, 7, 99), &a, &b, &c);\n'
                   '  return !(a == 3 && b == 7 && c == 99 && DTX_HIST_BINS == 400); }\n')
    exe = tmp_path / 't'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic',
                           '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    assert subprocess.call([str(exe)]) == 0


def _build_c_example(tmp_path):
    import subprocess
    exe = tmp_path / 'ds_host'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'ds_host.c'),
                           '-L', os.path.join(ROOT, 'detex_amd'), '-ldetexhip', '-lm',
                           '-Wl,-rpath,' + os.path.join(ROOT, 'detex_amd'),
                           '-Wl,-rpath,/opt/rocm/lib', '-o', str(exe)])
    return exe


def test_c_example_links(lib, tmp_path):
    """A plain-C program links against the library through the header alone; without
    a GPU it must fail loudly (exit code 2: context creation error), never fall back."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if not torch.cuda.is_available():
        r = subprocess.run([str(exe)], capture_output=True, text=True)
        assert r.returncode == 2 and 'HIP' in r.stderr


@pytest.mark.gpu
def test_c_example_runs(lib, tmp_path):
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'planted at 1500' in r.stdout


def test_chunk_table_check_on_the_host(lib):
    """dtx_chunk_table_check: host-side validation of a ragged batch's table, no GPU involved (the library's
    share of the planners that tests/test_plan_sanitizer.py sweeps under ASAN / UBSAN)."""
    import numpy as np

    def chk(rows, M, numel):
        tab = np.ascontiguousarray(rows, dtype=np.int64)
        return lib.dtx_chunk_table_check(tab.ctypes.data_as(ctypes.c_void_p), len(tab), M, numel)
    assert chk([[0, 100], [50, 120], [170, 0]], 120, 170) == 0
    assert chk([[0, 100], [80, 100]], 100, 170) == 2          # reaches past the buffer
    assert chk([[-1, 10]], 100, 170) == 1
    assert chk([[0, 101]], 100, 170) == 1                     # longer than M
    assert chk([[2 ** 62, 2 ** 62]], 2 ** 62, 2 ** 63 - 1) == 1      # no overflow in the bound test
    assert lib.dtx_chunk_table_check(None, 1, 10, 10) != 0

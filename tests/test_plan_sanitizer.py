"""Host-side sanitizer coverage: the planning code of the HIP
library -- pairs / units / half split of the frequency-domain scan, the ping-pong grid's group split, the
segment / lag-block choice and work list of the all-pairs lag search, the context's rings, the chunk-table
check -- is plain C++ in detex_amd/csrc/dtx_plan.h.  tests/csrc/plan_check.cpp sweeps it over the shapes of the
GPU test sweeps under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU build only (GPU sanitizers are not
available on this pool and gpurun refuses them)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_planning_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'plan_check')
    src = os.path.join(ROOT, 'tests', 'csrc', 'plan_check.cpp')
    cc = subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-Wall', '-Wextra', '-Werror', '-fsanitize=address,undefined',
                         '-fno-sanitize-recover=all', src, '-o', exe], capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    env.pop('LD_PRELOAD', None)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert 'plan_check ok' in run.stdout


def test_library_sources_use_the_sanitized_planners():
    """The launch paths call the planners of dtx_plan.h (no second copy of the arithmetic in the .hip files)."""
    csrc = os.path.join(ROOT, 'detex_amd', 'csrc')
    fft = open(os.path.join(csrc, 'dtx_fft.hip')).read()
    xc = open(os.path.join(csrc, 'dtx_xcfft.hip')).read()
    ctx = open(os.path.join(csrc, 'dtx_ctx.hip')).read()
    assert 'dtxplan::plan_fft_basis(' in fft and 'dtxplan::plan_pp_grid(' in fft
    assert 'dtxplan::plan_xc_segments(' in xc and 'dtxplan::plan_xc_items22(' in xc
    assert 'dtxplan::ring_last_slots(' in ctx and 'dtxplan::check_chunk_table(' in ctx
    plan = open(os.path.join(csrc, 'dtx_plan.h')).read()
    assert 'hip' not in plan.replace('.hip', '').replace('HIP', '').lower().replace('libdetexhip', '')

"""Kernel times of the on-device pre-processing of a station-day (24 chunks x 3 channels x 372 000 samples: detrend +
zero-phase Butterworth band-pass + multiplex): run under rocprofv3 --kernel-trace --stats, or alone for the total."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd.hip import HipContext  # noqa: E402

ctx = HipContext(0)
raw = torch.randn((24, 3, 372000), device='cuda') * 100 + 1000
filt = [1.0, 10.0, 2, True]
out = ctx.preprocess(raw, filt=filt, sampling_rate=100.0)
for rep in range(3):
    ctx.timer_start()
    for _ in range(10):
        ctx.preprocess(raw, filt=filt, sampling_rate=100.0)
    print('preprocess of a station-day: %.3f ms' % (ctx.timer_stop() / 10))

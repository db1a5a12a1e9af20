import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detex_amd.hip import HipContext
ctx = HipContext(0)
raw = torch.randn(24, 3, 372000, device='cuda')
for _ in range(3):
    out = ctx.preprocess(raw, filt=[1, 10, 2, True], sampling_rate=100.0)
ctx.sync()

"""How long a 518 MB [3, T] record takes to the device: pageable, pinned (.to non_blocking), pinned through a pre-allocated
device buffer -- the FAS leg ran SLOWER from a pinned fetcher (0.077 against 0.043 s)."""
import time
import numpy as np
import torch
T = 5 * 24 * 3600 * 100
a = np.random.randn(3, T).astype(np.float32)
dev = torch.device('cuda', 0)
def t(label, fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print('%-50s %.1f ms' % (label, 1e3 * (time.perf_counter() - t0) / n))
t('pageable from_numpy().to(dev)', lambda: torch.from_numpy(a).to(dev))
p = torch.from_numpy(a).pin_memory()
t('pinned .to(dev, non_blocking=True)', lambda: p.to(dev, non_blocking=True))
buf = torch.empty((3, T), dtype=torch.float32, device=dev)
t('pinned copy_ into a resident buffer', lambda: buf.copy_(p, non_blocking=True))
t('device tensor allocation + free only', lambda: torch.empty((3, T), dtype=torch.float32, device=dev))

import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from detex_amd.hip import HipContext
ctx = HipContext(0)
for Nc in (3, 5):
    planes = torch.randn((1, Nc, 8640000), device='cuda')
    out = ctx.preprocess(planes, filt=None, detrend=False)
    ref = planes[0].T.reshape(-1)
    assert torch.equal(out[0], ref), Nc
    for rep in range(2):
        ctx.timer_start()
        for _ in range(20):
            ctx.preprocess(planes, filt=None, detrend=False)
        print('Nc=%d multiplex of a station-day: %.1f us' % (Nc, ctx.timer_stop() / 20 * 1e3))

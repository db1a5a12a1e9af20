import cProfile, pstats, sys, io, os, runpy
sys.argv=['scripts/e2e_time.py']
src=open('scripts/e2e_time.py').read()
# replace the three-pass loop with a profiled pass
src=src.replace("for rep in range(3):\n    res = ss.detex(subspaceDB=None, ctx=ctx)\n    t0 = tick('detex() pass %d' % rep, t0)",
"res = ss.detex(subspaceDB=None, ctx=ctx)\npr=cProfile.Profile(); pr.enable()\nfor rep in range(3):\n    res = ss.detex(subspaceDB=None, ctx=ctx)\n    torch.cuda.synchronize()\npr.disable()\nimport pstats; pstats.Stats(pr).sort_stats('cumulative').print_stats(28)")
src=src.replace("res = ss.detex(subspaceDB='/tmp/e2e_SubSpace.db', ctx=ctx, estimateMags=True)","sys.exit(0)")
exec(compile(src,'e2e','exec'))

import cProfile, os, pstats, sys
os.environ['E2E_COUNTS']='1'; os.environ['E2E_STATIONS']='4'
root=os.environ.get('GRAFT_REPO_ROOT','.'); sys.path.insert(0, root)
sys.argv=['scripts/e2e_time.py']
src=open(os.path.join(root,'scripts','e2e_time.py')).read()
head,tail=src.split("for rep in range(4):")
exec(compile(head,'e2e_head','exec'))
for _ in range(2): ss.detex(subspaceDB=None, estimateMags=False)
pr=cProfile.Profile(); pr.enable()
for _ in range(3): ss.detex(subspaceDB=None, estimateMags=False)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)

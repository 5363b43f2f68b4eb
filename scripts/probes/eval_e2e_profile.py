"""host-side profile of Evaluate end to end (where the time of scripts/eval_e2e_bench.py goes): cProfile over the second call"""
import cProfile, io, pstats, runpy, sys, os
sys.argv = [sys.argv[0]] + sys.argv[1:]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eval_e2e_bench.py"))
import torch
pr = cProfile.Profile()
pr.enable()
ns["ev"](ns["model"], ns["items"], ns["log"], "cuda:0")
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25)
print(s.getvalue()[:6000])

import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, bench
import graphnets_jl_amd as gn
cps, rvs, nn = bench.make_hetero(5, 4096, 1_000_000)
g = gn.GNGraphBatch.from_csc(cps, rvs, nn); torch.cuda.synchronize()
os.environ["GNX_TIME_BUILD"] = "1"
for i in range(2):
    t0 = time.perf_counter(); g = gn.GNGraphBatch.from_csc(cps, rvs, nn); torch.cuda.synchronize(); print("from_csc ms", (time.perf_counter() - t0) * 1e3, flush=True)
os.environ.pop("GNX_TIME_BUILD")
pr = cProfile.Profile(); pr.enable()
for i in range(5): g = gn.GNGraphBatch.from_csc(cps, rvs, nn)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
import graphnets_jl_amd as gn
torch.cuda.init(); torch.zeros(1, device="cuda")
for name, (cp, rv, nn) in (("c3", bench.make_hetero(3, 512, 1_000_000)), ("c5", bench.make_hetero(5, 4096, 1_000_000))):
    adjs = []
    for c, r, n in zip(cp, rv, nn):
        a = np.zeros((n, n), dtype=np.uint8); a[r, np.repeat(np.arange(n), np.diff(c))] = 1; adjs.append(a)
    gn.GNGraphBatch(adjs[:2])
    os.environ["GNX_TIME_BUILD"] = "1"
    for rep in range(3):
        t0 = time.perf_counter(); g = gn.GNGraphBatch(adjs); torch.cuda.synchronize(); t = time.perf_counter() - t0
        print(name, "rep", rep, "dense uint8 total ms", round(t * 1e3, 3), flush=True)
    del os.environ["GNX_TIME_BUILD"]
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); g = gn.GNGraphBatch(adjs); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(6)
PY

cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
import graphnets_jl_amd as gn
torch.cuda.init(); torch.zeros(1, device="cuda")
for name, (cp, rv, nn) in (("c2", bench.make_c2()), ("c3", bench.make_hetero(3, 512, 1_000_000)), ("c5", bench.make_hetero(5, 4096, 1_000_000))):
    gn.GNGraphBatch.from_csc(cp[:1], rv[:1], nn[:1])
    os.environ["GNX_TIME_BUILD"] = "1"
    for rep in range(3):
        t0 = time.perf_counter(); g = gn.GNGraphBatch.from_csc(cp, rv, nn); torch.cuda.synchronize(); t = time.perf_counter() - t0
        print(name, "rep", rep, "from_csc total ms", round(t * 1e3, 3), flush=True)
    del os.environ["GNX_TIME_BUILD"]
PY

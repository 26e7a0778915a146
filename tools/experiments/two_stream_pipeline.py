import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import graphnets_jl_amd as gn
dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
E, N, G = g.n_edges, g.n_nodes, g.n_graphs
din, dout = bench.DIMS["readme"]
rng = np.random.default_rng(100)
blk = gn.GNBlock(din, dout, device=dev)
(de, dn, dg), (oe, on, og) = din, dout
blk.edgefn = gn.Dense.from_numpy(bench.glorot(rng, oe, de + 2 * dn + dg), np.zeros(oe, np.float32), device=dev)
blk.nodefn = gn.Dense.from_numpy(bench.glorot(rng, on, oe + dn + dg), np.zeros(on, np.float32), device=dev)
blk.graphfn = gn.Dense.from_numpy(bench.glorot(rng, og, oe + on + dg), np.zeros(og, np.float32), device=dev)
plan = gn.BlockPlan(blk, g, R=1)
NS = 8
tg = torch.Generator(device=dev); tg.manual_seed(1)
mk = lambda T, d: torch.rand((1, T, d), generator=tg, device=dev) if d > 0 else None
sets = [dict(ef=mk(E, de), nf=mk(N, dn), gf=mk(G, dg), out=plan.outputs(), ws=plan.new_workspace()) for _ in range(NS)]
def step(i):
    b = sets[i % NS]
    plan(b["ef"], b["nf"], b["gf"], *b["out"], ws=b["ws"])
for i in range(4): step(i)
torch.cuda.synchronize()
K = 40
def capture(idx):
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        for i in idx: step(i)
    return cg
one = capture(range(K))
one.replay(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); one.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
print("one stream us/step", sorted(ts))
ga, gb = capture(range(0, K, 2)), capture(range(1, K, 2))
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def two():
    with torch.cuda.stream(sa): ga.replay()
    with torch.cuda.stream(sb): gb.replay()
two(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); two(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
print("two streams us/step", sorted(ts))
# correctness: outputs equal to single-stream results
ref = [tuple(o.clone() for o in s["out"] if o is not None) for s in sets]
two(); torch.cuda.synchronize()
print("equal:", all(torch.equal(a, b) for s, r in zip(sets, ref) for a, b in zip([o for o in s["out"] if o is not None], r)))

import time, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
import bench, graphnets_jl_amd as gn
cp, rv, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(cp, rv, nn)
blk = gn.GNBlock((10, 5, 0), (3, 4, 5))
ps = []
for l in (blk.edgefn, blk.nodefn, blk.graphfn):
    l.weight.requires_grad_(True); l.bias.requires_grad_(True); ps += [l.weight, l.bias]
dev = g.device
ef = torch.rand((1, g.n_edges, 10), device=dev, requires_grad=True)
nf = torch.rand((1, g.n_nodes, 5), device=dev, requires_grad=True)
x = gn.NT(g, ef.permute(2, 1, 0), nf.permute(2, 1, 0), None)
def it():
    y = blk(x)
    (y.ef.sum() + y.nf.sum() + y.gf.sum()).backward()
for _ in range(3): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): it()
torch.cuda.synchronize(); print("C2 README dims forward+backward: %.1f us / iteration" % ((time.perf_counter() - t0) / 10 * 1e6))

"""what are the ~19 MB that k_block_wave moves beyond its compulsory bytes on C2 — HBM reads or Infinity-Cache hits?
The TCC counters cannot tell (profiles/r03_readme_pmc.json: every memory-side read request is DRAM-class, hits in the memory-side cache
included), so this A/B varies only the residency of the 2 MB node table, which every one of the 8 XCD L2s fetches once per launch:
  cold      : 8 rotating buffer sets (480 MB > 256 MiB Infinity Cache)         — the headline protocol
  shared nf : the same, but all 8 sets read ONE node table (it stays resident) — if the 7 duplicate fetches were HBM reads this is faster
  warm      : 2 buffer sets (120 MB: everything resident)
Prints us/step of a K-step hipGraph, median of 5."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import graphnets_jl_amd as gn

dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
E, N, G = g.n_edges, g.n_nodes, g.n_graphs
din, dout = bench.DIMS["readme"]
(de, dn, dg), (oe, on, og) = din, dout
rng = np.random.default_rng(100)
blk = gn.GNBlock(din, dout, device=dev)
blk.edgefn = gn.Dense.from_numpy(bench.glorot(rng, oe, de + 2 * dn + dg), np.zeros(oe, np.float32), device=dev)
blk.nodefn = gn.Dense.from_numpy(bench.glorot(rng, on, oe + dn + dg), np.zeros(on, np.float32), device=dev)
blk.graphfn = gn.Dense.from_numpy(bench.glorot(rng, og, oe + on + dg), np.zeros(og, np.float32), device=dev)
plan = gn.BlockPlan(blk, g, R=1)
tg = torch.Generator(device=dev); tg.manual_seed(1)
mk = lambda T, d: torch.rand((1, T, d), generator=tg, device=dev) if d > 0 else None
sets = [dict(ef=mk(E, de), nf=mk(N, dn), out=plan.outputs(), ws=plan.new_workspace()) for _ in range(8)]
K = 40


def run(nsets, shared_nf):
    def step(i):
        b = sets[i % nsets]
        plan(b["ef"], sets[0]["nf"] if shared_nf else b["nf"], None, *b["out"], ws=b["ws"])
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        for i in range(K):
            step(i)
    cg.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); cg.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
    return sorted(ts)[2]


for name, (ns, sh) in (("cold (8 sets)", (8, False)), ("cold, ONE shared node table", (8, True)), ("warm (2 sets)", (2, False)), ("cold (8 sets) again", (8, False))):
    print(f"{name:32s} {run(ns, sh):7.2f} us/step", flush=True)

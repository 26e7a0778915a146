"""Diagnostic (run by hand on the GPU box): one seed of tests/test_gpu_fuzz.py::test_random_chain_block_backward with every gradient's error against
torch float64 AND torch float32 of the same restatement printed side by side.  python tests/chain_bw_probe.py <seed>"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import graphnets_jl_amd as gn
import tests.test_gpu_fuzz as F
from oracle import gn_oracle as O
from tests import util as U
from tests.test_gpu_chain import _block, _torch_chain_block

seed = int(sys.argv[1])
rng = np.random.default_rng(9750 + seed)
big = rng.random() < 0.3
g, _ = F._random_big_batch(rng, gn) if big else F._random_batch(rng, gn)
widths = [8, 24, 48, 64] if big else [1, 3, 7, 12, 16, 33]
while True:
    in_dims = tuple(int(rng.choice([0] + widths)) for _ in range(3))
    if sum(in_dims) > 0:
        break
ew = F._random_chain(rng, widths, sum(in_dims)) or [int(rng.choice(widths))]
oe = next((w for w in reversed(ew) if w != "ln"), 0)
nw = F._random_chain(rng, widths, oe + in_dims[1] + in_dims[2])
on = next((w for w in reversed(nw) if w != "ln"), 0)
gw = F._random_chain(rng, widths, oe + on + in_dims[2])
acts = tuple(int(a) for a in rng.choice([0, 2, 3, 4], 3))
print("seed", seed, in_dims, ew, nw, gw, acts, "N", g.n_nodes, "E", g.n_edges, "G", g.n_graphs)
csc = (*g.csc(), g.node_off, g.edge_off)
p = O.make_chain_block_params(rng, in_dims, ew, nw, gw, acts=acts)
ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, in_dims)
res = {}
cots = None
for dt in (torch.float64, torch.float32):
    T = lambda a: None if a is None else torch.tensor(a[0], dtype=dt, requires_grad=True)
    xs = [T(ef), T(nf), T(gf)]
    Tw = lambda v: torch.tensor(v, dtype=dt, requires_grad=True)
    W = {name: [(w, Tw(b), Tw(a)) if isinstance(w, str) else (Tw(w), Tw(b), a) for w, b, a in p[name]] for name in ("edge", "node", "graph")}
    outs_r, pre = _torch_chain_block(csc, *xs, W)
    if cots is None:
        cots = [None if o is None else torch.from_numpy(rng.standard_normal(tuple(o.shape))) for o in outs_r]
    sum((o * c.to(dt)).sum() for o, c in zip(outs_r, cots) if o is not None).backward()
    refs = [q.grad for name in ("edge", "node", "graph") for w, b, a in W[name] for q in ((b, a) if isinstance(w, str) else (w, b))]
    res[dt] = [r.double().numpy() for r in refs] + [None if x is None else x.grad.double().numpy() for x in xs]
blk = _block(gn, p)
leaves = []
for ch in (blk.edgefn, blk.nodefn, blk.graphfn):
    for l in ch.layers:
        l.weight.requires_grad_(True); l.bias.requires_grad_(True)
        leaves += [l.weight, l.bias]
dev = g.device
leaf = lambda a: None if a is None else torch.from_numpy(a).to(dev).requires_grad_(True)
xt = [leaf(ef), leaf(nf), leaf(gf)]
y = blk(gn.NT(g, *(None if t is None else t.permute(2, 1, 0) for t in xt)))
loss = sum((o.permute(2, 1, 0)[0] * c.to(dev).float()).sum() for o, c in zip((y.ef, y.nf, y.gf), cots) if o is not None)
loss.backward()
got = [q.grad.double().cpu().numpy() for q in leaves] + [None if t is None else t.grad[0].double().cpu().numpy() for t in xt]
for i, (a, b, c) in enumerate(zip(res[torch.float64], res[torch.float32], got)):
    if a is None:
        continue
    e_hip, e_t32 = np.abs(c - a), np.abs(b - a)
    print(f"{i:2d} {str(a.shape):12s} max|ref| {np.abs(a).max():9.3g}  hip err {np.nanmax(e_hip):9.3g}  torch-fp32 err {np.nanmax(e_t32):9.3g}  nan(ref) {int(np.isnan(a).sum())} nan(hip) {int(np.isnan(c).sum())}")
    if np.nanmax(e_hip) > 1e-3 * max(1.0, np.abs(a).max()):
        k = np.unravel_index(np.nanargmax(e_hip), e_hip.shape)
        print("    worst at", k, "ref", a[k], "hip", c[k], "; entries beyond 1e-4 scale:", int((e_hip > 1e-4 * max(1.0, np.abs(a).max())).sum()), "of", a.size)

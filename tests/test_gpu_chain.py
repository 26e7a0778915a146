"""GNBlock with Flux `Chain`s of Dense layers as update functions (gnblock.jl:1-6; gnx_chain_block_forward) against the oracle."""
import numpy as np
import pytest

from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


def _block(gn, p):
    de, dn, dg = p["in_dims"]
    blk = gn.GNBlock((de, dn, dg), (1, 0, 0))  # placeholder layers: the chains below define the widths
    mk = lambda layers: gn.Chain([gn.LayerNorm.from_numpy(b, a) if isinstance(W, str) else gn.Dense.from_numpy(W, b, U.ACT_NAMES[a]) for W, b, a in layers])
    blk.edgefn, blk.nodefn, blk.graphfn = mk(p["edge"]), mk(p["node"]), mk(p["graph"])
    return blk


CASES = [
    ((10, 5, 0), [16, 3], [8, 4], [6, 5]),          # README widths with one hidden layer everywhere
    ((10, 5, 3), [32, 24, 7], [12], [9, 4]),        # three edge layers, one-layer node function
    ((0, 4, 2), [8, 5], [5], []),                   # ef = nothing, no graph output
    ((6, 0, 0), [4], [], []),                       # plain one-layer edge function only
    ((128, 64, 32), [256, 128], [128, 64], [64, 32]),  # wide: every layer on the matrix cores
    # a Flux LayerNorm(d) layer value between / behind the Dense layers (`Chain(Dense, LayerNorm, Dense)`, gnblock.jl:1-6)
    ((10, 5, 0), [16, "ln", 3], [8, "ln", 4], ["ln", 6, 5]),   # ... and as the FIRST layer of the graph function (its input is materialised)
    ((10, 5, 3), [12, 7, "ln"], ["ln", 6], [9, "ln", 4, "ln"]),  # as a chain's last layer; first layer of the node function
    ((128, 64, 32), [256, "ln", 128], [128, "ln", 64], [64, 32]),  # wide rows
    # ... and as the EDGE function's first layer (pre-norm of the concatenated input): behind an identity Dense the library puts in front
    ((10, 5, 3), ["ln", 12, 7], [6], [4]),
    ((10, 5, 0), ["ln"], ["ln", 4], []),               # the edge function IS the LayerNorm: ef' = LayerNorm(getedgefninput), 20 wide
    ((128, 64, 32), ["ln", 128, 64], [64], [32]),      # wide: K_e = 288
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_chain_block_matches_oracle(gn, case):
    in_dims, ew, nw, gw = case
    rng = np.random.default_rng(sum(in_dims) + len(ew) * 7 + len(nw) * 3)
    wide = max(in_dims) >= 64
    sizes = rng.integers(60, 200, 3) if wide else rng.integers(3, 40, 5)
    adjs = [(rng.random((n, n)) < (0.08 if wide else 0.3)).astype(np.int64) for n in sizes]
    csc = O.csc_from_adj(adjs)
    g = gn.GNGraphBatch(adjs)
    p = O.make_chain_block_params(rng, in_dims, ew, nw, gw)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, in_dims)
    y = _block(gn, p)(U.to_nt(gn, g, ef, nf, gf))
    ref, scale = O.chain_block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name)


def test_chain_block_shared_graph_replicas_and_errors(gn):
    rng = np.random.default_rng(5)
    adj = np.array([[1, 0, 1], [1, 1, 0], [0, 0, 1]])
    p = O.make_chain_block_params(rng, (10, 5, 0), [12, 3], [6, 4], [5])
    ef, nf = rng.random((10, 5, 2), dtype=np.float32), rng.random((5, 3, 2), dtype=np.float32)
    y = gn.unbatch(_block(gn, p)(gn.batch(dict(graphs=adj, ef=ef, nf=nf, gf=None))))
    assert tuple(y.ef.shape) == (3, 5, 2) and tuple(y.nf.shape) == (4, 3, 2) and tuple(y.gf.shape) == (5, 2)
    pk = (O.packed_from_julia_shared(ef), O.packed_from_julia_shared(nf), None)
    ref = O.chain_block_forward_sparse(p, O.csc_from_adj([adj]), *pk)
    np.testing.assert_allclose(np.transpose(y.ef.cpu().numpy(), (2, 1, 0)), ref[0], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.gf.cpu().numpy().T[:, None, :], ref[2], rtol=1e-5, atol=2e-5)
    # zero-width outputs: the reference computes getnodefninput / getgraphfninput over the 0-row h_ef / h_nf (gnblock.jl:63-69), and so do
    # one-layer blocks here — a Chain block must take the same dims: no node function, then no edge function
    for ew, nw, gw in (([4, 3], [], [5]), ([], [6, 4], [5])):
        pz = O.make_chain_block_params(rng, (10, 5, 0), ew, nw, gw)
        yz = gn.unbatch(_block(gn, pz)(gn.batch(dict(graphs=adj, ef=ef, nf=nf, gf=None))))
        rz, sz = O.chain_block_forward_sparse(pz, O.csc_from_adj([adj]), *pk, return_scale=True)
        assert (yz.ef is None) == (not ew) and (yz.nf is None) == (not nw)
        for got, r, sc in ((yz.ef, rz[0], sz[0]), (yz.nf, rz[1], sz[1])):
            if r is not None:
                U.assert_close(np.transpose(got.cpu().numpy(), (2, 1, 0)), r, sc)
        U.assert_close(yz.gf.cpu().numpy().T[:, None, :], rz[2], sz[2])


def _torch_chain_block(csc, ef, nf, gf, W):
    """float64 torch restatement of the block with Chain update functions (one replica); W[name] = list of (weight, bias, act)."""
    import torch
    ACT = {0: lambda x: x, 1: torch.relu, 2: torch.tanh, 3: torch.sigmoid, 4: lambda x: torch.nn.functional.gelu(x, approximate="tanh")}
    colptr, rowval, node_off, edge_off = (torch.from_numpy(np.asarray(a)) for a in csc)
    N, G = len(colptr) - 1, len(node_off) - 1
    dst = torch.repeat_interleave(torch.arange(N), colptr[1:] - colptr[:-1])
    ng = torch.repeat_interleave(torch.arange(G), node_off[1:] - node_off[:-1])
    eg = torch.repeat_interleave(torch.arange(G), edge_off[1:] - edge_off[:-1])
    cat = lambda parts: torch.cat([q for q in parts if q is not None], dim=1)

    def chain(x, layers, pre):
        for w, b, a in layers:
            if isinstance(w, str):  # ("layernorm", gamma, beta): Flux 0.14 normalise (x - mean) / (sigma + eps), then gamma . xhat + beta
                mu = x.mean(dim=1, keepdim=True)
                sd = ((x - mu) ** 2).mean(dim=1, keepdim=True).sqrt()
                x = b * ((x - mu) / (sd + 1e-5)) + a
                continue
            z = x @ w.T + b
            pre.append((z, a))
            x = ACT[a](z)
        return x

    pre = []
    he = chain(cat([ef, None if nf is None else nf[rowval], None if nf is None else nf[dst], None if gf is None else gf[eg]]), W["edge"], pre)
    hn = hg = None
    if W["node"]:
        agg = torch.zeros((N, he.shape[1]), dtype=he.dtype).index_add(0, dst, he)
        hn = chain(cat([agg, nf, None if gf is None else gf[ng]]), W["node"], pre)
    if W["graph"]:  # (a block without a node function: its zero-width nf' is an empty segment of the graph function's input, gnblock.jl:63-69)
        se = torch.zeros((G, he.shape[1]), dtype=he.dtype).index_add(0, eg, he)
        sn = None if hn is None else torch.zeros((G, hn.shape[1]), dtype=he.dtype).index_add(0, ng, hn)
        hg = chain(cat([se, sn, gf]), W["graph"], pre)
    return (he, hn, hg), pre


BW_CASES = [
    ((10, 5, 0), [16, 3], [8, 4], [6, 5], (1, 2, 0), False),
    ((10, 5, 3), [12, 9, 7], [6], [9, 4], (2, 3, 4), False),        # three edge layers (tanh, sigmoid), gelu hidden layers further down
    ((0, 4, 2), [8, 5], [5], [], (4, 2, 0), False),                 # ef = nothing, gelu first edge layer, no graph output
    ((6, 0, 0), [4, 3], [], [], (2, 0, 0), False),                  # edge function only
    ((48, 24, 8), [64, 40], [48, 24], [32, 16], (2, 3, 2), True),   # matrix-core row-wise pullbacks (>= 4096 rows)
    ((10, 5, 3), [16, "ln", 3], [8, "ln", 4, "ln"], ["ln", 6, 5], (2, 3, 2), False),   # LayerNorm layer values: between, last, first
    ((48, 24, 8), [64, "ln", 40], [48, "ln", 24], [32, 16], (2, 3, 2), True),          # ... behind matrix-core layers
    ((10, 5, 3), ["ln", 12, 7], [6, "ln"], [4], (2, 3, 2), False),                     # first layer of the EDGE function (identity Dense in front)
    ((48, 24, 8), ["ln", 64, 40], [24], [16], (2, 3, 2), True),
]


@pytest.mark.parametrize("case", BW_CASES, ids=[str(c[:4]) for c in BW_CASES])
def test_chain_block_backward_matches_torch_autograd(gn, case):
    """gnx_chain_block_backward through torch autograd against float64 torch autograd of an independent restatement: input gradients and
    every layer's weight / bias gradient.  Hidden activations are smooth or, for relu, drawn kink-free."""
    in_dims, ew, nw, gw, acts, big = case
    for attempt in range(20):
        rng = np.random.default_rng(700 + sum(in_dims) + 1000 * attempt)
        if big:
            sizes = rng.integers(1500, 2200, 3)
            cs = [U.er_csc(rng, int(n), 4 * int(n)) for n in sizes]
        else:
            sizes = rng.integers(3, 30, 5)
            cs = [U.er_csc(rng, int(n), int(0.2 * n * n) + 1) for n in sizes]
        g = gn.GNGraphBatch.from_csc([c[0] for c in cs], [c[1] for c in cs], [int(n) for n in sizes])
        if chain_block_backward_case(gn, g, rng, in_dims, ew, nw, gw, acts):
            return
    pytest.fail("no kink-free draw in 20 attempts")


def chain_block_backward_case(gn, g, rng, in_dims, ew, nw, gw, acts, fp32_yardstick=False):
    """One comparison on batch `g`; False when a relu pre-activation sits within fp32 rounding of its kink (re-draw).  The bar is 1e-3 of the
    gradient's largest entry; `fp32_yardstick` (the random sweeps: hubs of 150 in-edges in front of a LayerNorm make sums that cancel to 1 % of
    their terms) also accepts 20 x the error of the SAME restatement evaluated by torch in float32 — a conditioning yardstick, not a looser bar."""
    import torch
    csc = (*g.csc(), g.node_off, g.edge_off)
    p = O.make_chain_block_params(rng, in_dims, ew, nw, gw, acts=acts)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, in_dims)

    def restatement(dt, cot):
        T = lambda a: None if a is None else torch.tensor(a[0], dtype=dt, requires_grad=True)
        xs = [T(ef), T(nf), T(gf)]
        Tw = lambda v: torch.tensor(v, dtype=dt, requires_grad=True)
        W = {name: [(w, Tw(b), Tw(a)) if isinstance(w, str) else (Tw(w), Tw(b), a) for w, b, a in p[name]] for name in ("edge", "node", "graph")}
        outs_r, pre = _torch_chain_block(csc, *xs, W)
        if cot is None:
            if any(a == 1 and float(z.detach().abs().min()) < 5e-6 for z, a in pre if z.numel()):
                return None  # a relu pre-activation within fp32 rounding of its kink: re-draw
            cot = [None if o is None else torch.from_numpy(rng.standard_normal(tuple(o.shape))) for o in outs_r]
        sum((o * c.to(dt)).sum() for o, c in zip(outs_r, cot) if o is not None).backward()
        return xs, W, cot

    r64 = restatement(torch.float64, None)
    if r64 is None:
        return False
    xs, W, cot = r64
    yard = {}
    if fp32_yardstick:
        xs32, W32, _ = restatement(torch.float32, cot)
        g32 = [None if x is None else x.grad for x in xs32] + [q.grad for name in ("edge", "node", "graph") for w, b, a in W32[name] for q in ((b, a) if isinstance(w, str) else (w, b))]
        g64 = [None if x is None else x.grad for x in xs] + [q.grad for name in ("edge", "node", "graph") for w, b, a in W[name] for q in ((b, a) if isinstance(w, str) else (w, b))]
        names = ["d_ef", "d_nf", "d_gf"] + [f"param[{i}]" for i in range(len(g64) - 3)]
        yard = {n: 20.0 * float((a.double() - b).abs().max()) for n, a, b in zip(names, g32, g64) if b is not None and b.numel()}
    # HIP
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
    loss = sum((o.permute(2, 1, 0)[0] * c.to(dev).float()).sum() for o, c in zip((y.ef, y.nf, y.gf), cot) if o is not None)
    loss.backward()

    def close(got, ref, what):
        ref = ref.detach().numpy(); got = got.detach().double().cpu().numpy()
        if ref.size == 0:  # (the gradient of a (0, d) array: a batch without edges)
            assert got.shape == ref.shape, (what, got.shape, ref.shape)
            return
        scale = max(1.0, float(np.abs(ref).max()))
        assert got.shape == ref.shape, (what, got.shape, ref.shape)
        bar = max(1e-3 * scale, yard.get(what, 0.0))
        assert np.max(np.abs(got - ref)) <= bar, f"{what}: max err {np.max(np.abs(got - ref)):.3e} (scale {scale:.3g}, bar {bar:.3g})"

    for name, t, r in zip(("d_ef", "d_nf", "d_gf"), xt, xs):
        if t is not None:
            close(t.grad[0], r.grad, name)
    refs = [q.grad for name in ("edge", "node", "graph") for w, b, a in W[name] for q in ((b, a) if isinstance(w, str) else (w, b))]
    for i, (q, r) in enumerate(zip(leaves, refs)):
        close(q.grad, r, f"param[{i}]")
    return True


def test_chain_with_activation_identity_and_dropout_layer_values(gn):
    """Flux users write `Chain(Dense(a => b), relu, Dense(b => c), Dropout(p))`: the activation folds into the Dense in front of it, identity is
    dropped, Dropout is the identity in test mode — the same launches, the same bits as the explicit `Dense(a => b, relu)` form; a differentiable
    call with a Dropout(p > 0) inside a chain is refused."""
    import torch
    rng = np.random.default_rng(11)
    in_dims = (10, 5, 3)
    adjs = [(rng.random((n, n)) < 0.3).astype(np.int64) for n in rng.integers(3, 40, 5)]
    g = gn.GNGraphBatch(adjs)
    p = O.make_chain_block_params(rng, in_dims, [16, 3], [8, 4], [6, 5])
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, in_dims)
    x = U.to_nt(gn, g, ef, nf, gf)
    ref = _block(gn, p)(x)
    blk = gn.GNBlock(in_dims, (1, 0, 0))

    def spelled_out(layers, extra=()):
        out = []
        for W, b, a in layers:
            out.append(gn.Dense.from_numpy(W, b, "identity"))
            out.append(U.ACT_NAMES[a] if U.ACT_NAMES[a] != "relu" else torch.relu)
        return gn.Chain(out + list(extra))

    blk.edgefn, blk.nodefn, blk.graphfn = spelled_out(p["edge"], [gn.Dropout(0.3)]), spelled_out(p["node"], [None]), spelled_out(p["graph"])
    y = blk(x)
    for a, b in zip((y.ef, y.nf, y.gf), (ref.ef, ref.nf, ref.gf)):
        assert torch.equal(a, b)
    for l in blk.edgefn.layers:
        l.weight.requires_grad_(True)
    with pytest.raises(NotImplementedError, match="Dropout"):
        blk(x)
    with torch.no_grad():
        assert torch.equal(blk(x).ef, ref.ef)
    blk.edgefn = spelled_out(p["edge"])      # without the Dropout the differentiable call runs (gnx_chain_block_backward)
    for l in blk.edgefn.layers:
        l.weight.requires_grad_(True)
    blk(x).ef.sum().backward()
    assert all(l.weight.grad is not None and bool(torch.isfinite(l.weight.grad).all()) for l in blk.edgefn.layers)


def test_update_functions_replaced_after_construction_set_the_output_widths(gn):
    """gnblock.jl:1-6: a GNBlock is a struct of three functions; one-layer Chains (or Dense layers) assigned after construction run on the fused
    block kernels with THEIR widths, not the constructor's (found by the random chain sweep: the mirror kept the placeholder's widths)."""
    rng = np.random.default_rng(12)
    adjs = [(rng.random((n, n)) < 0.3).astype(np.int64) for n in (9, 14, 5)]
    g = gn.GNGraphBatch(adjs)
    p = O.make_chain_block_params(rng, (6, 4, 3), [8], [5], [7])
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, (6, 4, 3))
    blk = _block(gn, p)  # built as GNBlock((6, 4, 3) => (1, 0, 0)), then the three one-layer chains assigned
    y = blk(U.to_nt(gn, g, ef, nf, gf))
    assert blk.out_dims == (8, 5, 7)
    ref, scale = O.chain_block_forward_sparse(p, O.csc_from_adj(adjs), ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name)
    blk.nodefn = gn.Dense(3, 5)  # a function that does not take what the block feeds it: Flux's DimensionMismatch
    with pytest.raises(AssertionError):
        blk(U.to_nt(gn, g, ef, nf, gf))

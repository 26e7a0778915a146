"""Parity at BASELINE.json's FULL sizes (1M-edge batches): the float64 oracle is vectorised numpy and finishes in
seconds, so configs 2 and 3 are compared directly, plus the size-independent properties the domain offers
(affinity of the identity-activation block, graph independence inside a heterogeneous batch, bitwise reproducibility)."""
import numpy as np
import pytest

import bench
from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


def _graph(gn, maker):
    colptrs, rowvals, nn = maker()
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn)
    return g, (*g.csc(), g.node_off, g.edge_off)


@pytest.mark.parametrize("dims", [((10, 5, 0), (3, 4, 5)), ((10, 5, 3), (10, 5, 3))], ids=["readme", "core-readme"])
def test_config2_er_100k_nodes_1m_edges(gn, dims):
    """BASELINE configs[1]: one shared Erdős–Rényi graph, 100k nodes / 1M edges."""
    g, csc = _graph(gn, bench.make_c2)
    assert g.n_edges == 1_000_000 and g.n_nodes == 100_000
    rng = np.random.default_rng(100)
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, dims[0])
    y = U.block_from_params(gn, p)(U.to_nt(gn, g, ef, nf, gf))
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name, log=f"configs[1] C2 1M edges {dims[0]}=>{dims[1]}")


def test_config3_512_graphs_1m_edges_and_graph_independence(gn):
    """BASELINE configs[2]: 512 random graphs (32..256 nodes), 1M edges; and every graph's result equals the result of
    running that graph alone (runtests.jl:62-116 batch invariance at full size, sampled)."""
    colptrs, rowvals, nn = bench.make_hetero(3)
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn)
    csc = (*g.csc(), g.node_off, g.edge_off)
    assert g.n_edges == 1_000_000 and g.n_graphs == 512
    rng = np.random.default_rng(101)
    dims = ((10, 5, 0), (3, 4, 5))
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    blk = U.block_from_params(gn, p)
    y = blk(U.to_nt(gn, g, ef, nf, gf))
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name, log="configs[2] C3 512 graphs 1M edges (10,5,0)=>(3,4,5)")
    yb = [U.from_jl(a) for a in (y.ef, y.nf, y.gf)]
    for i in (0, 17, 511):
        e0, e1, n0, n1 = g.edge_off[i], g.edge_off[i + 1], g.node_off[i], g.node_off[i + 1]
        gi = gn.GNGraphBatch.from_csc([colptrs[i]], [rowvals[i]], [nn[i]])
        yi = blk(U.to_nt(gn, gi, ef[:, e0:e1], nf[:, n0:n1], None))
        np.testing.assert_allclose(U.from_jl(yi.ef)[0], yb[0][0, e0:e1], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(U.from_jl(yi.nf)[0], yb[1][0, n0:n1], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(U.from_jl(yi.gf)[0, 0], yb[2][0, i], rtol=1e-5, atol=1e-4)


def test_full_size_affinity_and_reproducibility(gn):
    """With identity activations the block is affine in its inputs: f(a·x + (1-a)·z) = a·f(x) + (1-a)·f(z); and two
    launches give identical bits (atomic-free fixed-order reductions)."""
    import torch
    g, _ = _graph(gn, bench.make_c2)
    rng = np.random.default_rng(102)
    dims = ((10, 5, 0), (3, 4, 5))
    blk = U.block_from_params(gn, O.make_block_params(rng, *dims))
    x = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, dims[0])
    z = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, dims[0])
    a = 0.25
    mix = tuple(None if u is None else (a * u + (1 - a) * v).astype(np.float32) for u, v in zip(x, z))
    fx, fz, fm = (blk(U.to_nt(gn, g, *t)) for t in (x, z, mix))
    for u, v, w in ((fx.ef, fz.ef, fm.ef), (fx.nf, fz.nf, fm.nf), (fx.gf, fz.gf, fm.gf)):
        lhs, rhs = w.double(), a * u.double() + (1 - a) * v.double()
        tol = 2e-5 * float(torch.maximum(u.abs().max(), v.abs().max())) * (30 if u is fx.gf else 1)  # gf sums 1M terms
        assert float((lhs - rhs).abs().max()) <= tol
    again = blk(U.to_nt(gn, g, *x))
    assert torch.equal(again.ef, fx.ef) and torch.equal(again.nf, fx.nf) and torch.equal(again.gf, fx.gf)


def test_config4_full_size_encoder_2cores_decoder(gn):
    """BASELINE configs[3] at its stated size: Encoder (10,5,0)=>(128,64,32) -> 2 x GNCore(128,64,32) -> Decoder =>(3,4,5) on
    the 100k-node / 1M-edge graph, against the float64 oracle (chunked): every layer elementwise at 1e-5·scale from the
    float32-rounded oracle input of that layer, and the free-running chain normwise at 1e-5 (tests/util.py::check_chain)."""
    g, csc = _graph(gn, bench.make_c2)
    rng = np.random.default_rng(110)
    core = (128, 64, 32)
    pe, pd = O.make_block_params(rng, (10, 5, 0), core), O.make_block_params(rng, core, (3, 4, 5))
    pcs = [O.make_core_params(rng, core) for _ in range(2)]
    layers = [("block", pe, U.block_from_params(gn, pe))] + [("core", p, U.core_from_params(gn, p)) for p in pcs] + [("block", pd, U.block_from_params(gn, pd))]
    ef, nf, _ = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, (10, 5, 0))
    stats = U.check_chain(gn, g, csc, layers, (ef, nf, None), "config 4 at 1M edges")
    print("config 4, 1M edges: worst layer-wise ratio to 1e-5*scale / end-to-end normwise error:", stats)


@pytest.mark.parametrize("which", ["c2", "c3"])
def test_core_dims_block_full_size(gn, which):
    """The matrix-core path at BASELINE sizes: GNBlock (128,64,32)=>(128,64,32) on the 1M-edge C2 graph (one graph, 7813 row
    tiles, 8 column tiles per XCD group) and on the 512-graph C3 batch, elementwise at 1e-5·scale."""
    g, csc = _graph(gn, bench.make_c2 if which == "c2" else (lambda: bench.make_hetero(3)))
    rng = np.random.default_rng(111)
    dims = ((128, 64, 32), (128, 64, 32))
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    y = U.block_from_params(gn, p)(U.to_nt(gn, g, ef, nf, gf))
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name, log=f"core dims (128,64,32)=>(128,64,32) on {which.upper()} (matrix-core path)")

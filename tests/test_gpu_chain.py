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
    mk = lambda layers: gn.Chain([gn.Dense.from_numpy(W, b, U.ACT_NAMES[a]) for W, b, a in layers])
    blk.edgefn, blk.nodefn, blk.graphfn = mk(p["edge"]), mk(p["node"]), mk(p["graph"])
    return blk


CASES = [
    ((10, 5, 0), [16, 3], [8, 4], [6, 5]),          # README widths with one hidden layer everywhere
    ((10, 5, 3), [32, 24, 7], [12], [9, 4]),        # three edge layers, one-layer node function
    ((0, 4, 2), [8, 5], [5], []),                   # ef = nothing, no graph output
    ((6, 0, 0), [4], [], []),                       # plain one-layer edge function only
    ((128, 64, 32), [256, 128], [128, 64], [64, 32]),  # wide: every layer on the matrix cores
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
    # a graph function without a node function has no getgraphfninput (graphfninput.jl:1-13)
    bad = O.make_chain_block_params(rng, (10, 5, 0), [4, 3], [], [5])
    with pytest.raises(gn.GnxError):
        _block(gn, bad)(gn.batch(dict(graphs=adj, ef=ef, nf=nf, gf=None)))

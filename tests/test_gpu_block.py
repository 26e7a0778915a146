"""GPU parity of the HIP GNBlock forward against the float64 oracle, through the C ABI (libgnx.so).

Mirrors the reference's tests (/root/reference/test/runtests.jl): README examples and shapes (:118-326, :627-652),
batch invariance (:62-116), batch/unbatch identity (:328-390), all `nothing` combinations (:519-625)."""
import itertools

import numpy as np
import pytest
import torch

from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu

README_ADJ = np.array([[1, 0, 1], [1, 1, 0], [0, 0, 1]])
README_ADJ2 = np.array([[1, 0, 1, 0], [1, 1, 0, 1], [0, 0, 1, 0], [1, 1, 0, 1]])


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


PATHS = [pytest.param(0, id="default"), pytest.param(1, id="generic")]  # GNX_FLAG_FORCE_GENERIC = 1


def _check_block(gn, p, g, csc, ef, nf, gf, flags):
    blk = U.block_from_params(gn, p)
    y = blk(U.to_nt(gn, g, ef, nf, gf), flags=flags)
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name)
    return y


def _csc_of(g):
    colptr, rowval = g.csc()
    return colptr, rowval, g.node_off, g.edge_off


@pytest.mark.parametrize("flags", PATHS)
def test_readme_example_1(gn, flags):
    """README ex.1 / BASELINE config 1: shared 3-node/5-edge adjacency, batch_size 2, (10,5,0)=>(3,4,5)."""
    rng = np.random.default_rng(1)
    p = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    ef = rng.random((10, 5, 2), dtype=np.float32)
    nf = rng.random((5, 3, 2), dtype=np.float32)
    x = gn.batch(dict(graphs=README_ADJ, ef=ef, nf=nf, gf=None))
    blk = U.block_from_params(gn, p)
    blk.flags = flags
    y = gn.unbatch(blk(x))
    assert tuple(y.ef.shape) == (3, 5, 2) and tuple(y.nf.shape) == (4, 3, 2) and tuple(y.gf.shape) == (5, 2)
    ref = O.unbatch_dense(O.block_forward_dense(p, O.batch_dense(README_ADJ, ef, nf, None)))
    np.testing.assert_allclose(y.ef.cpu().numpy(), ref["ef"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.nf.cpu().numpy(), ref["nf"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.gf.cpu().numpy(), ref["gf"], rtol=1e-5, atol=1e-5)
    # views (views.jl): first batch element
    assert tuple(gn.efview(blk(x), slice(None), slice(None), 0).shape) == (3, 5)
    np.testing.assert_allclose(gn.nfview(blk(x), slice(None), slice(None), 1).cpu().numpy(), ref["nf"][:, :, 1], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gn.gfview(blk(x), slice(None), 1).cpu().numpy(), ref["gf"][:, 1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("flags", PATHS)
def test_readme_example_2_vector_of_graphs(gn, flags):
    """README ex.2: two graphs of different structure; outputs are vectors of per-graph arrays."""
    rng = np.random.default_rng(2)
    adjs = [README_ADJ, README_ADJ2]
    p = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    ef = [rng.random((10, int(a.sum())), dtype=np.float32) for a in adjs]
    nf = [rng.random((5, a.shape[0]), dtype=np.float32) for a in adjs]
    x = gn.batch(dict(graphs=adjs, ef=ef, nf=nf, gf=None))
    assert x.graphs.node_block_size == 4 and x.graphs.edge_block_size == 16
    blk = U.block_from_params(gn, p)
    blk.flags = flags
    yb = blk(x)
    y = gn.unbatch(yb)
    ref = O.unbatch_dense(O.block_forward_dense(p, O.batch_dense(adjs, ef, nf, None)))
    for i in range(2):
        assert tuple(y.ef[i].shape) == (3, int(adjs[i].sum())) and tuple(y.nf[i].shape) == (4, adjs[i].shape[0])
        np.testing.assert_allclose(y.ef[i].cpu().numpy(), ref["ef"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(y.nf[i].cpu().numpy(), ref["nf"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(y.gf[i].cpu().numpy(), ref["gf"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gn.efview(yb, slice(None), slice(None), i).cpu().numpy(), ref["ef"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gn.gfview(yb, slice(None), i).cpu().numpy(), ref["gf"][i], rtol=1e-5, atol=1e-5)
    # flatunpadded* == graph-major packed (views.jl:80-98) and padded() == the reference's padded arrays on real slots
    dense = O.block_forward_dense(p, O.batch_dense(adjs, ef, nf, None))
    np.testing.assert_allclose(gn.flatunpaddednf(yb).cpu().numpy(), O.flat_from_dense(dense, "nf"), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gn.flatunpaddedef(yb).cpu().numpy(), O.flat_from_dense(dense, "ef"), rtol=1e-5, atol=1e-5)
    pad = gn.padded(yb)
    assert tuple(pad.ef.shape) == (3, 16, 2) and tuple(pad.nf.shape) == (4, 4, 2) and tuple(pad.gf.shape) == (5, 1, 2)
    em = dense["graphs"].flat_edge_unpadder.reshape(16, 2, order="F")
    pe = pad.ef.cpu().numpy()
    for b in range(2):
        np.testing.assert_allclose(pe[:, em[:, b], b], dense["ef"][:, em[:, b], b], rtol=1e-5, atol=1e-5)
        assert np.all(pe[:, ~em[:, b], b] == 0)
    assert np.array_equal(x.graphs.flat_edge_unpadder, dense["graphs"].flat_edge_unpadder)
    assert np.array_equal(x.graphs.flat_node_unpadder, dense["graphs"].flat_node_unpadder)

def test_unpadded_takes_the_references_padded_arrays_back_to_the_batch_form(gn):
    """unpad.jl:1-17 (`unpadef`, `unpadnf`, `unpadgf`) over gnx_unpad_features: the reference's padded batched arrays (the oracle's dense
    restatement builds them, pad.jl:12-64) with JUNK in the pads -> exactly what `batch` makes of the per-graph arrays, for a vector of
    graphs and for the shared-adjacency form (three replicas); `unpadded(padded(x)) == x` bit for bit."""
    rng = np.random.default_rng(77)
    adjs = [README_ADJ, README_ADJ2, (rng.random((7, 7)) < 0.4).astype(np.float32)]
    ef = [rng.standard_normal((6, int(a.sum()))).astype(np.float32) for a in adjs]
    nf = [rng.standard_normal((3, a.shape[0])).astype(np.float32) for a in adjs]
    gf = [rng.standard_normal(2).astype(np.float32) for _ in adjs]
    x = gn.batch(dict(graphs=adjs, ef=ef, nf=nf, gf=gf))
    dense = O.batch_dense(adjs, ef, nf, gf)
    g = dense["graphs"]
    PN, B = g.padded_adj_mats.shape[0], len(adjs)
    em = g.flat_edge_unpadder.reshape(PN * PN, B, order="F")
    nm = g.flat_node_unpadder.reshape(PN, B, order="F")
    pe, pn = dense["ef"].astype(np.float32), dense["nf"].astype(np.float32)
    for b in range(B):  # what the reference's padded arrays hold in the pads after a block: act(bias) junk
        pe[:, ~em[:, b], b] = 9.5
        pn[:, ~nm[:, b], b] = -3.25
    y = gn.unpadded(x.graphs, ef=pe, nf=pn, gf=dense["gf"].astype(np.float32))
    assert torch.equal(y.ef, x.ef) and torch.equal(y.nf, x.nf) and torch.equal(y.gf, x.gf)
    z = gn.unpadded(x.graphs, *list(gn.padded(x))[1:])
    assert torch.equal(z.ef, x.ef) and torch.equal(z.nf, x.nf) and torch.equal(z.gf, x.gf)
    only = gn.unpadded(x.graphs, nf=pn)
    assert only.ef is None and only.gf is None and torch.equal(only.nf, x.nf)
    with pytest.raises(AssertionError):
        gn.unpadded(x.graphs, ef=pe[:, :-1, :])
    # shared adjacency, three replicas (pad.jl:30-41)
    efs = rng.standard_normal((6, int(README_ADJ.sum()), 3)).astype(np.float32)
    nfs = rng.standard_normal((3, README_ADJ.shape[0], 3)).astype(np.float32)
    gfs = rng.standard_normal((2, 3)).astype(np.float32)
    xs = gn.batch(dict(graphs=README_ADJ, ef=efs, nf=nfs, gf=gfs))
    ds = O.batch_dense(README_ADJ, efs, nfs, gfs)
    ys = gn.unpadded(xs.graphs, ef=ds["ef"].astype(np.float32), nf=ds["nf"].astype(np.float32), gf=ds["gf"].astype(np.float32))
    assert torch.equal(ys.ef, xs.ef) and torch.equal(ys.nf, xs.nf) and torch.equal(ys.gf, xs.gf)
    zs = gn.unpadded(xs.graphs, *list(gn.padded(xs))[1:])
    assert torch.equal(zs.ef, xs.ef) and torch.equal(zs.nf, xs.nf) and torch.equal(zs.gf, xs.gf)



IN_COMBOS = [d for d in itertools.product((0, 3), (0, 2), (0, 4)) if any(d)]


@pytest.mark.parametrize("flags", PATHS)
@pytest.mark.parametrize("in_dims", IN_COMBOS)
@pytest.mark.parametrize("out_dims", [(3, 4, 5), (2, 0, 3), (0, 2, 2), (2, 3, 0)])
def test_nothing_combinations(gn, in_dims, out_dims, flags):
    """7 edge / 4 node / 2 graph input forms (edgefninput.jl, nodefninput.jl, graphfninput.jl) and zero-width
    outputs → nothing (gnblock.jl:71-78); heterogeneous batch incl. an edgeless graph and a 1-node graph."""
    rng = np.random.default_rng(abs(hash((in_dims, out_dims))) % 2**31)
    adjs = U.random_graphs(rng, (5, 1, 9, 3, 14), 0.4)
    adjs[3][:] = 0  # graph without edges
    g = gn.GNGraphBatch(adjs)
    csc = O.csc_from_adj(adjs)
    assert np.array_equal(csc[0], g.csc()[0]) and np.array_equal(csc[1], g.csc()[1])
    p = O.make_block_params(rng, in_dims, out_dims, act=(1, 0, 2))
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, in_dims)
    y = _check_block(gn, p, g, csc, ef, nf, gf, flags)
    for got, d in zip((y.ef, y.nf, y.gf), out_dims):
        assert (got is None) == (d == 0)


@pytest.mark.parametrize("flags", PATHS)
@pytest.mark.parametrize("act", [(0, 0, 0), (1, 2, 3), (4, 4, 1)])
def test_activations_shared_batch(gn, act, flags):
    rng = np.random.default_rng(5 + sum(act))
    adj = U.random_graphs(rng, (23,), 0.3)[0]
    g = gn.GNGraphBatch([adj])
    p = O.make_block_params(rng, (6, 7, 3), (5, 6, 4), act=act)
    ef, nf, gf = U.packed_inputs(rng, 3, g.n_edges, g.n_nodes, 1, (6, 7, 3))
    _check_block(gn, p, g, O.csc_from_adj([adj]), ef, nf, gf, flags)


@pytest.mark.parametrize("flags", PATHS)
def test_batch_invariance(gn, flags):
    """runtests.jl:62-116 through the HIP path: A alone == A inside [A, B]; (0,2,0)→(2,2,2)→(2,2,2), ef=gf=nothing."""
    rng = np.random.default_rng(3)
    enc = U.block_from_params(gn, O.make_block_params(rng, (0, 2, 0), (2, 2, 2)))
    dec = U.block_from_params(gn, O.make_block_params(rng, (2, 2, 2), (2, 2, 2)))
    enc.flags = dec.flags = flags
    A, B = np.ones((2, 2), dtype=int), np.ones((3, 3), dtype=int)
    nfs = [rng.random((2, 2), dtype=np.float32), rng.random((2, 3), dtype=np.float32)]
    y1 = dec(enc(gn.batch(dict(graphs=[A], ef=None, nf=nfs[:1], gf=None))))
    yn = dec(enc(gn.batch(dict(graphs=[A, B], ef=None, nf=nfs, gf=None))))
    s = slice(None)
    for a, b in ((gn.nfview(y1, s, s, 0), gn.nfview(yn, s, s, 0)), (gn.efview(y1, s, s, 0), gn.efview(yn, s, s, 0)),
                 (gn.gfview(y1, s, 0), gn.gfview(yn, s, 0))):
        assert a.shape == b.shape
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_batch_inverse_2d_and_3d(gn):
    """runtests.jl:328-390: unbatch(batch(x)) == x exactly."""
    rng = np.random.default_rng(5)
    adjs = [README_ADJ, README_ADJ2]
    ef = [rng.random((10, int(a.sum())), dtype=np.float32) for a in adjs]
    nf = [rng.random((5, a.shape[0]), dtype=np.float32) for a in adjs]
    gf = [rng.random((3,), dtype=np.float32) for a in adjs]
    y = gn.unbatch(gn.batch(dict(graphs=adjs, ef=ef, nf=nf, gf=gf)))
    for got, ref in ((y.ef, ef), (y.nf, nf), (y.gf, gf)):
        for a, b in zip(got, ref):
            assert np.array_equal(a.cpu().numpy(), b)
    ef, nf, gf = rng.random((10, 5, 2), dtype=np.float32), rng.random((5, 3, 2), dtype=np.float32), rng.random((3, 2), dtype=np.float32)
    y = gn.unbatch(gn.batch(dict(graphs=README_ADJ, ef=ef, nf=nf, gf=gf)))
    assert np.array_equal(y.ef.cpu().numpy(), ef) and np.array_equal(y.nf.cpu().numpy(), nf) and np.array_equal(y.gf.cpu().numpy(), gf)
    assert np.array_equal(y.graphs, README_ADJ)


def test_no_graph_features_output(gn):
    """runtests.jl:118-164: (10,5,0)=>(3,4,0) → y.gf === nothing."""
    rng = np.random.default_rng(8)
    blk = gn.GNBlock((10, 5, 0), (3, 4, 0))
    x = gn.batch(dict(graphs=README_ADJ, ef=rng.random((10, 5, 2), dtype=np.float32), nf=rng.random((5, 3, 2), dtype=np.float32), gf=None))
    y = gn.unbatch(blk(x))
    assert tuple(y.ef.shape) == (3, 5, 2) and tuple(y.nf.shape) == (4, 3, 2) and y.gf is None


def test_checks_mirror_reference_assertions(gn):
    rng = np.random.default_rng(9)
    with pytest.raises(AssertionError):  # batch.jl:56
        gn.batch(dict(graphs=README_ADJ, ef=None, nf=None, gf=None))
    with pytest.raises(AssertionError):  # checks.jl:44 wrong edge count
        gn.batch(dict(graphs=README_ADJ, ef=rng.random((10, 4, 2), dtype=np.float32), nf=None, gf=None))
    with pytest.raises(AssertionError):  # checks.jl:45 wrong node count
        gn.batch(dict(graphs=[README_ADJ], ef=None, nf=[rng.random((5, 4), dtype=np.float32)], gf=None))
    with pytest.raises(AssertionError):  # gnblock.jl:48
        gn.GNBlock((0, 0, 0), (1, 1, 1))
    with pytest.raises(gn.GnxError) as e:  # pad.jl:30 / gngraphbatch.jl:207: entries must be 0/1
        gn.GNGraphBatch([np.array([[1, 2], [0, 1]])])
    assert e.value.code == -4


@pytest.mark.parametrize("flags", PATHS)
@pytest.mark.parametrize("dims", [((10, 5, 0), (3, 4, 5)), ((8, 8, 8), (16, 8, 4))])
def test_er_graph_many_tiles(gn, dims, flags):
    """One 2000-node / 20000-edge Erdős–Rényi graph (C2's recipe, scaled): many tiles, two replicas."""
    rng = np.random.default_rng(12)
    colptr, rowval = U.er_csc(rng, 2000, 20000)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [2000])
    assert g.n_tiles > 10
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 2, 20000, 2000, 1, dims[0])
    _check_block(gn, p, g, _csc_of(g), ef, nf, gf, flags)


@pytest.mark.parametrize("flags", PATHS)
def test_skewed_degrees_and_isolated_nodes(gn, flags):
    """A hub with 3000 in-edges (more than one tile's edge capacity), nodes without in-edges, self loops."""
    rng = np.random.default_rng(13)
    N = 3200
    colptr = np.zeros(N + 1, dtype=np.int64)
    rows = []
    for j in range(N):
        if j == 7:
            r = np.sort(rng.choice(N, 3000, replace=False))
        elif j % 5 == 0:
            r = np.zeros(0, dtype=np.int64)
        else:
            r = np.sort(rng.choice(N, rng.integers(1, 6), replace=False))
        rows.append(r)
        colptr[j + 1] = colptr[j] + len(r)
    rowval = np.concatenate(rows).astype(np.int64)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [N])
    assert g.max_in_degree == 3000
    p = O.make_block_params(rng, (4, 3, 2), (3, 4, 5))
    ef, nf, gf = U.packed_inputs(rng, 1, len(rowval), N, 1, (4, 3, 2))
    _check_block(gn, p, g, _csc_of(g), ef, nf, gf, flags)


@pytest.mark.parametrize("flags", PATHS)
def test_heterogeneous_batch_64_graphs(gn, flags):
    """BASELINE config 3 scaled down: 64 random graphs, 32..256 nodes each, GNGraphBatch padding path."""
    rng = np.random.default_rng(14)
    sizes = rng.integers(32, 257, 64)
    colptrs, rowvals = [], []
    for n in sizes:
        cp, rv = U.er_csc(rng, int(n), int(0.05 * n * n))
        colptrs.append(cp); rowvals.append(rv)
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, [int(n) for n in sizes])
    assert g.node_block_size == sizes.max()
    p = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, (10, 5, 0))
    _check_block(gn, p, g, _csc_of(g), ef, nf, gf, flags)


@pytest.mark.parametrize("dims", [((10, 5, 0), (3, 4, 5)), ((10, 5, 3), (3, 4, 5)), ((4, 3, 2), (3, 4, 5))])
def test_small_graph_batches_run_the_graph_update_inside_the_block_kernel(gn, dims, monkeypatch):
    """Batches whose graphs all have <= 8 wave tiles (here 700 graphs of 1..300 nodes, some without a single edge): workgroups own whole
    graphs and run the graph update themselves (k_block_wave<..., PACK>) — ONE launch.  Every output is bit-identical to the two-launch
    form (GNX_FLAG_NO_PACK: partial rows in HBM + k_graph_t) and within 1e-5*S of the oracle."""
    rng = np.random.default_rng(77)
    sizes = np.concatenate([rng.integers(1, 301, 690), [1, 1, 2, 300, 300, 64, 65, 128, 129, 3]])
    colptrs, rowvals = [], []
    for i, n in enumerate(sizes):
        n = int(n)
        e = 0 if i % 50 == 0 else min(n * n, int(rng.integers(0, 3 * n + 1)))
        cp, rv = U.er_csc(rng, n, e) if e else (np.zeros(n + 1, np.int64), np.zeros(0, np.int64))
        colptrs.append(cp); rowvals.append(rv)
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, [int(n) for n in sizes])
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    blk = U.block_from_params(gn, p)
    x = U.to_nt(gn, g, ef, nf, gf)
    gn.profile_reset(); gn.profile_enable(True)
    y1 = blk(x)
    gn.profile_enable(False)
    kernels = set(gn.profile_read()); gn.profile_reset()
    assert "k_block_wave" in kernels and "k_graph_t" not in kernels, kernels  # one launch
    gn.profile_enable(True)
    y2 = blk(x, flags=gn._lib.FLAG_NO_PACK)
    gn.profile_enable(False)
    assert "k_graph_t" in set(gn.profile_read()); gn.profile_reset()  # the two-launch form
    for u, v in ((y1.ef, y2.ef), (y1.nf, y2.nf), (y1.gf, y2.gf)):
        assert np.array_equal(u.cpu().numpy(), v.cpu().numpy())
    ref, scale = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
    for name, got, r_, s_ in zip(("ef", "nf", "gf"), (y1.ef, y1.nf, y1.gf), ref, scale):
        U.assert_close(U.from_jl(got), r_, s_, name)


def test_results_are_bitwise_reproducible(gn):
    """Atomic-free, fixed-order reductions: two runs give identical bits."""
    rng = np.random.default_rng(15)
    colptr, rowval = U.er_csc(rng, 1000, 12000)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [1000])
    blk = U.block_from_params(gn, O.make_block_params(rng, (10, 5, 0), (3, 4, 5)))
    ef, nf, gf = U.packed_inputs(rng, 1, 12000, 1000, 1, (10, 5, 0))
    x = U.to_nt(gn, g, ef, nf, gf)
    a, b = blk(x), blk(x)
    for u, v in ((a.ef, b.ef), (a.nf, b.nf), (a.gf, b.gf)):
        assert np.array_equal(u.cpu().numpy(), v.cpu().numpy())


@pytest.mark.parametrize("in_dims", IN_COMBOS)
def test_exported_fn_input_building_blocks(gn, in_dims):
    """getedgefninput / getnodefninput / getgraphfninput (exported at src/GraphNets.jl:26-32) against the dense-form
    oracle's literal vcat/batched_mul restatement, on the real (unpadded) slots, for every `nothing` combination."""
    rng = np.random.default_rng(60 + sum(in_dims))
    adjs = U.random_graphs(rng, (4, 7, 2), 0.5)
    de, dn, dg = in_dims
    ef = [rng.random((de, int(a.sum())), dtype=np.float32) for a in adjs] if de else None
    nf = [rng.random((dn, a.shape[0]), dtype=np.float32) for a in adjs] if dn else None
    gf = [rng.random((dg,), dtype=np.float32) for a in adjs] if dg else None
    x = gn.batch(dict(graphs=adjs, ef=ef, nf=nf, gf=gf))
    xd = O.batch_dense(adjs, ef, nf, gf)
    g = xd["graphs"]
    em = g.flat_edge_unpadder.reshape(g.edge_block_size, len(adjs), order="F")
    nm = g.flat_node_unpadder.reshape(g.node_block_size, len(adjs), order="F")
    got = gn.getedgefninput(x.graphs, x.ef, x.nf, x.gf).cpu().numpy()[:, :, 0]
    ref = O.getedgefninput_dense(g, xd["ef"], xd["nf"], xd["gf"]) if de else O._vcat(
        ([O.batched_mul(xd["nf"], g.srcnode2edge), O.batched_mul(xd["nf"], g.dstnode2edge)] if dn else []) +
        ([O.batched_mul(xd["gf"], g.graph2edge)] if dg else []))
    ref_flat = np.concatenate([ref[:, em[:, b], b] for b in range(len(adjs))], axis=1)
    np.testing.assert_allclose(got, ref_flat, rtol=1e-6, atol=1e-6)
    if de and dn:  # node / graph inputs take the (updated) edge and node features: reuse ef, nf as stand-ins
        got = gn.getnodefninput(x.graphs, x.ef, x.nf, x.gf).cpu().numpy()[:, :, 0]
        ref = O.getnodefninput_dense(g, xd["ef"], xd["nf"], xd["gf"])
        np.testing.assert_allclose(got, np.concatenate([ref[:, nm[:, b], b] for b in range(len(adjs))], axis=1), rtol=1e-5, atol=1e-5)
        got = gn.getgraphfninput(x.graphs, x.ef, x.nf, x.gf).cpu().numpy()[:, :, 0]
        ref = O.getgraphfninput_dense(g, xd["ef"], xd["nf"], xd["gf"])
        np.testing.assert_allclose(got, ref[:, 0, :], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dims", [((10, 5, 0), (3, 4, 5)), ((128, 64, 32), (128, 64, 32)), ((5, 6, 7), (7, 6, 5))], ids=["narrow", "wide", "generic"])
def test_deferred_graph_update_on_second_stream(gn, dims):
    """GNX_FLAG_DEFER_GRAPH_UPDATE + gnx_block_graph_update (second stream, event-ordered) == the one-call forward, bitwise."""
    import torch
    rng = np.random.default_rng(70)
    colptr, rowval = U.er_csc(rng, 500, 4000)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [500])
    blk = U.block_from_params(gn, O.make_block_params(rng, *dims))
    ef, nf, gf = U.packed_inputs(rng, 1, 4000, 500, 1, dims[0])
    dev = g.device
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    ef, nf, gf = t(ef), t(nf), t(gf)
    plan = gn.BlockPlan(blk, g)
    o1, o2 = plan.outputs(), plan.outputs()
    plan(ef, nf, gf, *o1)
    ws2 = plan.new_workspace()
    side = torch.cuda.Stream(device=dev)
    plan(ef, nf, gf, *o2, ws=ws2, defer_graph_update=True)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        plan.graph_update(gf, o2[2], ws=ws2)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)


def test_edge_collapsing(gn):
    """test/runtests.jl:4-59: flatunpaddedcollapsedef on fully connected 2- and 3-node graphs after two stacked blocks;
    the reference indexes the raw padded array (`ef[:, slot, graph]`, 1-based slots of the PN^2 grid) — here through
    padded()."""
    rng = np.random.default_rng(90)
    enc, dec = gn.GNBlock((0, 2, 0), (2, 2, 2)), gn.GNBlock((2, 2, 2), (2, 2, 2))
    A, B = np.ones((2, 2), dtype=int), np.ones((3, 3), dtype=int)
    nf = [rng.random((2, 2), dtype=np.float32), rng.random((2, 3), dtype=np.float32)]
    y = dec(enc(gn.batch(dict(graphs=[A, B], ef=None, nf=nf, gf=None))))
    flat = gn.flatunpaddedcollapsedef(y).cpu().numpy()
    ef = gn.padded(y).ef.cpu().numpy()  # (2, 9, 2), PN = 3
    s = lambda slot, g: ef[:, slot - 1, g]  # 1-based slot as in the reference test
    assert flat.shape == (2, 9)
    expect = [s(1, 0), (s(2, 0) + s(4, 0)) / 2, s(5, 0),
              s(1, 1), (s(2, 1) + s(4, 1)) / 2, (s(3, 1) + s(7, 1)) / 2, s(5, 1), (s(6, 1) + s(8, 1)) / 2, s(9, 1)]
    for c, e in enumerate(expect):
        np.testing.assert_allclose(flat[:, c], e, rtol=1e-6, atol=1e-6)
    per_graph = gn.unpaddedcollapsedef(y)
    assert [tuple(a.shape) for a in per_graph] == [(2, 3), (2, 6)]
    # a graph with a one-way edge: the missing reverse contributes 0
    adj = np.array([[1, 0], [1, 0]])  # edges 0->0, 1->0 ; lower triangle: (0,0) and (1,0); reverse of 1->0 is 0->1: absent
    x = gn.batch(dict(graphs=adj, ef=np.array([[[1.0], [4.0]]], dtype=np.float32).reshape(1, 2, 1), nf=None, gf=None))
    np.testing.assert_allclose(gn.flatunpaddedcollapsedef(x).cpu().numpy(), [[1.0, 2.0]])


def test_collapsef_padded_array_form(gn):
    """collapsef (gngraphbatch.jl:83-85) as the padded (DE, PN(PN+1)/2, B) array: against the oracle's literal edge_collapser matmul
    on random adjacency matrices of different sizes (one-way edges, missing self loops, pad rows), vector and shared mode."""
    rng = np.random.default_rng(92)
    adjs = [(rng.random((n, n)) < 0.5).astype(int) for n in (1, 4, 7, 3)]
    efs = [rng.random((3, int(a.sum())), dtype=np.float32) for a in adjs]
    got = gn.collapsef(gn.batch(dict(graphs=adjs, ef=efs, nf=None, gf=None))).cpu().numpy()
    ref = O.collapsef_dense(O.batch_dense(adjs, efs, None, None))
    assert got.shape == ref.shape == (3, 28, 4)
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7)
    adj = adjs[2]
    ef = rng.random((3, int(adj.sum()), 5), dtype=np.float32)  # shared adjacency, batch of 5
    got = gn.collapsef(gn.batch(dict(graphs=adj, ef=ef, nf=None, gf=None))).cpu().numpy()
    ref = O.collapsef_dense(O.batch_dense(adj, ef, None, None))
    assert got.shape == ref.shape == (3, 28, 5)
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7)
    # unpaddedcollapsedef = the real lower-triangle columns of collapsef
    x = gn.batch(dict(graphs=adjs, ef=efs, nf=None, gf=None))
    full = gn.collapsef(x).cpu().numpy()
    idxs = O.getcollapsededgeidxs(O.padadjmats([a.astype(float) for a in adjs]))
    for b, part in enumerate(gn.unpaddedcollapsedef(x)):
        np.testing.assert_allclose(part.cpu().numpy(), full[:, idxs[b], b], rtol=1e-6, atol=1e-7)


def test_device_side_batch_construction(gn):
    """large dense batches are scanned/compacted into CSC on the GPU (gnx_build_device.hip).  The result must
    equal the oracle's column-major edge order (pad.jl:30) for row-major (numpy) and column-major (Julia) input alike,
    and non-0/1 entries must still be rejected (pad.jl:30, gngraphbatch.jl:207)."""
    import ctypes as C
    rng = np.random.default_rng(95)
    sizes = rng.integers(32, 257, 200)
    adjs = [(rng.random((n, n)) < 0.07).astype(np.float32) for n in sizes]
    assert sum(a.size for a in adjs) >= (1 << 22)  # above the device-path threshold
    g = gn.GNGraphBatch(adjs)
    colptr, rowval, node_off, edge_off = O.csc_from_adj(adjs)
    gc, gr = g.csc()
    assert np.array_equal(gc, colptr) and np.array_equal(gr, rowval)
    assert np.array_equal(g.node_off, node_off) and np.array_equal(g.edge_off, edge_off)
    # column-major input (what Julia hands over): pass the transposes with row_major = 0
    lib = gn._lib.load()
    tr = [np.ascontiguousarray(a.T) for a in adjs]
    ptrs = (C.c_void_p * len(tr))(*[a.ctypes.data for a in tr])
    nn = np.asarray([a.shape[0] for a in tr], dtype=np.int64)
    h = C.c_void_p(None)
    gn._lib.check(lib.gnx_graphs_create_dense(ptrs, nn.ctypes.data_as(C.POINTER(C.c_int64)), len(tr), gn._lib.ELEM_F32, 0, C.byref(h)))
    cp2 = np.zeros(len(colptr), dtype=np.int64); rv2 = np.zeros(len(rowval), dtype=np.int64)
    gn._lib.check(lib.gnx_graphs_get_csc(h, cp2.ctypes.data_as(C.POINTER(C.c_int64)), rv2.ctypes.data_as(C.POINTER(C.c_int64))))
    lib.gnx_graphs_destroy(h)
    assert np.array_equal(cp2, colptr) and np.array_equal(rv2, rowval)
    adjs[57][3, 4] = 2.0
    with pytest.raises(gn.GnxError) as e:
        gn.GNGraphBatch(adjs)
    assert e.value.code == -4
    # and a forward on the device-built handle
    adjs[57][3, 4] = 1.0
    g = gn.GNGraphBatch(adjs)
    csc = O.csc_from_adj(adjs)
    p = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, (10, 5, 0))
    _check_block(gn, p, g, csc, ef, nf, gf, 0)


def test_readout_logitcrossentropy(gn):
    """SURVEY 8f f2 / examples/sort/sort.jl:69-81: loss = logitcrossentropy(flatunpaddednf(ŷ), y_nf) + (same on ef)."""
    import torch
    rng = np.random.default_rng(96)
    adjs = U.random_graphs(rng, (6, 9, 4), 0.5)
    blk = gn.GNBlock((0, 3, 0), (4, 5, 0))
    nf = [rng.random((3, a.shape[0]), dtype=np.float32) for a in adjs]
    y = blk(gn.batch(dict(graphs=adjs, ef=None, nf=nf, gf=None)))
    for flat in (gn.flatunpaddednf(y), gn.flatunpaddedef(y)):
        d, cols = flat.shape
        tgt = np.eye(d, dtype=np.float32)[rng.integers(0, d, cols)].T  # one-hot columns
        got = float(gn.logitcrossentropy(flat, torch.from_numpy(tgt)))
        x = flat.double().cpu().numpy()
        lse = np.log(np.exp(x - x.max(0)).sum(0)) + x.max(0)
        ref = float(np.mean(-(tgt * (x - lse)).sum(0)))
        assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref))


@pytest.mark.parametrize("case", ["one-graph", "many-graphs", "small-graphs(pack form)", "wide(matrix cores)", "runtime-specialised"])
def test_chained_forward_runs_the_previous_graph_update_inside_the_next_launch(gn, case):
    """gnx_block_forward_chained: a loop over batches whose step is ONE launch — the edge + node update of batch i with the graph update of
    batch i - 1 in workgroups at the front of the same kernel (k_block_wave<..., CHAIN>) — and a flush at the end.  Every output of every
    step is BIT-identical to gnx_block_forward's; where the two-launch narrow form is not what runs the call falls back to the plain
    form and leaves nothing pending."""
    import torch
    rng = np.random.default_rng(300 + len(case))
    dims = ((10, 5, 0), (3, 4, 5))
    if case == "one-graph":
        cp, rv = U.er_csc(rng, 30_000, 200_000)
        cps, rvs, nn = [cp], [rv], [30_000]
    elif case == "many-graphs":  # graphs with more than 8 wave tiles: the two-launch form, one wavefront per graph in the chained blocks
        sizes = rng.integers(300, 700, 37)
        cs = [U.er_csc(rng, int(n), 12 * int(n)) for n in sizes]
        cps, rvs, nn = [c[0] for c in cs], [c[1] for c in cs], [int(n) for n in sizes]
    elif case == "small-graphs(pack form)":
        sizes = rng.integers(20, 90, 200)
        cs = [U.er_csc(rng, int(n), 4 * int(n)) for n in sizes]
        cps, rvs, nn = [c[0] for c in cs], [c[1] for c in cs], [int(n) for n in sizes]
    else:
        cp, rv = U.er_csc(rng, 3_000, 20_000)
        cps, rvs, nn = [cp], [rv], [3_000]
        dims = ((40, 36, 8), (36, 40, 8)) if case.startswith("wide") else ((7, 3, 2), (5, 6, 1))
    g = gn.GNGraphBatch.from_csc(cps, rvs, nn)
    p = O.make_block_params(rng, *dims)
    blk = U.block_from_params(gn, p)
    plan = gn.BlockPlan(blk, g)
    dev = g.device
    (de, dn, dg), _ = dims
    mk = lambda T, d: torch.rand((1, T, d), device=dev) if d > 0 else None
    steps = 7
    ins = [(mk(g.n_edges, de), mk(g.n_nodes, dn), mk(g.n_graphs, dg)) for _ in range(steps)]
    ref = []
    for ef, nf, gf in ins:  # the plain form
        out = plan.outputs()
        plan(ef, nf, gf, *out)
        ref.append(out)
    torch.cuda.synchronize()
    outs = [plan.outputs() for _ in range(steps)]
    wss = [plan.new_workspace(), plan.new_workspace()]
    pending = None
    took = []
    for i, (ef, nf, gf) in enumerate(ins):
        pending = plan.chained(ef, nf, gf, *outs[i], ws=wss[i & 1], prev=pending)
        took.append(bool(pending.workspace))
    plan.flush(pending)
    torch.cuda.synchronize()
    expect_chain = case in ("one-graph", "many-graphs")
    assert all(t == expect_chain for t in took), (case, took)
    for i in range(steps):
        for name, a, b in zip(("ef", "nf", "gf"), outs[i], ref[i]):
            assert torch.equal(a, b), f"{case}: step {i} {name} differs between the chained and the plain form"
    # the same workspace for two consecutive chained calls is refused (its graph update has not run yet)
    if expect_chain:
        pend = plan.chained(*ins[0], *outs[0], ws=wss[0], prev=None)
        with pytest.raises(gn._lib.GnxError):
            plan.chained(*ins[1], *outs[1], ws=wss[0], prev=pend)
        plan.flush(pend)
        torch.cuda.synchronize()


@pytest.mark.parametrize("case", ["one-graph", "small-graphs(pack form)", "wide(matrix cores)"])
def test_forward_steps_is_a_loop_of_forwards_in_one_call(gn, case):
    """gnx_block_forward_steps (round 6; bench.py's headline form): n gnx_block_forward calls in order as ONE call — outputs bit-identical; where the
    two-launch narrow form runs the library chains the steps itself (n block launches + ONE graph-update launch instead of 2 n launches: counted
    through the profiler's scopes), every gf' complete when the call's work is; steps that share a workspace with their predecessor run unchained;
    capture-safe (a hipGraph of the call replays to the same bits); argument errors are reported after the pending step was finished."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(900 + len(case))
    dims = ((10, 5, 0), (3, 4, 5))
    if case == "one-graph":
        cp, rv = U.er_csc(rng, 30_000, 200_000)
        cps, rvs, nn = [cp], [rv], [30_000]
    elif case.startswith("small"):
        sizes = rng.integers(20, 90, 200)
        cs = [U.er_csc(rng, int(n), 4 * int(n)) for n in sizes]
        cps, rvs, nn = [c[0] for c in cs], [c[1] for c in cs], [int(n) for n in sizes]
    else:
        cp, rv = U.er_csc(rng, 3_000, 20_000)
        cps, rvs, nn = [cp], [rv], [3_000]
        dims = ((40, 36, 8), (36, 40, 8))
    g = gn.GNGraphBatch.from_csc(cps, rvs, nn)
    blk = U.block_from_params(gn, O.make_block_params(rng, *dims))
    plan = gn.BlockPlan(blk, g)
    dev = g.device
    (de, dn, dg), _ = dims
    mk = lambda T, d: torch.rand((1, T, d), device=dev) if d > 0 else None
    n = 6
    sets = [dict(ef=mk(g.n_edges, de), nf=mk(g.n_nodes, dn), gf=mk(g.n_graphs, dg), out=plan.outputs(), ws=plan.new_workspace()) for _ in range(n)]
    ref = []
    for b in sets:
        out = plan.outputs()
        plan(b["ef"], b["nf"], b["gf"], *out)
        ref.append(out)
    torch.cuda.synchronize()
    gn.profile_reset(); gn.profile_enable(True)
    plan.steps(sets)
    torch.cuda.synchronize()
    gn.profile_enable(False)
    prof = gn.profile_read(); gn.profile_reset()
    for i, b in enumerate(sets):
        for name, a, r in zip(("ef", "nf", "gf"), b["out"], ref[i]):
            assert torch.equal(a, r), f"{case}: step {i} {name}"
    if case == "one-graph":  # chained: n block launches, ONE separate graph update (the flush)
        assert prof["k_block_wave"]["launches"] == n and prof["k_graph_t"]["launches"] == 1, prof
    # consecutive steps on ONE workspace (and one gf'): every step runs unchained, results unchanged
    for b in sets:
        for t in b["out"]:
            t.fill_(float("nan"))
    shared = [dict(b, ws=sets[0]["ws"]) for b in sets]
    gn.profile_enable(True)
    plan.steps(shared)
    torch.cuda.synchronize()
    gn.profile_enable(False)
    prof = gn.profile_read(); gn.profile_reset()
    for i, b in enumerate(sets):
        for a, r in zip(b["out"], ref[i]):
            assert torch.equal(a, r)
    if case == "one-graph":
        assert prof["k_graph_t"]["launches"] == n, prof
    # inside a hipGraph
    for b in sets:
        for t in b["out"]:
            t.fill_(float("nan"))
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        plan.steps(sets)
    cg.replay(); cg.replay()
    torch.cuda.synchronize()
    for i, b in enumerate(sets):
        for a, r in zip(b["out"], ref[i]):
            assert torch.equal(a, r)
    # zero steps: nothing to do; a NULL step table with n > 0 and the deferral flag are refused
    lib = gn._lib.load()
    s = torch.cuda.current_stream(dev).cuda_stream
    assert lib.gnx_block_forward_steps(g._h, C.byref(plan.p), None, 0, 1, 0, s) == 0
    assert lib.gnx_block_forward_steps(g._h, C.byref(plan.p), None, 2, 1, 0, s) == gn._lib.ERR_INVALID_ARG
    arr = (gn._lib.BlockStep * 1)()
    assert lib.gnx_block_forward_steps(g._h, C.byref(plan.p), arr, 1, 1, gn._lib.FLAG_DEFER_GRAPH_UPDATE, s) == gn._lib.ERR_INVALID_ARG
    # an invalid second step (workspace too small): the error comes back AND the first step is complete
    for t in sets[0]["out"]:
        t.fill_(float("nan"))
    two = (gn._lib.BlockStep * 2)()
    P = lambda t: None if t is None else t.data_ptr()
    for i in range(2):
        b = sets[i]
        two[i] = gn._lib.BlockStep(P(b["ef"]), P(b["nf"]), P(b["gf"]), P(b["out"][0]), P(b["out"][1]), P(b["out"][2]), b["ws"].data_ptr(), b["ws"].numel() if i == 0 else 16)
    assert lib.gnx_block_forward_steps(g._h, C.byref(plan.p), two, 2, 1, 0, s) == gn._lib.ERR_WORKSPACE
    torch.cuda.synchronize()
    for a, r in zip(sets[0]["out"], ref[0]):
        assert torch.equal(a, r)

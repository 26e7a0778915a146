"""Seeded randomised parity sweep: random width sets (0 = `nothing`, narrow, mid, wide), random batches (single nodes,
graphs without edges, hubs, replicas of a shared graph), random activations — every dispatch path (ahead-of-time fused
kernel, run-time specialised kernel, matrix-core path, generic kernels) against the float64 oracle at 1e-5 of the magnitude
bound (GNCore: the bound propagated through LayerNorm, block and FeedForward)."""
import numpy as np
import pytest

from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
import os

WIDTHS = [0, 1, 2, 3, 5, 8, 12, 16, 20, 24, 28, 32, 33, 40, 64]
EXTRA = int(os.environ.get("GNX_FUZZ_EXTRA", "0"))  # one-off deeper sweeps: GNX_FUZZ_EXTRA=300 python -m pytest tests/test_gpu_fuzz.py -m gpu


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


def _random_batch(rng, gn):
    cps, rvs, sizes, R = _random_csc(rng)
    return gn.GNGraphBatch.from_csc(cps, rvs, sizes), R


def _random_csc(rng):
    shared = rng.random() < 0.3
    G = 1 if shared else int(rng.integers(1, 7))
    cps, rvs, sizes = [], [], []
    for _ in range(G):
        n = int(rng.choice([1, 2, 3, 7, 20, 60, 150]))
        kind = rng.integers(0, 4)
        if kind == 0:                                   # no edges at all
            cp, rv = np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64)
        elif kind == 1:                                 # fully connected (the reference's sort example)
            cp, rv = np.arange(n + 1, dtype=np.int64) * n, np.tile(np.arange(n, dtype=np.int64), n)
        else:
            cp, rv = U.er_csc(rng, n, int(rng.integers(1, max(2, n * n // 3))))
        cps.append(cp); rvs.append(rv); sizes.append(n)
    R = int(rng.integers(1, 4)) if shared else 1
    return cps, rvs, sizes, R


def _dims(rng, core=False):
    if core:
        d = tuple(int(rng.choice([1, 3, 5, 8, 12, 16, 20, 33, 40, 64])) for _ in range(3))
        return d, d
    while True:
        din = tuple(int(rng.choice(WIDTHS)) for _ in range(3))
        dout = tuple(int(rng.choice(WIDTHS)) for _ in range(3))
        if sum(din) > 0 and sum(dout) > 0:
            return din, dout


@pytest.mark.parametrize("seed", range(60 + EXTRA))
def test_random_block(gn, seed):
    rng = np.random.default_rng(9000 + seed)
    g, R = _random_batch(rng, gn)
    din, dout = _dims(rng)
    p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.integers(0, 5, 3)))
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, din)
    csc = (*g.csc(), g.node_off, g.edge_off)
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    blk = U.block_from_params(gn, p)
    for flags in (0, 1, 2):  # default dispatch, generic kernels, no matrix cores
        y = blk(U.to_nt(gn, g, ef, nf, gf), flags=flags)
        for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
            U.assert_close(U.from_jl(got), r, s, f"seed {seed} dims {din}=>{dout} R={R} flags={flags} {name}")


@pytest.mark.parametrize("seed", range(16 + EXTRA // 4))
def test_random_core(gn, seed):
    rng = np.random.default_rng(9500 + seed)
    g, R = _random_batch(rng, gn)
    dims, _ = _dims(rng, core=True)
    p = O.make_core_params(rng, dims, eps_mode=int(rng.integers(0, 2)))
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, dims)
    if g.n_edges == 0:
        ef = np.zeros((R, 0, dims[0]), dtype=np.float32)
    csc = (*g.csc(), g.node_off, g.edge_off)
    ref, scale = O.core_forward_sparse(p, csc, ef, nf, gf, return_scale=True)  # 1e-5 of the scale propagated through LayerNorm's 1/σ
    core = U.core_from_params(gn, p)
    for flags in (0, 1):
        y = core(U.to_nt(gn, g, ef, nf, gf), flags=flags)
        for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
            U.assert_close(U.from_jl(got), r, s, f"seed {seed} dims {dims} R={R} flags={flags} {name}")


WIDE_BLOCKS = [((128, 64, 32), (128, 64, 32)), ((128, 64, 0), (128, 64, 0)), ((10, 5, 3), (128, 64, 32)), ((10, 5, 0), (128, 64, 32)),
               ((128, 64, 32), (3, 4, 2)), ((128, 64, 32), (10, 5, 3)), ((64, 64, 16), (64, 64, 16)), ((64, 64, 8), (128, 64, 4)),
               ((128, 64, 32), (128, 0, 8)), ((96, 48, 12), (128, 64, 32)), ((128, 128, 0), (128, 64, 0))]
WIDE_CORES = [(128, 64, 32), (128, 64, 4), (64, 64, 16), (128, 128, 8), (64, 32, 8)]


def _random_big_batch(rng, gn):
    cps, rvs, sizes, R = _random_big_csc(rng)
    return gn.GNGraphBatch.from_csc(cps, rvs, sizes), R


def _random_big_csc(rng):
    """Batches around the row thresholds of the six-term kernels (4096 edge / node rows) with ragged last tiles: one graph (maybe replicas)
    or a few, edge counts from below the threshold to a few tiles above it, now and then a hub that takes a third of the edges."""
    shared = rng.random() < 0.4
    G = 1 if shared else int(rng.integers(2, 5))
    cps, rvs, sizes = [], [], []
    for _ in range(G):
        n = int(rng.choice([37, 300, 1500, 4095, 4097, 4500, 6001]))
        e = int(rng.choice([n, 3 * n + 7, 4096, 4099, 9000, 14001])) if n * n > 14001 else int(rng.integers(1, n * n // 2))
        cp, rv = U.er_csc(rng, n, min(e, n * n // 2))
        if rng.random() < 0.25 and n > 300:  # a hub: every third edge re-aimed at node n // 2 (order by destination restored)
            dst = np.repeat(np.arange(n), np.diff(cp))
            pick = rng.random(dst.size) < 0.33
            dst[pick] = n // 2
            key = np.unique(dst * n + rv)
            dst, rv = key // n, key % n
            cp = np.zeros(n + 1, dtype=np.int64)
            np.add.at(cp, dst + 1, 1)
            cp = np.cumsum(cp)
        cps.append(cp); rvs.append(rv.astype(np.int64)); sizes.append(n)
    return cps, rvs, sizes, (int(rng.integers(1, 3)) if shared else 1)


@pytest.mark.parametrize("seed", range(10 + EXTRA // 8))
def test_random_wide_block(gn, seed):
    """The wide forms (projected edge update, six-term kernels from 4096 rows on, the general kernels below) on random big batches: default
    forms, the fp32-instruction forms and the generic kernels against the float64 oracle."""
    rng = np.random.default_rng(9900 + seed)
    g, R = _random_big_batch(rng, gn)
    din, dout = WIDE_BLOCKS[int(rng.integers(0, len(WIDE_BLOCKS)))]
    p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.integers(0, 5, 3)))
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, din)
    csc = (*g.csc(), g.node_off, g.edge_off)
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    blk = U.block_from_params(gn, p)
    for flags in (0, gn._lib.FLAG_FP32_MFMA | gn._lib.FLAG_PROJ_FP32 | gn._lib.FLAG_EDGE_NARROW_FP32, 1):
        y = blk(U.to_nt(gn, g, ef, nf, gf), flags=flags)
        for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
            if r is None or (hasattr(r, "shape") and 0 in r.shape):
                continue
            U.assert_close(U.from_jl(got), r, s, f"seed {seed} dims {din}=>{dout} N={g.n_nodes} E={g.n_edges} G={g.n_graphs} R={R} flags={flags:#x} {name}")


@pytest.mark.parametrize("seed", range(8 + EXTRA // 8))
def test_random_wide_core(gn, seed):
    """GNCore at wide widths on the same batches: the one-launch edge kernel, the table forms (GNX_FLAG_LN_ON_LOAD), the fp32-instruction
    forms and the generic kernels."""
    rng = np.random.default_rng(9950 + seed)
    g, R = _random_big_batch(rng, gn)
    dims = WIDE_CORES[int(rng.integers(0, len(WIDE_CORES)))]
    p = O.make_core_params(rng, dims, eps_mode=int(rng.integers(0, 2)))
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, dims)
    csc = (*g.csc(), g.node_off, g.edge_off)
    ref, scale = O.core_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    core = U.core_from_params(gn, p)
    for flags in (0, gn._lib.FLAG_LN_ON_LOAD, gn._lib.FLAG_FP32_MFMA | gn._lib.FLAG_PROJ_FP32 | gn._lib.FLAG_EDGE_NARROW_FP32, 1):
        y = core(U.to_nt(gn, g, ef, nf, gf), flags=flags)
        for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
            U.assert_close(U.from_jl(got), r, s, f"seed {seed} dims {dims} N={g.n_nodes} E={g.n_edges} G={g.n_graphs} R={R} flags={flags:#x} {name}")


@pytest.mark.parametrize("seed", range(8 + EXTRA // 8))
def test_random_model(gn, seed):
    """encoder block -> 0-2 GNCores -> decoder block at random widths on random batches (README ex.3's shape, sort.jl:68-75), three ways: the
    eager chain of layer calls against the oracle (layer by layer at 1e-5 of the bound, end to end normwise), and the library's one-hipGraph model
    (gnx_model_*; captured and eager) bit-identical to the eager chain."""
    import torch
    rng = np.random.default_rng(9600 + seed)
    big = rng.random() < 0.3
    g, R = _random_big_batch(rng, gn) if big else _random_batch(rng, gn)
    din = tuple(int(rng.choice([0, 2, 5, 10])) for _ in range(3))
    if din[0] + din[1] == 0:
        din = (4, 3, din[2])
    core = tuple(int(v) for v in (rng.choice([(128, 64, 32), (64, 64, 16), (128, 64, 8)]) if big else rng.choice([(10, 5, 3), (8, 8, 8), (16, 12, 4), (33, 20, 5), (3, 4, 5)])))
    dout = tuple(int(rng.choice([1, 3, 4, 7])) for _ in range(3))
    n_cores = int(rng.integers(0, 3))
    specs = [("block", O.make_block_params(rng, din, core, act=tuple(int(a) for a in rng.integers(0, 3, 3))))]
    specs += [("core", O.make_core_params(rng, core, eps_mode=int(rng.integers(0, 2)))) for _ in range(n_cores)]
    specs += [("block", O.make_block_params(rng, core, dout, act=(0, 0, 0)))]
    layers = [(k, p, U.block_from_params(gn, p) if k == "block" else U.core_from_params(gn, p)) for k, p in specs]
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, din)
    csc = (*g.csc(), g.node_off, g.edge_off)
    U.check_chain(gn, g, csc, layers, (ef, nf, gf), f"seed {seed} {din}=>{core} x{n_cores} =>{dout} N={g.n_nodes} E={g.n_edges} G={g.n_graphs} R={R}",
                  normwise=1e-5 * (1 + n_cores))
    x = U.to_nt(gn, g, ef, nf, gf)
    y = x
    for _, _, layer in layers:
        y = layer(y)
    eager = [None if a is None else a.clone() for a in (y.ef, y.nf, y.gf)]
    for flags in (0, gn._lib.FLAG_NO_GRAPH):
        m = gn.Model([l for _, _, l in layers], x, flags=flags)
        for _ in range(2):
            ym = m(x)
        for name, a, b in zip(("ef", "nf", "gf"), eager, (ym.ef, ym.nf, ym.gf)):
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), f"seed {seed} model flags {flags:#x}: {name} differs from the eager chain"


@pytest.mark.parametrize("seed", range(10 + EXTRA // 4))
def test_random_batch_views_padding_and_collapse(gn, seed):
    """The data-format side of the path on random vectors of adjacency matrices (graphs of one node, without edges, one-way edges, missing self
    loops) and on the shared form: batch -> unbatch is the identity (batch.jl:53-64, unbatch.jl:6-39); efview / nfview / gfview, flatunpadded*,
    padded() and unpadded() against the oracle's padded arrays (pad.jl, unpad.jl, views.jl); collapsef / unpaddedcollapsedef against the literal
    edge_collapser product (gngraphbatch.jl:56-111)."""
    import torch
    rng = np.random.default_rng(9300 + seed)
    shared = rng.random() < 0.3
    de, dn, dg = (int(rng.choice([0, 1, 3, 8])) for _ in range(3))
    if de + dn + dg == 0:
        de = 2
    if shared:
        n, B = int(rng.choice([1, 2, 5, 9])), int(rng.integers(1, 4))
        adj = (rng.random((n, n)) < rng.choice([0.0, 0.3, 1.0])).astype(np.int64)
        ne = int(adj.sum())
        ef = rng.standard_normal((de, ne, B)).astype(np.float32) if de else None
        nf = rng.standard_normal((dn, n, B)).astype(np.float32) if dn else None
        gf = rng.standard_normal((dg, B)).astype(np.float32) if dg else None
        graphs, adjs = adj, [adj]
    else:
        adjs = [(rng.random((n, n)) < rng.choice([0.0, 0.2, 0.6, 1.0])).astype(np.int64) for n in rng.choice([1, 2, 3, 6, 11], int(rng.integers(1, 6)))]
        ef = [rng.standard_normal((de, int(a.sum()))).astype(np.float32) for a in adjs] if de else None
        nf = [rng.standard_normal((dn, a.shape[0])).astype(np.float32) for a in adjs] if dn else None
        gf = [rng.standard_normal(dg).astype(np.float32) for _ in adjs] if dg else None
        graphs, B = adjs, len(adjs)
    x = gn.batch(dict(graphs=graphs, ef=ef, nf=nf, gf=gf))
    xd = O.batch_dense(graphs, ef, nf, gf)
    gd = xd["graphs"]
    what = f"seed {seed} shared={shared} dims {(de, dn, dg)} sizes {[a.shape[0] for a in adjs]} edges {[int(a.sum()) for a in adjs]} B={B}"
    # batch -> unbatch: the identity
    u = gn.unbatch(x)
    if not shared and B == 1:  # unbatch.jl:15-17: a GNGraphBatch of ONE graph unbatches through the shared-adjacency branch, whatever batch() was given
        for name, got, want in (("ef", u.ef, ef), ("nf", u.nf, nf)):
            assert (got is None) == (want is None), what
            if want is not None:
                assert np.array_equal(got.cpu().numpy(), want[0][:, :, None]), f"{what}: unbatch {name} of a one-graph vector"
        assert (u.gf is None) == (gf is None) and (gf is None or np.array_equal(u.gf.cpu().numpy(), gf[0][:, None])), f"{what}: unbatch gf of a one-graph vector"
    elif shared:
        for name, got, want in (("ef", u.ef, ef), ("nf", u.nf, nf), ("gf", u.gf, gf)):
            assert (got is None) == (want is None), what
            if want is not None:
                assert np.array_equal(got.cpu().numpy(), want), f"{what}: unbatch {name}"
    else:
        for name, got, want in (("ef", u.ef, ef), ("nf", u.nf, nf), ("gf", u.gf, gf)):
            assert (got is None) == (want is None), what
            if want is not None:
                for i in range(B):
                    assert np.array_equal(got[i].cpu().numpy(), want[i]), f"{what}: unbatch {name}[{i}]"
    # views and the flat forms against the padded arrays of the oracle
    PN = gd.node_block_size
    em = gd.flat_edge_unpadder.reshape(PN * PN, -1, order="F")
    nm = gd.flat_node_unpadder.reshape(PN, -1, order="F")
    for b in range(B):
        eb, nb = (em[:, 0], nm[:, 0]) if shared else (em[:, b], nm[:, b])
        if de:
            assert np.array_equal(gn.efview(x, slice(None), slice(None), b).cpu().numpy(), xd["ef"][:, eb, b].astype(np.float32)), f"{what}: efview {b}"
        if dn:
            assert np.array_equal(gn.nfview(x, slice(None), slice(None), b).cpu().numpy(), xd["nf"][:, nb, b].astype(np.float32)), f"{what}: nfview {b}"
        if dg:
            assert np.array_equal(gn.gfview(x, slice(None), b).cpu().numpy(), xd["gf"][:, 0, b].astype(np.float32)), f"{what}: gfview {b}"
    if not shared:
        if de:
            assert np.array_equal(gn.flatunpaddedef(x).cpu().numpy(), O.flat_from_dense(xd, "ef").astype(np.float32)), f"{what}: flatunpaddedef"
        if dn:
            assert np.array_equal(gn.flatunpaddednf(x).cpu().numpy(), O.flat_from_dense(xd, "nf").astype(np.float32)), f"{what}: flatunpaddednf"
    # padded() == the oracle's padded arrays (zeros in the pads), unpadded(padded(x)) == x
    pad = gn.padded(x)
    for name, got, want in (("ef", pad.ef, xd["ef"]), ("nf", pad.nf, xd["nf"]), ("gf", pad.gf, xd["gf"])):
        assert (got is None) == (want is None), f"{what}: padded {name}"
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want.astype(np.float32)), f"{what}: padded {name}"
    back = gn.unpadded(x.graphs, pad.ef, pad.nf, pad.gf)
    for name in ("ef", "nf", "gf"):
        a, b_ = getattr(back, name), getattr(x, name)
        assert (a is None) == (b_ is None) and (a is None or torch.equal(a, b_)), f"{what}: unpadded(padded) {name}"
    # edge collapsing
    if de:
        np.testing.assert_allclose(gn.collapsef(x).cpu().numpy(), O.collapsef_dense(xd), rtol=1e-6, atol=1e-6, err_msg=f"{what}: collapsef")
        for b, (got, want) in enumerate(zip(gn.unpaddedcollapsedef(x), O.unpaddedcollapsedef_dense(xd))):
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-6, err_msg=f"{what}: unpaddedcollapsedef {b}")


@pytest.mark.parametrize("seed", range(10 + EXTRA // 4))
def test_random_fn_inputs_and_readout_loss(gn, seed):
    """The exported building blocks on random batches (graphs of one node, without edges): getedgefninput / getnodefninput / getgraphfninput
    against the oracle's literal one-hot products (edgefninput.jl, nodefninput.jl, graphfninput.jl) for every `nothing` combination they
    admit, and logitcrossentropy (+ its gradient) against torch on random (d, cols) arrays."""
    import torch
    rng = np.random.default_rng(9200 + seed)
    adjs = [(rng.random((n, n)) < rng.choice([0.0, 0.2, 0.6, 1.0])).astype(np.int64) for n in rng.choice([1, 2, 3, 6, 11, 30], int(rng.integers(1, 6)))]
    de, dn, dg = (int(rng.choice([0, 1, 3, 8, 33])) for _ in range(3))
    if de + dn + dg == 0:
        dn = 4
    ef = [rng.standard_normal((de, int(a.sum()))).astype(np.float32) for a in adjs] if de else None
    nf = [rng.standard_normal((dn, a.shape[0])).astype(np.float32) for a in adjs] if dn else None
    gf = [rng.standard_normal(dg).astype(np.float32) for _ in adjs] if dg else None
    x = gn.batch(dict(graphs=adjs, ef=ef, nf=nf, gf=gf))
    xd = O.batch_dense(adjs, ef, nf, gf)
    g = xd["graphs"]
    B, PN = len(adjs), g.node_block_size
    em = g.flat_edge_unpadder.reshape(PN * PN, B, order="F")
    nm = g.flat_node_unpadder.reshape(PN, B, order="F")
    what = f"seed {seed} dims {(de, dn, dg)} sizes {[a.shape[0] for a in adjs]} edges {[int(a.sum()) for a in adjs]}"
    got = gn.getedgefninput(x.graphs, x.ef, x.nf, x.gf).cpu().numpy()[:, :, 0]
    ref = O._vcat(([xd["ef"]] if de else []) + ([O.batched_mul(xd["nf"], g.srcnode2edge), O.batched_mul(xd["nf"], g.dstnode2edge)] if dn else []) +
                  ([O.batched_mul(xd["gf"], g.graph2edge)] if dg else []))
    np.testing.assert_allclose(got, np.concatenate([ref[:, em[:, b], b] for b in range(B)], axis=1), rtol=1e-6, atol=1e-6, err_msg=f"{what}: getedgefninput")
    if de:  # the node function sees the (updated) edge features; nf / gf optional
        got = gn.getnodefninput(x.graphs, x.ef, x.nf, x.gf).cpu().numpy()[:, :, 0]
        ref = O._vcat([O.batched_mul(xd["ef"], g.edge2node)] + ([xd["nf"]] if dn else []) + ([O.batched_mul(xd["gf"], g.graph2node)] if dg else []))
        np.testing.assert_allclose(got, np.concatenate([ref[:, nm[:, b], b] for b in range(B)], axis=1), rtol=1e-5, atol=1e-5, err_msg=f"{what}: getnodefninput")
    if de and dn:
        got = gn.getgraphfninput(x.graphs, x.ef, x.nf, x.gf).cpu().numpy()[:, :, 0]
        ref = O._vcat([O.batched_mul(xd["ef"], g.edge2graph), O.batched_mul(xd["nf"], g.node2graph)] + ([xd["gf"]] if dg else []))
        np.testing.assert_allclose(got, ref[:, 0, :], rtol=1e-5, atol=1e-5, err_msg=f"{what}: getgraphfninput")
    # readout loss
    d, cols = int(rng.choice([1, 2, 5, 33, 130])), int(rng.choice([1, 3, 64, 257, 5000]))
    yh = torch.from_numpy((rng.standard_normal((d, cols)) * rng.choice([0.1, 1.0, 30.0])).astype(np.float32))
    y = torch.softmax(torch.from_numpy(rng.standard_normal((d, cols)).astype(np.float32)), dim=0)
    a = yh.clone().cuda().requires_grad_(True)
    loss = gn.logitcrossentropy(a, y.cuda())
    loss.backward()
    r = yh.double().clone().requires_grad_(True)
    lr = -(y.double() * torch.log_softmax(r, dim=0)).sum(dim=0).mean()
    lr.backward()
    lv, lrv = float(loss.detach()), float(lr.detach())
    assert abs(lv - lrv) <= 1e-5 * max(1.0, abs(lrv)), f"{what}: logitcrossentropy ({d}, {cols}): {lv} vs {lrv}"
    np.testing.assert_allclose(a.grad.cpu().numpy(), r.grad.numpy(), rtol=1e-4, atol=1e-6 / cols + 1e-9, err_msg=f"logitcrossentropy gradient ({d}, {cols})")


@pytest.mark.parametrize("seed", range(8 + EXTRA // 4))
def test_random_batch_constructors_agree(gn, seed):
    """GNGraphBatch (gngraphbatch.jl:33-54, :136-211) through every constructor — a list of dense matrices (the host builder, or the device
    builder for big ones), ONE packed buffer of several element types in pageable / device memory, per-graph CSC, concatenated CSC — on random
    batches (one-node graphs, graphs without edges, dense ones, many small ones): the same CSC, offsets, block sizes and unpadder masks, all
    equal to the oracle's literal construction."""
    import torch
    rng = np.random.default_rng(9100 + seed)
    G = int(rng.choice([1, 2, 5, 40, 300]))
    sizes = [int(v) for v in rng.choice([1, 2, 3, 7, 20, 64, 130] if G <= 5 else [1, 2, 3, 7, 20], G)]
    adjs = [(rng.random((n, n)) < rng.choice([0.0, 0.05, 0.3, 1.0])).astype(np.int64) for n in sizes]
    colptr, rowval, node_off, edge_off = O.csc_from_adj(adjs)
    what = f"seed {seed} G={G} N={sum(sizes)} E={len(rowval)}"
    cps, rvs = [], []
    for g_, n in enumerate(sizes):
        n0, e0, e1 = node_off[g_], edge_off[g_], edge_off[g_ + 1]
        cps.append(np.asarray(colptr[n0:n0 + n + 1]) - e0)
        rvs.append(np.asarray(rowval[e0:e1]) - n0)
    cat = np.concatenate([a.reshape(-1) for a in adjs])
    dt = [np.uint8, np.bool_, np.int32, np.int64, np.float32, np.float64][int(rng.integers(0, 6))]
    built = {
        "dense list": gn.GNGraphBatch(adjs),
        "dense list float32": gn.GNGraphBatch([a.astype(np.float32) for a in adjs]),
        f"packed {np.dtype(dt).name} pageable": gn.GNGraphBatch.from_dense_packed(cat.astype(dt), sizes),
        "packed uint8 device": gn.GNGraphBatch.from_dense_packed(torch.from_numpy(cat.astype(np.uint8)).cuda(), sizes),
        "csc": gn.GNGraphBatch.from_csc(cps, rvs, sizes),
        "csc packed int32": gn.GNGraphBatch.from_csc_packed(np.concatenate(cps).astype(np.int32), np.concatenate(rvs).astype(np.int32) if len(rowval) else np.zeros(0, np.int32), sizes),
    }
    dense = O.batch_dense(adjs, None, [np.zeros((1, n), np.float32) for n in sizes], None)["graphs"]
    for name, g in built.items():
        cp, rv = g.csc()
        assert (g.n_graphs, g.n_nodes, g.n_edges) == (G, sum(sizes), len(rowval)), f"{what}: {name}: counts"
        assert np.array_equal(cp, colptr) and np.array_equal(rv, rowval), f"{what}: {name}: CSC"
        assert np.array_equal(g.node_off, node_off) and np.array_equal(g.edge_off, edge_off), f"{what}: {name}: offsets"
        assert g.node_block_size == dense.node_block_size and g.edge_block_size == dense.edge_block_size, f"{what}: {name}: block sizes"
        assert np.array_equal(g.flat_node_unpadder, dense.flat_node_unpadder) and np.array_equal(g.flat_edge_unpadder, dense.flat_edge_unpadder), f"{what}: {name}: unpadders"


@pytest.mark.parametrize("seed", range(8 + EXTRA // 6))
def test_random_plan_steps_and_chained_calls_equal_single_forwards(gn, seed):
    """A loop over resident batches in the library's three forms — gnx_block_forward per step, gnx_block_forward_steps (one call), and the
    chained calls + flush — on random width sets / batches (replicas, graphs without edges): bit-identical outputs, also when the steps share
    ONE workspace, and against the oracle once."""
    import torch
    rng = np.random.default_rng(9050 + seed)
    big = rng.random() < 0.25
    g, R = _random_big_batch(rng, gn) if big else _random_batch(rng, gn)
    if big:
        din, dout = WIDE_BLOCKS[int(rng.integers(0, len(WIDE_BLOCKS)))]
    else:
        din, dout = _dims(rng)
        if rng.random() < 0.5:
            din, dout = ((10, 5, 0), (3, 4, 5)) if rng.random() < 0.5 else ((10, 5, 3), (10, 5, 3))  # (the ahead-of-time sets: the chained form exists for them)
    p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.integers(0, 5, 3)))
    blk = U.block_from_params(gn, p)
    plan = gn.BlockPlan(blk, g, R=R)
    dev = g.device
    mk = lambda T, d: torch.from_numpy(rng.standard_normal((R, T, d)).astype(np.float32)).to(dev) if d > 0 else None
    n = int(rng.integers(1, 6))
    sets = [dict(ef=mk(g.n_edges, din[0]), nf=mk(g.n_nodes, din[1]), gf=mk(g.n_graphs, din[2]), out=plan.outputs(), ws=plan.new_workspace()) for _ in range(n)]
    what = f"seed {seed} dims {din}=>{dout} N={g.n_nodes} E={g.n_edges} G={g.n_graphs} R={R} steps={n}"
    ref = []
    for b in sets:
        out = plan.outputs()
        plan(b["ef"], b["nf"], b["gf"], *out)
        ref.append(out)
    torch.cuda.synchronize()

    def same(tag):
        torch.cuda.synchronize()
        for i, b in enumerate(sets):
            for name, a, r in zip(("ef", "nf", "gf"), b["out"], ref[i]):
                assert (a is None) == (r is None) and (a is None or torch.equal(a, r)), f"{what}: {tag}: step {i} {name}"
                if a is not None:
                    a.fill_(float("nan"))
    plan.steps(sets)
    same("steps")
    plan.steps([dict(b, ws=sets[0]["ws"]) for b in sets])
    same("steps on one workspace")
    pend = None
    for b in sets:
        pend = plan.chained(b["ef"], b["nf"], b["gf"], *b["out"], ws=b["ws"], prev=pend)
    plan.flush(pend)
    same("chained calls + flush")
    # ... and the values themselves, once
    b = sets[0]
    host = lambda t: None if t is None else t.cpu().numpy()
    r_o, s_o = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), host(b["ef"]), host(b["nf"]), host(b["gf"]), return_scale=True)
    for name, got, r, sc in zip(("ef", "nf", "gf"), ref[0], r_o, s_o):
        if r is None or 0 in r.shape:
            continue
        U.assert_close(got.cpu().numpy(), r, sc, f"{what}: {name}")


@pytest.mark.parametrize("seed", range(6 + EXTRA // 8))
def test_random_by_graph_sharding_over_virtual_ranks(gn, seed):
    """The N > 1 data path except the wire on random batches and world sizes (shards that differ by one graph, graphs without edges, one graph
    per rank): gnx_dist_partition's shards run one after the other on this GPU, their padded send buffers concatenated as the all-gather would,
    the gather plan restores original graph order — gf' of the whole batch and every shard's ef' / nf' equal the single-process oracle; the C
    boundary's gnx_dist_permute_rows gives the same table."""
    import torch
    from graphnets_jl_amd.dist import GfGather, gather_plan, partition_graphs
    rng = np.random.default_rng(9400 + seed)
    G = int(rng.choice([1, 2, 3, 9, 40, 130]))
    world = int(rng.integers(1, min(G, 8) + 1))
    cps, rvs, sizes = [], [], []
    for _ in range(G):
        n = int(rng.choice([1, 2, 5, 17, 60]))
        kind = rng.integers(0, 3)
        if kind == 0:
            cp, rv = np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64)
        else:
            cp, rv = U.er_csc(rng, n, int(rng.integers(1, max(2, n * n // 2))))
        cps.append(cp); rvs.append(rv); sizes.append(n)
    e_all = np.array([len(r) for r in rvs], dtype=np.int64)
    shards = partition_graphs(e_all, world)
    assert sorted(int(i) for s_ in shards for i in s_) == list(range(G)) and max(len(s_) for s_ in shards) - min(len(s_) for s_ in shards) <= 1
    node_off, edge_off = np.concatenate([[0], np.cumsum(sizes)]), np.concatenate([[0], np.cumsum(e_all)])
    din, dout = (((10, 5, 0), (3, 4, 5)), ((10, 5, 3), (10, 5, 3)), ((4, 3, 2), (3, 4, 2)), ((33, 20, 5), (7, 12, 3)))[int(rng.integers(0, 4))]
    p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.integers(0, 5, 3)))
    ef, nf, gf = U.packed_inputs(rng, 1, int(edge_off[-1]), int(node_off[-1]), G, din)
    blk = U.block_from_params(gn, p)
    og = dout[2]
    what = f"seed {seed} G={G} world={world} dims {din}=>{dout} N={node_off[-1]} E={edge_off[-1]}"
    gathers = [GfGather(shards, r, world, og, "cuda", overlap=False) for r in range(world)]
    outs = []
    for r in range(world):
        mine = shards[r]
        g = gn.GNGraphBatch.from_csc([cps[i] for i in mine], [rvs[i] for i in mine], [sizes[i] for i in mine])
        take = lambda a, off: None if a is None else torch.from_numpy(np.concatenate([a[0, off[i]:off[i + 1]] for i in mine])[None]).to(g.device)
        gfl = None if gf is None else torch.from_numpy(gf[0, mine][None]).to(g.device)
        plan = gn.BlockPlan(blk, g)
        eo, no, _ = plan.outputs()
        plan(take(ef, edge_off), take(nf, node_off), gfl, eo, no, gathers[r].send[:, :len(mine)])
        outs.append((eo, no))
    torch.cuda.synchronize()
    wire = torch.cat([gt.send.view(-1, og) for gt in gathers], dim=0).contiguous()
    gathers[0].recv.copy_(wire)
    gf_all = gathers[0].result().cpu().numpy()
    src, mc = gather_plan(shards)
    assert mc == max(len(s_) for s_ in shards)
    out_c = torch.empty((G, og), dtype=torch.float32, device="cuda")
    gn._lib.check(gn._lib.load().gnx_dist_permute_rows(wire.data_ptr(), torch.from_numpy(src).cuda().data_ptr(), G, og, out_c.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(out_c.cpu().numpy(), gf_all), f"{what}: gnx_dist_permute_rows"
    cp = np.concatenate([[0]] + [c[1:] + edge_off[i] for i, c in enumerate(cps)])
    rv = np.concatenate([r_ + node_off[i] for i, r_ in enumerate(rvs)]) if edge_off[-1] else np.zeros(0, dtype=np.int64)
    ref, scale = O.block_forward_sparse(p, (cp, rv, node_off, edge_off), ef, nf, gf, return_scale=True)
    U.assert_close(gf_all[None], ref[2], scale[2], f"{what}: gathered gf'")
    for r in range(world):
        eidx = np.concatenate([np.arange(edge_off[i], edge_off[i + 1]) for i in shards[r]]).astype(np.int64)
        nidx = np.concatenate([np.arange(node_off[i], node_off[i + 1]) for i in shards[r]]).astype(np.int64)
        if len(eidx):
            U.assert_close(outs[r][0].cpu().numpy(), ref[0][:, eidx], scale[0][:, eidx], f"{what}: ef' of shard {r}")
        U.assert_close(outs[r][1].cpu().numpy(), ref[1][:, nidx], scale[1][:, nidx], f"{what}: nf' of shard {r}")


@pytest.mark.parametrize("seed", range(6 + EXTRA // 8))
def test_random_prepared_parameters_are_bit_identical_and_follow_in_place_updates(gn, seed):
    """`prepare()` (gnx_block_prepare / gnx_core_prepare: the weight planes of the matrix-core kernels made once — `model |> device`) on random
    blocks / cores and batches on both sides of the 4096-row thresholds: the forward is bit-identical to the unprepared layer's, and after an
    in-place update of a random subset of the parameters the prepared layer equals a FRESH layer built from the updated values."""
    import torch
    rng = np.random.default_rng(9000 + seed)
    g, R = _random_big_batch(rng, gn)
    is_core = rng.random() < 0.5
    if is_core:
        dims = WIDE_CORES[int(rng.integers(0, len(WIDE_CORES)))]
        p = O.make_core_params(rng, dims, eps_mode=int(rng.integers(0, 2)))
        mk = lambda: U.core_from_params(gn, p)
        din = dims
    else:
        din, dout = WIDE_BLOCKS[int(rng.integers(0, len(WIDE_BLOCKS)))]
        p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.integers(0, 5, 3)))
        mk = lambda: U.block_from_params(gn, p)
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, din)
    x = U.to_nt(gn, g, ef, nf, gf)
    what = f"seed {seed} {'core' if is_core else 'block'} {din} N={g.n_nodes} E={g.n_edges} G={g.n_graphs} R={R}"

    def same(a, b, tag):
        for name in ("ef", "nf", "gf"):
            u, v = getattr(a, name), getattr(b, name)
            assert (u is None) == (v is None) and (u is None or torch.equal(u, v)), f"{what}: {tag}: {name}"
    plain, prepared = mk(), mk().prepare()
    with torch.no_grad():
        same(prepared(x), plain(x), "prepared vs unprepared")
        params = prepared._param_list() if is_core else [t for l in (prepared.edgefn, prepared.nodefn, prepared.graphfn) for t in (l.weight, l.bias)]
        fresh = mk()
        fparams = fresh._param_list() if is_core else [t for l in (fresh.edgefn, fresh.nodefn, fresh.graphfn) for t in (l.weight, l.bias)]
        touched = [i for i in range(len(params)) if rng.random() < 0.4] or [0]
        for i in touched:
            if params[i] is None or params[i].numel() == 0:
                continue
            delta = torch.from_numpy((rng.standard_normal(tuple(params[i].shape)) * 0.05).astype(np.float32)).to(params[i].device)
            params[i].add_(delta)    # in place: the version counter moves, the planes are refreshed before the next forward
            fparams[i].add_(delta)
        same(prepared(x), fresh(x), f"after in-place updates of parameters {touched}")


@pytest.mark.parametrize("seed", range(8 + EXTRA // 8))
def test_random_magnitudes(gn, seed):
    """Inputs and weights at random magnitudes, 1e-12 .. 1e10 per tensor (mixed within one call): block and GNCore, narrow and wide forms, against
    the float64 oracle at 1e-5 of its magnitude bound — the fused FMAs, the LayerNorm statistics (sigma + eps with sigma far below and far above
    eps) and the six-term split (exponent range of the bf16 parts) away from unit scale."""
    rng = np.random.default_rng(9500 + 31 * seed)
    big = rng.random() < 0.4
    g, R = _random_big_batch(rng, gn) if big else _random_batch(rng, gn)
    core = rng.random() < 0.5
    mag = lambda: float(10.0 ** rng.uniform(-12, 10))
    if core:
        dims = WIDE_CORES[int(rng.integers(0, len(WIDE_CORES)))] if big else tuple(int(v) for v in rng.choice([(10, 5, 3), (8, 8, 8), (33, 20, 5)]))
        p = O.make_core_params(rng, dims, eps_mode=int(rng.integers(0, 2)))
        din = dims
    else:
        din, dout = WIDE_BLOCKS[int(rng.integers(0, len(WIDE_BLOCKS)))] if big else _dims(rng)
        p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.choice([0, 1], 3)))  # (identity / relu: scale-equivariant, no saturation at 1e10)
        for k in ("We", "Wn", "Wg"):
            p[k] = (p[k] * mag() ** 0.5).astype(np.float32)
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, din)
    ef, nf, gf = (None if a is None else (a * np.float32(mag())).astype(np.float32) for a in (ef, nf, gf))
    csc = (*g.csc(), g.node_off, g.edge_off)
    layer = U.core_from_params(gn, p) if core else U.block_from_params(gn, p)
    ref, scale = (O.core_forward_sparse if core else O.block_forward_sparse)(p, csc, ef, nf, gf, return_scale=True)
    y = layer(U.to_nt(gn, g, ef, nf, gf))
    for name, got, r, s_ in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        if r is None or 0 in r.shape:
            continue
        assert np.isfinite(r).all()
        U.assert_close(U.from_jl(got), r, s_, f"seed {seed} {'core' if core else 'block'} dims {din} N={g.n_nodes} E={g.n_edges} R={R} {name}")


def _random_chain(rng, widths, first_in, ln_min=1):
    """0-3 Dense layers with LayerNorm layer values sprinkled in, each over at least `ln_min` columns (never in front of a zero-width input; the
    backward sweeps ask for 2: the derivative of sigma at a one-column row is 0 / 0 in the float64 reference too)."""
    n = int(rng.integers(0, 4))
    out, k = [], first_in
    for i in range(n):
        if rng.random() < 0.3 and k >= ln_min:
            out.append("ln")
        k = int(rng.choice(widths))
        out.append(k)
    if out and rng.random() < 0.2 and k >= ln_min:
        out.append("ln")
    return out


@pytest.mark.parametrize("seed", range(12 + EXTRA // 4))
def test_random_chain_block(gn, seed):
    """GNBlock whose update functions are random Chains of Dense / LayerNorm layers (gnblock.jl:1-6) on random small and big batches."""
    from tests.test_gpu_chain import _block
    rng = np.random.default_rng(9700 + seed)
    big = rng.random() < 0.35
    g, _ = _random_big_batch(rng, gn) if big else _random_batch(rng, gn)
    widths = [8, 32, 64, 128] if big else [1, 3, 7, 12, 16, 33]
    while True:
        in_dims = tuple(int(rng.choice([0] + widths)) for _ in range(3))
        if sum(in_dims) > 0:
            break
    ew = _random_chain(rng, widths, in_dims[0] + 2 * in_dims[1] + in_dims[2])
    oe = next((w for w in reversed(ew) if w != "ln"), 0)
    nw = _random_chain(rng, widths, oe + in_dims[1] + in_dims[2]) if oe + in_dims[1] + in_dims[2] > 0 else []  # (nothing to feed it: vcat of nothings)
    on = next((w for w in reversed(nw) if w != "ln"), 0)
    gw = _random_chain(rng, widths, oe + on + in_dims[2]) if oe + on + in_dims[2] > 0 else []
    if not (ew or nw or gw):
        ew = [int(rng.choice(widths))]
    p = O.make_chain_block_params(rng, in_dims, ew, nw, gw, acts=tuple(int(a) for a in rng.integers(0, 5, 3)))
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, in_dims)
    csc = (*g.csc(), g.node_off, g.edge_off)
    ref, scale = O.chain_block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    y = _block(gn, p)(U.to_nt(gn, g, ef, nf, gf))
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        if r is None:
            assert got is None, f"seed {seed}: {name} should be nothing"
            continue
        U.assert_close(U.from_jl(got), r, s, f"seed {seed} in {in_dims} edge {ew} node {nw} graph {gw} N={g.n_nodes} E={g.n_edges} {name}")


@pytest.mark.parametrize("seed", range(12 + EXTRA // 4))
def test_random_block_backward(gn, seed):
    """gnx_block_backward on random width sets / batches (smooth activations: no relu kink) against torch float64 autograd."""
    rng = np.random.default_rng(9800 + seed)
    g, _ = _random_batch(rng, gn)
    din, dout = _dims(rng)
    _check_block_backward(gn, rng, g, din, dout, seed)


@pytest.mark.parametrize("seed", range(8 + EXTRA // 6))
def test_random_chain_block_backward(gn, seed):
    """gnx_chain_block_backward for random Chains of Dense / LayerNorm layers (smooth activations) against torch float64 autograd."""
    from tests.test_gpu_chain import chain_block_backward_case
    rng = np.random.default_rng(9750 + seed)
    big = rng.random() < 0.3
    g, _ = _random_big_batch(rng, gn) if big else _random_batch(rng, gn)
    widths = [8, 24, 48, 64] if big else [1, 3, 7, 12, 16, 33]
    while True:
        in_dims = tuple(int(rng.choice([0] + widths)) for _ in range(3))
        if sum(in_dims) > 0:
            break
    ke = in_dims[0] + 2 * in_dims[1] + in_dims[2]
    ew = _random_chain(rng, widths, ke, 2) or [int(rng.choice(widths))]   # (the chain pullback needs an edge function with outputs)
    oe = next((w for w in reversed(ew) if w != "ln"), 0)
    nw = _random_chain(rng, widths, oe + in_dims[1] + in_dims[2], 2)
    on = next((w for w in reversed(nw) if w != "ln"), 0)
    gw = _random_chain(rng, widths, oe + on + in_dims[2], 2)
    acts = tuple(int(a) for a in rng.choice([0, 2, 3, 4], 3))
    assert chain_block_backward_case(gn, g, rng, in_dims, ew, nw, gw, acts, fp32_yardstick=True), f"seed {seed} in {in_dims} edge {ew} node {nw} graph {gw}"


@pytest.mark.parametrize("seed", range(6 + EXTRA // 8))
def test_random_core_backward(gn, seed):
    """gnx_core_backward (LayerNorms, block, FeedForward, residual; smooth hidden activation) at random widths on random small and big batches
    against torch float64 autograd of the independent restatement in tests/test_gpu_backward.py."""
    from tests.test_gpu_backward import _core_backward_case
    rng = np.random.default_rng(9850 + seed)
    big = rng.random() < 0.4
    cps, rvs, sizes, _ = _random_big_csc(rng) if big else _random_csc(rng)
    dims = tuple(int(v) for v in (rng.choice([(128, 64, 32), (64, 64, 16), (64, 32, 8), (40, 36, 33)]) if big else rng.choice([(3, 4, 5), (10, 5, 3), (8, 8, 8), (16, 12, 4), (33, 20, 5), (64, 32, 8)])))
    hidden = str(rng.choice(["tanh", "gelu"]))
    assert _core_backward_case(gn, dims, big, int(rng.integers(0, 2)), rng, hidden_act=hidden, graphs=(sizes, cps, rvs))


@pytest.mark.parametrize("seed", range(4 + EXTRA // 12))
def test_random_core_training_with_dropout(gn, seed):
    """Training-mode GNCore (Dropout inside the FeedForward, masks regenerated from the seed) forward and every gradient against float64 with the
    same masks (tests/test_gpu_dropout.py), at random widths on random batches — batches without edges included."""
    from tests.test_gpu_dropout import _case
    rng = np.random.default_rng(9880 + seed)
    big = rng.random() < 0.3
    cps, rvs, sizes, _ = _random_big_csc(rng) if big else _random_csc(rng)
    dims = tuple(int(v) for v in (rng.choice([(64, 32, 16), (40, 36, 33), (64, 64, 8)]) if big else rng.choice([(3, 4, 5), (10, 5, 3), (8, 8, 8), (16, 12, 4), (33, 20, 5)])))
    _case(gn, dims, big, rng, float(rng.choice([0.1, 0.3, 0.5])), str(rng.choice(["tanh", "gelu"])), graphs=(sizes, cps, rvs))


@pytest.mark.parametrize("din,dout", [((20, 0, 0), (0, 24, 24)),   # node function without inputs (oe = dn = dg = 0): bias only
                                      ((0, 0, 7), (3, 0, 2)),     # only graph features in; no node function
                                      ((0, 5, 0), (0, 0, 4)),     # graph function fed by nothing but zero-width sums
                                      ((4, 0, 0), (0, 0, 3)),
                                      ((0, 0, 2), (0, 4, 0))])    # node function fed by gf alone
def test_block_backward_degenerate_widths(gn, din, dout):
    """Width sets where a function has NO input columns (K = 0: its dX does not exist, its bias gradient does) — found by the
    extended sweep as a zero-size launch; kept as a fixed case."""
    rng = np.random.default_rng(31337)
    g, _ = _random_batch(rng, gn)
    _check_block_backward(gn, rng, g, din, dout, f"degenerate {din}=>{dout}")


def _check_block_backward(gn, rng, g, din, dout, seed):
    import torch
    from tests.test_gpu_backward import _torch_block
    p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.choice([0, 2, 3, 4], 3)))
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, din)
    csc = (*g.csc(), g.node_off, g.edge_off)
    W = {k: torch.tensor(p[k], dtype=torch.float64, requires_grad=True) for k in ("We", "be", "Wn", "bn", "Wg", "bg")}
    t64 = lambda a: None if a is None else torch.tensor(a[0], dtype=torch.float64, requires_grad=True)
    ef_r, nf_r, gf_r = t64(ef), t64(nf), t64(gf)
    outs_r = _torch_block(p, csc, ef_r, nf_r, gf_r, W)
    cot = [torch.from_numpy(rng.standard_normal(tuple(o.shape))) for o in outs_r]
    terms = [(o * c).sum() for o, c in zip(outs_r, cot) if o.numel() > 0]
    if not terms:
        pytest.skip("no output elements")
    sum(terms).backward()
    blk = U.block_from_params(gn, p)
    for layer in (blk.edgefn, blk.nodefn, blk.graphfn):
        layer.weight.requires_grad_(True); layer.bias.requires_grad_(True)
    dev = g.device
    leaf = lambda a: None if a is None else torch.from_numpy(a).to(dev).requires_grad_(True)
    ef_t, nf_t, gf_t = leaf(ef), leaf(nf), leaf(gf)
    jl = lambda t: None if t is None else t.permute(2, 1, 0)
    y = blk(gn.NT(g, jl(ef_t), jl(nf_t), jl(gf_t)))
    loss = 0.0
    for o, c in zip((y.ef, y.nf, y.gf), cot):
        if o is not None and o.numel() > 0:
            loss = loss + (o.permute(2, 1, 0)[0] * c.to(dev).float()).sum()
    loss.backward()

    def close(got, ref, what):
        if ref is None or got is None:
            return
        ref = ref.detach().numpy(); got = got.detach().double().cpu().numpy()
        if ref.size == 0:  # (the gradient of a (0, d) array: a batch without edges)
            assert got.shape == ref.shape, (what, got.shape, ref.shape)
            return
        scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
        assert got.shape == ref.shape and (ref.size == 0 or np.max(np.abs(got - ref)) <= 5e-4 * scale), \
            f"seed {seed} dims {din}=>{dout} {what}: max err {np.max(np.abs(got - ref)) if ref.size else 0:.3e} (scale {scale:.3g})"

    for name, t, r in (("d_ef", ef_t, ef_r), ("d_nf", nf_t, nf_r), ("d_gf", gf_t, gf_r)):
        if t is not None and r.grad is not None and t.grad is not None:
            close(t.grad[0], r.grad, name)
    for name, layer, kw, kb in (("edge", blk.edgefn, "We", "be"), ("node", blk.nodefn, "Wn", "bn"), ("graph", blk.graphfn, "Wg", "bg")):
        if layer.weight.numel() and W[kw].grad is not None and layer.weight.grad is not None:
            close(layer.weight.grad, W[kw].grad, f"dW_{name}")
            close(layer.bias.grad, W[kb].grad, f"db_{name}")


@pytest.mark.parametrize("seed", range(12))
def test_random_dense_adjacency_api(gn, seed):
    """The reference-facing API end to end on random vectors of dense adjacency matrices — batch, block, unbatch, views,
    flatunpadded*, padded — against the LITERAL dense one-hot restatement of the reference (oracle form (i))."""
    rng = np.random.default_rng(9900 + seed)
    G = int(rng.integers(1, 5))
    adjs = [(rng.random((n, n)) < rng.choice([0.0, 0.2, 0.6, 1.0])).astype(np.int64) for n in rng.integers(1, 9, G)]
    din, dout = _dims(rng)
    din = tuple(min(d, 12) for d in din); dout = tuple(min(d, 12) for d in dout)
    if sum(din) == 0 or sum(dout) == 0:
        din, dout = (3, 2, 1), (2, 3, 1)
    p = O.make_block_params(rng, din, dout, act=tuple(int(a) for a in rng.integers(0, 5, 3)))
    mk = lambda d, cols: rng.random((d, cols), dtype=np.float32)
    ef = [mk(din[0], int(a.sum())) for a in adjs] if din[0] else None
    nf = [mk(din[1], a.shape[0]) for a in adjs] if din[1] else None
    gf = [rng.random(din[2], dtype=np.float32) for _ in adjs] if din[2] else None
    x = gn.batch(dict(graphs=adjs, ef=ef, nf=nf, gf=gf))
    yb = U.block_from_params(gn, p)(x)
    y = gn.unbatch(yb)
    dense = O.block_forward_dense(p, O.batch_dense(adjs, ef, nf, gf))
    ref = O.unbatch_dense(dense)
    for i in range(G):
        for k in ("ef", "nf", "gf"):
            got, r = getattr(y, k), ref[k]
            if r is None:
                assert got is None
                continue
            np.testing.assert_allclose(got[i].cpu().numpy(), r[i], rtol=2e-5, atol=2e-5, err_msg=f"seed {seed} graph {i} {k}")
    if dout[1]:
        np.testing.assert_allclose(gn.flatunpaddednf(yb).cpu().numpy(), O.flat_from_dense(dense, "nf"), rtol=2e-5, atol=2e-5)
    if dout[0]:
        np.testing.assert_allclose(gn.flatunpaddedef(yb).cpu().numpy(), O.flat_from_dense(dense, "ef"), rtol=2e-5, atol=2e-5)
    assert np.array_equal(x.graphs.flat_edge_unpadder, dense["graphs"].flat_edge_unpadder)
    assert np.array_equal(x.graphs.flat_node_unpadder, dense["graphs"].flat_node_unpadder)

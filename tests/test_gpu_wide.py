"""GPU parity of the fp32-MFMA wide path (k_rows_gemm) against the float64 oracle: core dims, encoder/decoder shapes
of README ex.3 at core_dims (128,64,32) (BASELINE config 4), unaligned widths, replicas, heterogeneous batches."""
import numpy as np
import pytest

from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


def _check(gn, p, g, csc, ef, nf, gf, flags=0):
    blk = U.block_from_params(gn, p)
    y = blk(U.to_nt(gn, g, ef, nf, gf), flags=flags)
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name)
    return y


DIMS = [
    pytest.param(((128, 64, 32), (128, 64, 32)), id="core"),
    pytest.param(((10, 5, 0), (128, 64, 32)), id="encoder"),
    pytest.param(((128, 64, 32), (3, 4, 5)), id="decoder"),
    pytest.param(((33, 17, 5), (40, 35, 7)), id="unaligned"),
    pytest.param(((0, 48, 0), (64, 0, 16)), id="no-ef-in_no-nf-out"),
    pytest.param(((64, 0, 8), (0, 96, 8)), id="no-ef-out"),
]


@pytest.mark.parametrize("dims", DIMS)
def test_wide_er_graph(gn, dims):
    rng = np.random.default_rng(50)
    N, E = 700, 5000
    colptr, rowval = U.er_csc(rng, N, E)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [N])
    p = O.make_block_params(rng, *dims, act=(1, 0, 2))
    ef, nf, gf = U.packed_inputs(rng, 1, E, N, 1, dims[0])
    _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)


# One case per instantiation family of k_rows_gemm (gnx_wide.hip): loader class (quad rows / row sums / element + packed segments),
# quad or element outputs, epilogue operands (projected form: two gathered addends), transcendental activations, replicas.
VARIANTS = [
    # (din, dout, acts, R, E)                                                       what it exercises
    pytest.param((128, 64, 32), (128, 64, 32), (2, 3, 4), 1, 5000, id="lean-loader_tanh-sigmoid-gelu"),
    pytest.param((64, 32, 8), (64, 32, 8), (1, 1, 1), 3, 5000, id="lean-loader_replicas"),
    pytest.param((10, 5, 3), (128, 64, 32), (1, 2, 0), 2, 5000, id="packed-narrow-segments_quad-out_fused-agg"),
    pytest.param((8, 4, 4), (64, 64, 4), (1, 1, 1), 1, 5000, id="packed-quad-segments"),
    pytest.param((128, 64, 32), (10, 5, 3), (1, 1, 1), 1, 5000, id="projected_element-out_two-operands"),
    pytest.param((126, 64, 32), (64, 32, 8), (2, 1, 0), 1, 5000, id="projected_element-ef_quad-out"),
    pytest.param((64, 18, 6), (36, 34, 7), (1, 3, 1), 2, 5000, id="projected_element-nf_quad-edge-out_element-node-out"),
    pytest.param((40, 12, 0), (48, 40, 0), (4, 1, 0), 1, 5000, id="direct-form_gathered-quad-segments"),
    pytest.param((33, 15, 1), (35, 33, 2), (1, 1, 1), 1, 5000, id="direct-form_element-everything"),
    pytest.param((64, 64, 16), (64, 64, 16), (1, 1, 1), 1, 900, id="sparse_E<2N_no-projection"),
    pytest.param((0, 64, 8), (128, 0, 8), (1, 0, 2), 1, 5000, id="no-ef-in_no-nf-out"),
    pytest.param((192, 64, 0), (128, 128, 0), (1, 1, 0), 1, 5000, id="K-192_node-out-128"),
]


@pytest.mark.parametrize("din,dout,acts,R,E", VARIANTS)
def test_wide_gemm_variants(gn, din, dout, acts, R, E):
    rng = np.random.default_rng(5200 + E + sum(din))
    N = 700
    colptr, rowval = U.er_csc(rng, N, E)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [N])
    p = O.make_block_params(rng, din, dout, act=acts)
    ef, nf, gf = U.packed_inputs(rng, R, E, N, 1, din)
    gn.profile_reset(); gn.profile_enable(True)
    _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    assert any(n.startswith("k_rows_gemm") for n in names), names


def test_wide_hub_node_many_partial_sums(gn):
    """A destination whose in-edges span more than two 64-row chunks of the edge GEMM (in-degree 400): the node update reads its
    sum as first partial + second partial + the rare further ones."""
    rng = np.random.default_rng(53)
    N = 600
    src = np.concatenate([rng.integers(0, N, 3000), rng.integers(0, N, 400)])
    dst = np.concatenate([rng.integers(0, N, 3000), np.full(400, 17)])
    pairs = np.unique(np.stack([dst, src], 1), axis=0)  # sorted by (dst, src): CSC order, no duplicates
    colptr = np.zeros(N + 1, np.int64); np.add.at(colptr, pairs[:, 0] + 1, 1); colptr = np.cumsum(colptr)
    g = gn.GNGraphBatch.from_csc([colptr], [pairs[:, 1]], [N])
    E = len(pairs)
    dims = (64, 32, 8)
    p = O.make_block_params(rng, dims, dims)
    ef, nf, gf = U.packed_inputs(rng, 1, E, N, 1, dims)
    _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)


def test_wide_uses_mfma_kernels(gn):
    rng = np.random.default_rng(51)
    colptr, rowval = U.er_csc(rng, 300, 2000)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [300])
    p = O.make_block_params(rng, (128, 64, 32), (128, 64, 32))
    ef, nf, gf = U.packed_inputs(rng, 1, 2000, 300, 1, (128, 64, 32))
    gn.profile_reset(); gn.profile_enable(True)
    _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    assert {"k_rows_gemm_edge", "k_rows_gemm_node", "k_graph_wide"} <= names, names


def test_wide_replicas_and_hub(gn):
    """shared graph with 3 replicas, one node with 900 in-edges (segment sum spanning many rows), isolated nodes."""
    rng = np.random.default_rng(52)
    N = 1000
    colptr = np.zeros(N + 1, dtype=np.int64)
    rows = []
    for j in range(N):
        r = np.sort(rng.choice(N, 900, replace=False)) if j == 3 else (np.zeros(0, dtype=np.int64) if j % 4 == 0 else np.sort(rng.choice(N, rng.integers(1, 5), replace=False)))
        rows.append(r); colptr[j + 1] = colptr[j] + len(r)
    rowval = np.concatenate(rows).astype(np.int64)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [N])
    dims = ((64, 32, 16), (64, 32, 16))
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 3, len(rowval), N, 1, dims[0])
    _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)


def test_wide_heterogeneous_batch(gn):
    rng = np.random.default_rng(53)
    sizes = rng.integers(32, 200, 24)
    cps, rvs = [], []
    for n in sizes:
        cp, rv = U.er_csc(rng, int(n), int(0.06 * n * n))
        cps.append(cp); rvs.append(rv)
    g = gn.GNGraphBatch.from_csc(cps, rvs, [int(n) for n in sizes])
    dims = ((128, 64, 32), (128, 64, 32))
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)


def test_wide_matches_generic_path(gn):
    rng = np.random.default_rng(54)
    colptr, rowval = U.er_csc(rng, 400, 3000)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [400])
    dims = ((96, 40, 8), (72, 48, 24))
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 1, 3000, 400, 1, dims[0])
    blk = U.block_from_params(gn, p)
    x = U.to_nt(gn, g, ef, nf, gf)
    a, b = blk(x, flags=0), blk(x, flags=1)
    for u, v in ((a.ef, b.ef), (a.nf, b.nf), (a.gf, b.gf)):
        np.testing.assert_allclose(U.from_jl(u), U.from_jl(v), rtol=2e-4, atol=2e-4)


def _dense_indegree_csc(rng, N, lo, hi):
    """every node has lo..hi-1 in-edges: a 128-edge tile's destinations are then a run of at most 128 / lo + 1 consecutive nodes"""
    colptr = np.zeros(N + 1, dtype=np.int64)
    rows = []
    for j in range(N):
        r = np.sort(rng.choice(N, int(rng.integers(lo, hi)), replace=False))
        rows.append(r); colptr[j + 1] = colptr[j] + len(r)
    return colptr, np.concatenate(rows).astype(np.int64)


_PD_LDS_CASES = [
    # (din, dout, R, N)                        what it exercises in k_rows_gemm<..., NL = 3> (launch_gemm_any narrows the column tile of small launches)
    ((128, 64, 32), (128, 64, 32), 1, 3200),   # > 256 row tiles: 128-column tiles
    ((64, 32, 8), (192, 32, 8), 2, 500),       # three 64-column tiles, replicas
    ((32, 16, 0), (160, 16, 4), 1, 500),       # five 32-column tiles
    ((64, 32, 4), (64, 16, 4), 1, 3200),       # one 64-column tile
    ((64, 32, 4), (320, 16, 4), 1, 3200),      # three 128-column tiles, the last one half empty
    ((0, 32, 4), (64, 16, 4), 2, 3200),        # no edge features in: no K loop at all, the LDS-DMA is retired by the explicit barrier
    ((96, 32, 0), (128, 0, 8), 1, 3200),       # K = 96 (three chunks), no node update behind it (no fused per-destination sums)
]


def _pd_lds_case(gn, case):
    din, dout, R, N = _PD_LDS_CASES[case]
    rng = np.random.default_rng(600 + case)
    colptr, rowval = _dense_indegree_csc(rng, N, 6, 15)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [N])
    p = O.make_block_params(rng, din, dout, act=(1, 0, 2))
    ef, nf, gf = U.packed_inputs(rng, R, len(rowval), N, 1, din)
    return p, g, ef, nf, gf


@pytest.mark.parametrize("case", range(len(_PD_LDS_CASES)))
def test_wide_projected_edge_update_destination_rows_through_lds(gn, case, tmp_path):
    """The projected edge update on a graph whose edge tiles' destinations are short runs of rows (every in-degree >= 6: the kernel
    that stages a tile's destination projections in LDS and requests the source projections a pass at a time) against the oracle, and
    bit for bit against the two-stream form of the same kernel (GNX_GEMM_PD_LDS=0, read once per process: a child process)."""
    import os
    import subprocess
    import sys
    p, g, ef, nf, gf = _pd_lds_case(gn, case)
    # (case 0 is 128 -> 128: this test is about k_rows_gemm's two forms, not about k_edge_x6)
    y = _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, flags=gn._lib.FLAG_EDGE_FP32)
    out = str(tmp_path / "two_streams.npz")
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "import graphnets_jl_amd as gn\n"
            "from tests import util as U\n"
            "from tests.test_gpu_wide import _pd_lds_case\n"
            "p, g, ef, nf, gf = _pd_lds_case(gn, %d)\n"
            "y = U.block_from_params(gn, p)(U.to_nt(gn, g, ef, nf, gf))\n"
            "np.savez(%r, **{k: U.from_jl(v) for k, v in (('ef', y.ef), ('nf', y.nf), ('gf', y.gf)) if v is not None})\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), case, out)
    env = dict(os.environ, GNX_GEMM_PD_LDS="0", GNX_EDGE_FP32="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    for name, got in (("ef", y.ef), ("nf", y.nf), ("gf", y.gf)):
        if got is None:
            assert name not in z.files
        else:
            np.testing.assert_array_equal(U.from_jl(got), z[name], err_msg=name)


_EDGE_X6_CASES = [
    # (dout, act, R, graphs as (nodes, edges))      what it exercises in k_edge_x6 (the projected edge update at 128 -> 128 as six bf16 terms)
    ((128, 64, 32), (1, 0, 2), 1, ((900, 12000), (300, 5000), (64, 700))),  # tiles that end at graph boundaries (ragged rows), per-destination sums, column sums
    ((128, 64, 32), (2, 1, 0), 2, ((700, 9000),)),                          # tanh (the run-time activation switch), replicas
    ((128, 0, 8), (1, 0, 0), 1, ((800, 10000),)),                           # no node update behind it: no per-destination sums, column sums from the rows
    ((128, 64, 0), (0, 1, 0), 1, ((800, 10000),)),                          # no graph update: no column sums
    # the narrow form (128 -> at most 32 outputs, one zero-padded slice, single-float addends and stores; the node update adds the ef' rows itself)
    ((3, 4, 5), (1, 1, 0), 1, ((900, 12000), (300, 5000), (64, 700))),      # config 4's decoder widths; ragged tiles, column sums of 3 columns
    ((3, 4, 5), (0, 0, 0), 2, ((700, 9000),)),                              # identity, replicas
    ((7, 5, 0), (3, 1, 0), 1, ((800, 10000),)),                             # sigmoid (act(0) != 0 in the padded columns), 7 columns: two quads, no graph update
    ((30, 0, 2), (1, 0, 1), 1, ((800, 10000),)),                            # 30 columns (last quad partial), no node update
]


def _hub_csc(rng, N, hubs, hub_deg, lo, hi):
    """in-degrees lo..hi-1, except `hubs` nodes with hub_deg in-edges (with repetition-free sources): destinations whose runs cross 64-row chunks and whole tiles"""
    colptr = np.zeros(N + 1, dtype=np.int64)
    rows = []
    hub_at = set(int(v) for v in rng.choice(N, hubs, replace=False))
    for j in range(N):
        d = hub_deg if j in hub_at else int(rng.integers(lo, hi))
        r = np.sort(rng.choice(N, min(d, N), replace=False))
        rows.append(r); colptr[j + 1] = colptr[j] + len(r)
    return colptr, np.concatenate(rows).astype(np.int64)


def test_wide_projected_edge_update_on_bf16_matrix_cores_hub_destinations(gn):
    """k_edge_x6 with destinations of 150-500 in-edges (runs that cross 64-row chunks and whole 128-edge tiles: the node update then adds a node's
    first, second and further partial rows) and isolated nodes, GNCore (gn1 on load) and GNBlock, against the oracle."""
    rng = np.random.default_rng(777)
    N = 700
    colptr, rowval = _hub_csc(rng, N, 6, 500, 0, 9)
    colptr2, rowval2 = _hub_csc(rng, 300, 3, 150, 1, 6)
    g = gn.GNGraphBatch.from_csc([colptr, colptr2], [rowval, rowval2], [N, 300])
    assert g.n_edges >= 4096
    dims = (128, 64, 32)
    p = O.make_block_params(rng, dims, dims, act=(1, 1, 0))
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims)
    gn.profile_reset(); gn.profile_enable(True)
    _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    if not U.default_flags(gn) & gn._lib.FLAG_EDGE_FP32:
        assert "k_edge_x6_prep" in names, names
    pc = O.make_core_params(rng, dims)
    core = U.core_from_params(gn, pc)
    y = core(U.to_nt(gn, g, ef, nf, gf))
    ref, scale = O.core_forward_sparse(pc, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
    for name, got, r, sc in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, sc, name)


@pytest.mark.parametrize("case", range(len(_EDGE_X6_CASES)))
def test_wide_projected_edge_update_on_bf16_matrix_cores(gn, case):
    """GNBlock (128, 64, 32) => (128, ...): the projected edge update as k_edge_x6 — every fp32 product as six bf16 matrix-core terms —
    against the float64 oracle at 1e-5·scale (ef', and nf' / gf' which read its per-destination and column sums), and against the
    fp32-MFMA form (the call's GNX_FLAG_EDGE_FP32: k_rows_gemm) normwise at 2e-6."""
    import os
    dout, act, R, graphs = _EDGE_X6_CASES[case]
    din = (128, 64, 32)
    rng = np.random.default_rng(700 + case)
    cs = [U.er_csc(rng, n, e) for n, e in graphs]
    g = gn.GNGraphBatch.from_csc([c for c, _ in cs], [r for _, r in cs], [n for n, _ in graphs])
    p = O.make_block_params(rng, din, dout, act=act)
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, din)
    gn.profile_reset(); gn.profile_enable(True)
    y = _check(gn, p, g, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    if not U.default_flags(gn) & gn._lib.FLAG_EDGE_FP32:
        assert "k_edge_x6_prep" in names, names
    y0 = U.block_from_params(gn, p)(U.to_nt(gn, g, ef, nf, gf), flags=gn._lib.FLAG_EDGE_FP32)
    for a, b in ((y.ef, y0.ef), (y.nf, y0.nf), (y.gf, y0.gf)):
        if a is None:
            assert b is None
            continue
        a, b = U.from_jl(a).astype(np.float64), U.from_jl(b).astype(np.float64)
        assert np.max(np.abs(a - b)) <= 2e-6 * max(np.max(np.abs(b)), 1e-30)


@pytest.mark.parametrize("R,graphs,core", [(2, ((4300, 9000),), False), (1, ((2500, 6000), (1700, 5000), (300, 700)), False), (1, ((4200, 9500), (150, 400)), True)])
def test_wide_node_projections_on_bf16_matrix_cores(gn, R, graphs, core):
    """(128, 64, 32) => (128, ...) from 4096 nodes on: the node projections Ps = Ws^T nf, Pd = Wd^T nf + b (+ gf fold per graph) of the projected edge
    update run as k_proj_x6 — both tables in one launch, six bf16 matrix-core terms per fp32 product.  GNBlock (replicas of one graph; several graphs: per-graph biases, node
    tiles that end at graph boundaries) and GNCore (gn1 on load from the statistics table) against the float64 oracle at 1e-5·scale, and
    against the build's fp32-MFMA form (GNX_FLAG_EDGE_FP32) normwise at 2e-6."""
    import os
    if U.default_flags(gn) & gn._lib.FLAG_EDGE_FP32:
        pytest.skip("GNX_EDGE_FP32 is set for the whole run: the six-term kernels are switched off")
    rng = np.random.default_rng(1300 + R + len(graphs))
    dims = (128, 64, 32)
    cs = [U.er_csc(rng, n, e) for n, e in graphs]
    g = gn.GNGraphBatch.from_csc([c for c, _ in cs], [r for _, r in cs], [n for n, _ in graphs])
    assert g.n_nodes >= 4096
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, dims)
    nf = nf * 3.0 - 1.0
    x = U.to_nt(gn, g, ef, nf, gf)
    csc = (*g.csc(), g.node_off, g.edge_off)
    if core:
        p = O.make_core_params(rng, dims)
        layer = U.core_from_params(gn, p)
        ref, scale = O.core_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    else:
        p = O.make_block_params(rng, dims, dims, act=(1, 1, 0))
        layer = U.block_from_params(gn, p)
        ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    gn.profile_reset(); gn.profile_enable(True)
    y = layer(x)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    assert "k_proj_x6_prep" in names, names
    for name, got, r_, s_ in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r_, s_, name)
    gn.profile_enable(True)
    y0 = layer(x, flags=gn._lib.FLAG_EDGE_FP32)
    gn.profile_enable(False)
    names0 = set(gn.profile_read()); gn.profile_reset()
    assert "k_proj_x6_prep" not in names0, names0
    for a, b in ((y.ef, y0.ef), (y.nf, y0.nf), (y.gf, y0.gf)):
        a, b = U.from_jl(a).astype(np.float64), U.from_jl(b).astype(np.float64)
        assert np.isfinite(a).all() and np.max(np.abs(a - b)) <= 2e-6 * max(np.max(np.abs(b)), 1e-30)


def _csc_with_hubs_and_isolated(rng, N, E, hubs, isolated):
    """an ER graph whose nodes `hubs` receive an edge from EVERY node (their in-edges run through several 64-row chunks of the edge tiles: several
    per-destination partial rows to add up) and whose nodes `isolated` receive none"""
    colptr, rowval = U.er_csc(rng, N, E)
    cols = [rowval[colptr[j]:colptr[j + 1]] for j in range(N)]
    for j in hubs:
        cols[j] = np.arange(N, dtype=np.int64)
    for j in isolated:
        cols[j] = np.zeros(0, dtype=np.int64)
    cp = np.zeros(N + 1, dtype=np.int64)
    cp[1:] = np.cumsum([len(c) for c in cols])
    return cp, np.concatenate(cols)


@pytest.mark.parametrize("R,act_n,core,graphs", [(1, 1, False, ((4300, 9000),)), (2, 0, False, ((4400, 20000),)), (1, 1, False, ((2500, 6000), (1700, 5000), (300, 700))),
                                                 (1, 1, True, ((4200, 9500), (150, 400)))])
def test_wide_node_update_on_bf16_matrix_cores(gn, R, act_n, core, graphs):
    """(128, 64, 32) => (., 64, .) from 4096 nodes on: the node update nf' = act(Wn^T [sum of in-edge ef' | nf] + b (+ gf fold)) runs as k_node_x6 — six bf16
    matrix-core terms per fp32 product, the summed rows taken from the edge kernel's per-destination partial sums (hub nodes: several parts in chunk order;
    nodes without in-edges: zeros), per-tile column sums for the graph update.  GNBlock (replicas; several graphs: per-graph biases, tiles ending at graph
    boundaries) and GNCore (gn1 on load) against the float64 oracle at 1e-5·scale and against the fp32-instruction form (GNX_FLAG_PROJ_FP32 switches the
    node-side kernels back) normwise; the prepared form bit-identical and without a preparation launch."""
    import torch
    F = gn._lib
    if U.default_flags(gn) & (F.FLAG_EDGE_FP32 | F.FLAG_PROJ_FP32 | F.FLAG_EDGE_N):
        pytest.skip("the six-term node-side kernels are switched off for the whole run")
    rng = np.random.default_rng(1700 + R + len(graphs) + act_n)
    dims = (128, 64, 32)
    cs = []
    for i, (n, e) in enumerate(graphs):
        cs.append(_csc_with_hubs_and_isolated(rng, n, e, hubs=(7, n // 2) if i == 0 else (), isolated=(0, 3, n - 1)))
    g = gn.GNGraphBatch.from_csc([c for c, _ in cs], [r for _, r in cs], [n for n, _ in graphs])
    assert g.n_nodes >= 4096
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, dims)
    nf = nf * 2.0 - 0.5
    x = U.to_nt(gn, g, ef, nf, gf)
    csc = (*g.csc(), g.node_off, g.edge_off)
    if core:
        p = O.make_core_params(rng, dims)
        layer = U.core_from_params(gn, p)
        ref, scale = O.core_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    else:
        p = O.make_block_params(rng, dims, dims, act=(1, act_n, 0))
        layer = U.block_from_params(gn, p)
        ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    gn.profile_reset(); gn.profile_enable(True)
    y = layer(x)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    assert "k_node_x6_prep" in names, names
    for name, got, r_, s_ in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r_, s_, name)
    gn.profile_enable(True)
    y0 = layer(x, flags=F.FLAG_PROJ_FP32)
    gn.profile_enable(False)
    names0 = set(gn.profile_read()); gn.profile_reset()
    assert "k_node_x6_prep" not in names0 and "k_proj_x6_prep" not in names0, names0
    for name, a, b in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), (y0.ef, y0.nf, y0.gf)):
        U.assert_same_formula(U.from_jl(a), U.from_jl(b), name)
    layer.prepare()
    gn.profile_enable(True)
    yp = layer(x)
    gn.profile_enable(False)
    names_p = set(gn.profile_read()); gn.profile_reset()
    assert not {n_ for n_ in names_p if n_.endswith("_prep")}, names_p
    for a, b in zip((y.ef, y.nf, y.gf), (yp.ef, yp.nf, yp.gf)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("din,act,R,graphs", [((10, 5, 0), (0, 0, 0), 1, ((4300, 9000),)), ((10, 5, 3), (1, 1, 0), 1, ((2500, 6000), (1700, 5000), (300, 700))),
                                             ((10, 5, 0), (2, 0, 0), 2, ((700, 4137),))])
def test_wide_encoder_edge_update_on_bf16_matrix_cores(gn, din, act, R, graphs):
    """GNBlock (10, 5, .) => (128, 64, 32) — README ex.3's / config 4's ENCODER — from 4096 edges on: the unprojected edge update
    W^T [ef ; nf[src] ; nf[dst]] + b runs as the encoder form of k_edge_x6 (the row's 20 inputs assembled in the lane's registers as one zero-padded
    K = 32, six bf16 matrix-core terms per fp32 product, the per-destination and column sums of the projected form) — against the float64 oracle
    at 1e-5*scale (ef', and nf' / gf' which read its sums), against the fp32-MFMA form of the same call (GNX_FLAG_EDGE_FP32: k_rows_gemm's packed
    element loader) normwise at 2e-6, hub destinations (runs that cross chunks and tiles), several graphs with gf (per-graph bias), replicas, a
    transcendental activation, a ragged edge count — and with prepared parameters (no preparation launch, same bits)."""
    import torch
    rng = np.random.default_rng(8800 + din[2] + R)
    cs = [U.er_csc(rng, n, e) for n, e in graphs]
    if len(graphs) == 1 and R == 1:  # hub destinations on the single-graph case
        cs = [_hub_csc(rng, graphs[0][0], 5, 400, 0, 4)]
    g = gn.GNGraphBatch.from_csc([c for c, _ in cs], [r for _, r in cs], [n for n, _ in graphs])
    assert g.n_edges >= 4096
    dout = (128, 64, 32)
    p = O.make_block_params(rng, din, dout, act=act)
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, din)
    ef, nf = (ef * 2 - 0.5).astype(np.float32), (nf * 3 - 1).astype(np.float32)
    blk = U.block_from_params(gn, p)
    x = U.to_nt(gn, g, ef, nf, gf)
    gn.profile_reset(); gn.profile_enable(True)
    y = blk(x)
    gn.profile_enable(False)
    prof = gn.profile_read(); gn.profile_reset()
    if not U.default_flags(gn) & gn._lib.FLAG_EDGE_FP32:
        assert "k_edge_x6_prep" in prof, set(prof)  # (the encoder form's weight preparation goes by that name)
    ref, scale = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
    for name, got, r_, s_ in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r_, s_, name)
    gn.profile_enable(True)
    y0 = blk(x, flags=gn._lib.FLAG_EDGE_FP32)
    gn.profile_enable(False)
    assert "k_edge_x6_prep" not in set(gn.profile_read()); gn.profile_reset()
    for a, b in ((y.ef, y0.ef), (y.nf, y0.nf), (y.gf, y0.gf)):
        a_, b_ = U.from_jl(a).astype(np.float64), U.from_jl(b).astype(np.float64)
        assert np.isfinite(a_).all() and np.max(np.abs(a_ - b_)) <= 2e-6 * max(np.max(np.abs(b_)), 1e-30)
    blk.prepare()
    assert blk._prepared.nbytes() == 4 * 6 * 1024
    gn.profile_enable(True)
    y1 = blk(x)
    gn.profile_enable(False)
    assert not {n for n in gn.profile_read() if n.endswith("_prep")}; gn.profile_reset()
    for a, b in ((y.ef, y1.ef), (y.nf, y1.nf), (y.gf, y1.gf)):
        assert torch.equal(a, b)

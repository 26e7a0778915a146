"""GNGraphBatch construction from CSC on the device (csrc/gnx_build_csc.hip) against the host builder (csrc/gnx_graphs.cpp::finalize), which
stays as the validator: every device table bit-identical (colptr, rowval, offsets, both tile tables, the pack table), the same info, the same
errors for malformed input, and the same forward results — and the matrix-core path's ten tables (csrc/gnx_build_csc.hip::
build_wide_tables_on_device vs gnx_graphs.cpp::build_wide_tables).  The host builders run in a child process with GNX_BUILD_CSC_DEVICE=0
GNX_BUILD_WIDE_DEVICE=0 (the switches are read once per process)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "c2_small": dict(kind="c2", N=20_000, E=200_000),
    "c2_full": dict(kind="c2", N=100_000, E=1_000_000),
    "hetero_512": dict(kind="hetero", seed=3, G=512, E=1_000_000),
    "hetero_4096": dict(kind="hetero", seed=5, G=4096, E=1_000_000),
    "hetero_dense": dict(kind="hetero", seed=11, G=300, E=2_000_000),   # in-degrees up to ~190: single-node tiles, wave tiles on the edge cap
    "edgeless_mix": dict(kind="mix"),
    "dense_uint8_600": dict(kind="dense", seed=13, G=600, E=400_000),   # the reference's own input form: dense 0/1 matrices (>= 2^22 entries: scan + compaction on the GPU)
}


def _graphs(spec):
    if spec["kind"] == "c2":
        return bench.make_c2(seed=2, N=spec["N"], E=spec["E"])
    if spec["kind"] == "hetero":
        return bench.make_hetero(spec["seed"], spec["G"], spec["E"])
    rng = np.random.default_rng(4)
    colptrs, rowvals, nn = bench.make_hetero(21, 200, 90_000)
    for g in rng.choice(200, 40, replace=False):  # graphs without a single edge, single-node graphs, a self loop
        n = int(rng.integers(1, 4))
        colptrs[g], rowvals[g], nn[g] = np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64), n
    colptrs[0], rowvals[0], nn[0] = np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int64), 1
    return colptrs, rowvals, nn


def _tables(spec, index_dtype="int64"):
    """sha-free: the raw bytes of the nine device tables + the handle's info, as a dict of numpy arrays"""
    import graphnets_jl_amd as gn
    lib = gn._lib.load()
    if spec["kind"] == "dense":
        colptrs, rowvals, nn = bench.make_hetero(spec["seed"], spec["G"], spec["E"])
        adjs = []
        for cp, rv, n in zip(colptrs, rowvals, nn):
            a = np.zeros((n, n), dtype=np.uint8 if index_dtype == "int64" else np.float32)  # (two element kinds in place of the two index widths)
            a[rv, np.repeat(np.arange(n), np.diff(cp))] = 1
            adjs.append(a)
        assert sum(a.size for a in adjs) >= 1 << 22
        g = gn.GNGraphBatch(adjs)
    else:
        colptrs, rowvals, nn = _graphs(spec)
        cat = lambda parts: np.concatenate(parts).astype(index_dtype)
        g = gn.GNGraphBatch.from_csc_packed(cat(colptrs), cat(rowvals), nn)
    out = {}
    for which, name in enumerate(("colptr", "rowval", "node_off", "edge_off", "tile_off", "tiles", "wtile_off", "wtiles", "packs")):
        n = C.c_int64(0)
        gn._lib.check(lib.gnx_graphs_get_table(g._h, which, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value // 4, 1), dtype=np.int32)
        gn._lib.check(lib.gnx_graphs_get_table(g._h, which, buf.ctypes.data, buf.nbytes, C.byref(n)))
        out[name] = buf[: n.value // 4]
    # the matrix-core path's tables (built on the spot): device builder (kernels over the handle's device arrays) vs host builder
    for which, name in zip(range(9, 20), ("etiles", "ntiles", "gtiles", "etile_off", "ntile_off", "edge_dst", "chunk_row0", "node_agg_row", "node_agg_parts",
                                          "node_agg_chunk", "wide_info")):
        n = C.c_int64(0)
        gn._lib.check(lib.gnx_graphs_get_table(g._h, which, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value // 4, 1), dtype=np.int32)
        gn._lib.check(lib.gnx_graphs_get_table(g._h, which, buf.ctypes.data, buf.nbytes, C.byref(n)))
        out[name] = buf[: n.value // 4]
    out["info"] = np.array([g.n_graphs, g.n_nodes, g.n_edges, g.node_block_size, g.n_tiles, g.max_in_degree], dtype=np.int64)
    cp, rv = g.csc()  # the lazily downloaded int64 host copies
    out["host_colptr"], out["host_rowval"] = cp, rv
    return out


def _table(gn, g, which):
    """one device table of a handle (gnx_graphs_get_table) as int32 words"""
    lib = gn._lib.load()
    n = C.c_int64(0)
    gn._lib.check(lib.gnx_graphs_get_table(g._h, which, None, 0, C.byref(n)))
    buf = np.zeros(max(n.value // 4, 1), dtype=np.int32)
    gn._lib.check(lib.gnx_graphs_get_table(g._h, which, buf.ctypes.data, buf.nbytes, C.byref(n)))
    return buf[: n.value // 4].copy()


def _child(name, path):
    np.savez(path, **_tables(CASES[name]))


@pytest.mark.parametrize("name", list(CASES))
def test_device_builder_tables_are_bit_identical_to_the_host_builders(name, tmp_path):
    path = str(tmp_path / "host.npz")
    env = dict(os.environ, GNX_BUILD_CSC_DEVICE="0", GNX_BUILD_WIDE_DEVICE="0")
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from tests import test_gpu_build as T; T._child({name!r}, {path!r})"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    host = np.load(path)
    for dt in ("int64", "int32"):
        dev = _tables(CASES[name], dt)
        for k in host.files:
            assert dev[k].shape == host[k].shape and np.array_equal(dev[k], host[k]), f"{name} ({dt} indices): table `{k}` differs between the device and the host builder"


def test_device_builder_rejects_what_the_host_builder_rejects():
    """malformed input through the device builder: same status code and message as the host pass, first error in column order"""
    import graphnets_jl_amd as gn
    colptrs, rowvals, nn = bench.make_hetero(7, 64, 120_000)
    cpc, rvc = np.concatenate(colptrs), np.concatenate(rowvals)

    def build(cp, rv):
        return gn.GNGraphBatch.from_csc_packed(cp, rv, nn)
    build(cpc, rvc)  # well-formed
    msgs = {}
    for what, (cp, rv) in {
        "rowval out of range": (cpc, np.where(np.arange(rvc.size) == 77_777, 1 << 40, rvc)),
        "rowval negative": (cpc, np.where(np.arange(rvc.size) == 5, -3, rvc)),
        "rowval not increasing": (cpc, np.concatenate([rvc[:1000], rvc[999:1000], rvc[1001:]]) if rvc[999] != rvc[1000] else rvc),
    }.items():
        with pytest.raises(gn._lib.GnxError) as e:
            build(cp, rv)
        assert e.value.code == gn._lib.ERR_CSC and "rowval out of range or not strictly increasing" in str(e.value), what
        msgs[what] = str(e.value)
    # colptr: a decrease inside a graph; a rise above colptr[n] that comes back (nothing may be read or written past the graph's slice)
    off = np.cumsum([0] + [n + 1 for n in nn])
    bad = cpc.copy(); bad[off[3] + 5] = bad[off[3] + 4] - 1
    with pytest.raises(gn._lib.GnxError) as e:
        build(bad, rvc)
    assert e.value.code == gn._lib.ERR_CSC and "colptr must be non-decreasing" in str(e.value)
    bad = cpc.copy(); bad[off[63] + 1: off[63] + 3] = bad[off[64] - 1] + 1000
    with pytest.raises(gn._lib.GnxError) as e:
        build(bad, rvc)
    assert e.value.code == gn._lib.ERR_CSC and "colptr must be non-decreasing" in str(e.value)
    # the FIRST error in column order wins, as in the host pass: a colptr error in graph 10 and a rowval error in graph 2 -> rowval
    bad = cpc.copy(); bad[off[10] + 2] = bad[off[10] + 1] - 1
    e_off = np.cumsum([0] + [len(r) for r in rowvals])
    rbad = rvc.copy(); rbad[e_off[2]] = 10 ** 9
    with pytest.raises(gn._lib.GnxError) as e:
        build(bad, rbad)
    assert "rowval out of range" in str(e.value)
    # lengths are checked before anything is read
    with pytest.raises(gn._lib.GnxError):
        build(cpc, rvc[:-1])
    with pytest.raises(ValueError):
        build(cpc[:-1], rvc)


def test_device_builder_abandons_the_tiling_of_an_oscillating_colptr():
    """a colptr that swings between 0 and E passes the O(G) host pre-check (first entry = base, last = E) and would make the greedy
    tiling emit ~3 tiles per 130 nodes — several times the bound the tile tables are sized with.  The tiling kernels must not run on it: the call is
    rejected as malformed CSC, nothing is written past the tables (the next, well-formed batch of the same size — which takes the recycled arena —
    builds the same tables as before), and the device stays usable."""
    import graphnets_jl_amd as gn
    n, eg = 100_000, 1000
    cp = np.zeros(n + 1, dtype=np.int64)
    pat = np.array([0, 0] + [eg] * 128, dtype=np.int64)
    cp[1:] = np.tile(pat, n // len(pat) + 1)[:n]
    cp[0], cp[n] = 0, eg
    rv = np.arange(eg, dtype=np.int64)  # (every column that announces edges announces rows 0..999: valid; the first error in column order is colptr's first fall)
    good_cp, good_rv, good_n = bench.make_c2(seed=4, N=n, E=200_000)
    ref = gn.GNGraphBatch.from_csc_packed(good_cp[0], good_rv[0], good_n)
    want = [_table(gn, ref, w) for w in (0, 1, 5, 7)]
    del ref
    for bits in (np.int64, np.int32):
        with pytest.raises(gn._lib.GnxError) as e:
            gn.GNGraphBatch.from_csc_packed(cp.astype(bits), rv.astype(bits), [n])
        assert e.value.code == gn._lib.ERR_CSC and "colptr must be non-decreasing" in str(e.value)
        again = gn.GNGraphBatch.from_csc_packed(good_cp[0], good_rv[0], good_n)
        for w, t in zip((0, 1, 5, 7), want):
            assert np.array_equal(_table(gn, again, w), t)
        del again


def test_two_host_threads_build_batches_on_one_device_concurrently():
    """the device builder's scratch buffer and stream are per device, not per handle — two host threads that batch on the same
    device (data-loader workers) must not see each other's arrays.  Each thread builds its own batches (different sizes, so the scratch is also
    re-sized under load) many times; every handle's tables equal the ones built alone."""
    import threading
    import graphnets_jl_amd as gn
    import torch
    specs = [(11, 300, 400_000), (12, 900, 150_000), (13, 64, 700_000), (14, 2000, 90_000)]
    batches = [bench.make_hetero(*sp) for sp in specs]
    packed = [(np.concatenate(c), np.concatenate(r), nn) for c, r, nn in batches]
    which = (0, 1, 4, 5, 6, 7)
    alone = []
    for cpc, rvc, nn in packed:
        g = gn.GNGraphBatch.from_csc_packed(cpc, rvc, nn)
        alone.append([_table(gn, g, w) for w in which])
        del g
    errors = []

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            for it in range(12):
                k = (tid + 2 * it) % len(packed) if tid else (it * 3 + 1) % len(packed)
                cpc, rvc, nn = packed[k]
                g = gn.GNGraphBatch.from_csc_packed(cpc, rvc, nn)
                for w, t in zip(which, alone[k]):
                    if not np.array_equal(_table(gn, g, w), t):
                        errors.append((tid, it, k, w))
                del g
        except Exception as ex:  # noqa: BLE001
            errors.append((tid, repr(ex)))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:5]


def test_dense_packed_input_builds_the_same_batch_from_host_pinned_and_device_memory():
    """`batch`'s own input form — dense 0/1 matrices (src/batch.jl:53-64) — as ONE buffer (GNGraphBatch.from_dense_packed ->
    gnx_graphs_create_dense_packed): pageable numpy memory, a pinned tensor (one DMA) and a device tensor (no copy) give the very tables the
    list-of-matrices constructor gives; bool and float32 elements too; a value other than 0 / 1 and a wrong length are rejected; `adj_mats`
    of the packed batch are views of the buffer (what unbatch hands back)."""
    import torch
    import graphnets_jl_amd as gn
    colptrs, rowvals, nn = bench.make_hetero(13, 600, 400_000)
    adjs = []
    for cp, rv, n in zip(colptrs, rowvals, nn):
        a = np.zeros((n, n), dtype=np.uint8)
        a[rv, np.repeat(np.arange(n), np.diff(cp))] = 1
        adjs.append(a)
    cat = np.concatenate([a.reshape(-1) for a in adjs])
    assert cat.size >= 1 << 22  # (the device builder's threshold)
    which = (0, 1, 2, 3, 4, 5, 6, 7, 8)
    ref = gn.GNGraphBatch(adjs)
    want = [_table(gn, ref, w) for w in which]
    forms = {"pageable": cat, "bool": cat.astype(bool), "float32": cat.astype(np.float32), "pinned": torch.from_numpy(cat).pin_memory(),
             "device": torch.from_numpy(cat).cuda()}
    for name, buf in forms.items():
        g = gn.GNGraphBatch.from_dense_packed(buf, nn)
        assert (g.n_graphs, g.n_nodes, g.n_edges) == (ref.n_graphs, ref.n_nodes, ref.n_edges), name
        for w, t in zip(which, want):
            assert np.array_equal(_table(gn, g, w), t), (name, w)
        if name == "pageable":
            assert all(np.array_equal(a, b) for a, b in zip(g.adj_mats[:5] + g.adj_mats[-5:], adjs[:5] + adjs[-5:]))
        del g
    small = gn.GNGraphBatch.from_dense_packed(np.concatenate([a.reshape(-1) for a in adjs[:3]]), nn[:3])  # (below the threshold: the host scan)
    ref3 = gn.GNGraphBatch(adjs[:3])
    assert np.array_equal(_table(gn, small, 1), _table(gn, ref3, 1))
    small_dev = gn.GNGraphBatch.from_dense_packed(torch.from_numpy(np.concatenate([a.reshape(-1) for a in adjs[:3]])).cuda(), nn[:3])  # (device memory: always the kernels)
    assert np.array_equal(_table(gn, small_dev, 1), _table(gn, ref3, 1)) and np.array_equal(_table(gn, small_dev, 7), _table(gn, ref3, 7))
    bad = cat.copy(); bad[12345] = 2
    for buf in (bad, torch.from_numpy(bad).cuda()):
        with pytest.raises(gn._lib.GnxError) as e:
            gn.GNGraphBatch.from_dense_packed(buf, nn)
        assert e.value.code == gn._lib.ERR_ADJ_VALUE
    with pytest.raises(ValueError):
        gn.GNGraphBatch.from_dense_packed(cat[:-1], nn)


def test_forward_on_a_device_built_batch_equals_the_oracle_and_lazy_host_tables_work():
    import torch
    import graphnets_jl_amd as gn
    from oracle import gn_oracle as O
    from tests import util as U
    colptrs, rowvals, nn = bench.make_hetero(9, 256, 300_000)
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn)
    rng = np.random.default_rng(1)
    for dims in (((10, 5, 3), (3, 4, 5)), ((40, 36, 8), (36, 40, 8))):  # the fused narrow kernel; the matrix-core path (its tables come from the lazily downloaded host CSC)
        p = O.make_block_params(rng, *dims)
        ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
        y = U.block_from_params(gn, p)(U.to_nt(gn, g, ef, nf, gf))
        ref, scale = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
        for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
            U.assert_close(U.from_jl(got), r, s, name)

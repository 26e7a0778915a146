"""Run-time specialised fused kernels (csrc/gnx_jit.cpp): width sets outside the ahead-of-time list are compiled with
hiprtc on first use and must agree with the float64 oracle (1e-5 of the magnitude bound) like every other path."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import gn_oracle as O
from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


def _stats():
    lib = sys.modules["graphnets_jl_amd._lib"].load()
    out = (C.c_int64 * 4)()
    assert lib.gnx_jit_stats(out) == 0
    return dict(zip(("compiled", "disk_hits", "failures", "capture_misses"), out))


def _check(gn, p, g, ef, nf, gf, flags=0):
    blk = U.block_from_params(gn, p)
    y = blk(U.to_nt(gn, g, ef, nf, gf), flags=flags)
    colptr, rowval = g.csc()
    ref, scale = O.block_forward_sparse(p, (colptr, rowval, g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name)
    return y


WIDTHS = [((7, 3, 2), (5, 6, 1)), ((1, 1, 1), (1, 1, 1)), ((16, 12, 4), (12, 16, 5)), ((0, 9, 0), (13, 0, 2)),
          ((11, 0, 0), (2, 7, 3)), ((0, 0, 6), (4, 3, 0)), ((6, 15, 1), (1, 12, 9)), ((13, 2, 7), (0, 5, 4))]


@pytest.mark.parametrize("dims", WIDTHS, ids=[f"{a}-{b}".replace(" ", "") for a, b in WIDTHS])
def test_unlisted_width_sets_run_the_fused_kernel(gn, dims):
    """heterogeneous batch (many tiles, several graphs per tile boundary), two activations; the fused path must have been
    compiled (stats) and agree with the oracle AND with the generic kernels."""
    before = _stats()
    rng = np.random.default_rng(sum(dims[0]) * 31 + sum(dims[1]))
    sizes = rng.integers(3, 200, 40)
    cps, rvs = zip(*(U.er_csc(rng, int(n), int(0.08 * n * n) + 1) for n in sizes))
    g = gn.GNGraphBatch.from_csc(list(cps), list(rvs), [int(n) for n in sizes])
    p = O.make_block_params(rng, *dims)
    p["act_e"], p["act_n"], p["act_g"] = 1, 2, 0
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    y = _check(gn, p, g, ef, nf, gf)
    after = _stats()
    assert after["failures"] == before["failures"] == 0
    assert after["compiled"] + after["disk_hits"] == before["compiled"] + before["disk_hits"] + 1
    yg = _check(gn, p, g, ef, nf, gf, flags=1)
    for a, b in ((y.ef, yg.ef), (y.nf, yg.nf), (y.gf, yg.gf)):
        assert (a is None) == (b is None)


def test_hub_node_and_replicas(gn):
    """single-node tiles walked in chunks (in-degree 3000) + isolated nodes, shared adjacency with 3 replicas"""
    rng = np.random.default_rng(77)
    N = 3100
    rows = [np.sort(rng.choice(N, 3000, replace=False)) if j == 5 else
            (np.zeros(0, dtype=np.int64) if j % 4 == 0 else np.sort(rng.choice(N, rng.integers(1, 5), replace=False))) for j in range(N)]
    colptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    rowval = np.concatenate(rows).astype(np.int64)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [N])
    dims = ((9, 4, 2), (6, 2, 3))
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 3, len(rowval), N, 1, dims[0])
    _check(gn, p, g, ef, nf, gf)
    assert _stats()["failures"] == 0


def test_first_use_inside_a_capture_never_compiles(gn):
    """Graphed warms the model up before capturing, so the specialised kernels are in place when the capture starts
    (capture_misses stays 0) and the replay matches the eager result bit for bit."""
    import torch
    rng = np.random.default_rng(78)
    colptr, rowval = U.er_csc(rng, 1500, 15000)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [1500])
    dims = ((5, 7, 3), (4, 2, 6))
    blk = U.block_from_params(gn, O.make_block_params(rng, *dims))
    ef, nf, gf = U.packed_inputs(rng, 1, 15000, 1500, 1, dims[0])
    x = U.to_nt(gn, g, ef, nf, gf)
    before = _stats()
    graphed = gn.Graphed(blk, x)
    y = graphed(x)
    torch.cuda.synchronize()
    e = blk(x)
    after = _stats()
    assert after["capture_misses"] == before["capture_misses"] and after["failures"] == 0
    for a, b in ((y.ef, e.ef), (y.nf, e.nf), (y.gf, e.gf)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())


def test_code_objects_can_be_kept_on_disk(gn, tmp_path):
    os.environ["GNX_JIT_CACHE"] = str(tmp_path)
    try:
        rng = np.random.default_rng(79)
        colptr, rowval = U.er_csc(rng, 300, 2000)
        g = gn.GNGraphBatch.from_csc([colptr], [rowval], [300])
        dims = ((3, 8, 1), (2, 5, 2))
        p = O.make_block_params(rng, *dims)
        ef, nf, gf = U.packed_inputs(rng, 1, 2000, 300, 1, dims[0])
        _check(gn, p, g, ef, nf, gf)
        files = [f for f in os.listdir(tmp_path) if f.startswith("gnx_wave_3_8_1_2_5_")]
        assert len(files) == 1 and os.path.getsize(tmp_path / files[0]) > 4096
        # compile-only requests (several-graphs variant of another width set): the first compiles and writes the file,
        # the second is served from it
        lib = sys.modules["graphnets_jl_amd._lib"]
        n = C.c_size_t(0)
        before = _stats()
        assert lib.load().gnx_jit_precompile(C.byref(lib.BlockParams(3, 8, 1, 2, 6, 2)), 128, C.byref(n)) == 0
        mid = _stats()
        assert mid["compiled"] == before["compiled"] + 1 and n.value > 4096
        assert lib.load().gnx_jit_precompile(C.byref(lib.BlockParams(3, 8, 1, 2, 6, 2)), 128, C.byref(n)) == 0
        assert _stats()["disk_hits"] == mid["disk_hits"] + 1 and n.value > 4096
    finally:
        del os.environ["GNX_JIT_CACHE"]


def test_jit_can_be_disabled(gn):
    """Run-time specialisation switched off for a process (env GNX_JIT=0, read once by the library; GNX_FLAG_NO_JIT on a call does the same for
    that call): nothing is compiled — the statistics do not move — and the generic kernels give the oracle's result."""
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r)\n"
            "import graphnets_jl_amd as gn\n"
            "from tests import util as U\n"
            "from tests.test_gpu_jit import _stats\n"
            "from oracle import gn_oracle as O\n"
            "rng = np.random.default_rng(80)\n"
            "colptr, rowval = U.er_csc(rng, 300, 2000)\n"
            "g = gn.GNGraphBatch.from_csc([colptr], [rowval], [300])\n"
            "dims = ((2, 9, 1), (3, 3, 3))\n"
            "p = O.make_block_params(rng, *dims)\n"
            "ef, nf, gf = U.packed_inputs(rng, 1, 2000, 300, 1, dims[0])\n"
            "before = _stats()\n"
            "y = U.block_from_params(gn, p)(U.to_nt(gn, g, ef, nf, gf))\n"
            "ref, scale = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)\n"
            "[U.assert_close(U.from_jl(a), r, s, n) for n, a, r, s in zip(('ef', 'nf', 'gf'), (y.ef, y.nf, y.gf), ref, scale)]\n"
            "print(json.dumps([before, _stats(), U.default_flags(gn)]))\n") % ROOT
    import subprocess
    import sys
    import json
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GNX_JIT="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    before, after, dflt = json.loads(r.stdout.strip().splitlines()[-1])
    assert before == after and dflt & gn._lib.FLAG_NO_JIT  # (the environment's default, read once by that process: nothing was compiled)


def test_core_feedforward_kernel_is_kept_on_disk_too(tmp_path):
    """The run-time specialised one-launch FeedForward kernel of a narrow GNCore (k_core_post3) goes through the same disk cache: a
    second PROCESS loads the code object instead of compiling it."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, json, ctypes as C, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import graphnets_jl_amd as gn\n"
        "from oracle import gn_oracle as O\n"
        "from tests import util as U\n"
        "rng = np.random.default_rng(3)\n"
        "cp, rv = U.er_csc(rng, 70000, 140000)\n"
        "g = gn.GNGraphBatch.from_csc([cp], [rv], [70000])\n"
        "core = U.core_from_params(gn, O.make_core_params(rng, (5, 4, 3)))\n"
        "y = core(U.to_nt(gn, g, *U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, (5, 4, 3))))\n"
        "s = (C.c_int64 * 4)(); gn._lib.load().gnx_jit_stats(s)\n"
        "print(json.dumps(dict(compiled=s[0], disk_hits=s[1], failures=s[2], ef=float(y.ef.sum()))))\n")
    env = dict(os.environ, GNX_JIT_CACHE=str(tmp_path))
    runs = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    assert any(f.startswith("gnx_post3_5_4_3_") for f in os.listdir(tmp_path)), os.listdir(tmp_path)
    assert runs[0]["compiled"] >= 1 and runs[0]["failures"] == 0
    assert runs[1]["compiled"] == 0 and runs[1]["disk_hits"] >= 1 and runs[1]["failures"] == 0, runs
    assert runs[0]["ef"] == runs[1]["ef"]

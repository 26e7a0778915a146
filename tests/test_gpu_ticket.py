"""Single-launch graph update (GNX_FLAG_WS_TICKETS): the workgroup / wavefront whose ticket add comes last reduces the
partial sums inside k_block_wave.  It must give the SAME BITS as the two-launch form (k_graph_t) — the reduction order depends
only on the tile table — on every launch: first use of a workspace, back-to-back launches (the counters reset themselves),
hipGraph replays under load, one graph and many graphs, replicas, run-time specialised width sets."""
import ctypes as C

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


@pytest.fixture(autouse=True)
def _single_launch_at_every_size(monkeypatch):
    """By default only launch-bound batches (<= 1024 wave tiles) take the single-launch form; these tests want it everywhere."""
    monkeypatch.setenv("GNX_TICKET_MAX_ROWS", "100000000")


def _plan_pair(gn, g, dims, seed, R=1):
    """(block, single-launch plan, two-launch plan, inputs) on the same handle and parameters."""
    import torch
    rng = np.random.default_rng(seed)
    p = O.make_block_params(rng, *dims)
    blk = U.block_from_params(gn, p)
    one, two = gn.BlockPlan(blk, g, R=R), gn.BlockPlan(blk, g, R=R)
    two.flags &= ~gn._lib.FLAG_WS_TICKETS
    assert one.flags & gn._lib.FLAG_WS_TICKETS
    x = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    xt = tuple(None if a is None else torch.from_numpy(a).to(g.device) for a in x)
    return p, one, two, x, xt


def _kernels_of(gn, fn):
    import torch
    gn.profile_reset(); gn.profile_enable(True)
    fn(); torch.cuda.synchronize()
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    return names


@pytest.mark.parametrize("case", ["c2-readme", "hetero512-readme", "hetero4096-readme", "er-jit-dims", "replicas"])
def test_single_launch_equals_two_launch_bitwise(gn, case):
    import torch
    R = 1
    if case == "c2-readme":
        g = gn.GNGraphBatch.from_csc(*bench.make_c2()); dims = ((10, 5, 0), (3, 4, 5))
    elif case == "hetero512-readme":
        g = gn.GNGraphBatch.from_csc(*bench.make_hetero(3)); dims = ((10, 5, 0), (3, 4, 5))
    elif case == "hetero4096-readme":
        g = gn.GNGraphBatch.from_csc(*bench.make_hetero(5, 4096, 1_000_000)); dims = ((10, 5, 3), (3, 4, 5))
    elif case == "er-jit-dims":
        rng = np.random.default_rng(5)
        g = gn.GNGraphBatch.from_csc(*[[a] for a in U.er_csc(rng, 3000, 40000)], [3000]); dims = ((7, 3, 2), (5, 6, 9))
    else:
        rng = np.random.default_rng(6)
        g = gn.GNGraphBatch.from_csc(*[[a] for a in U.er_csc(rng, 5000, 30000)], [5000]); dims = ((10, 5, 0), (3, 4, 5)); R = 3
    p, one, two, x, xt = _plan_pair(gn, g, dims, 11, R)
    o1, o2 = one.outputs(), two.outputs()
    assert "k_graph_t" not in _kernels_of(gn, lambda: one(*xt, *o1)), "the single-launch form must not launch k_graph_t"
    assert "k_graph_t" in _kernels_of(gn, lambda: two(*xt, *o2))
    for rep in range(3):  # back-to-back launches on the same workspace: the counters reset themselves
        for o in o1:
            o.fill_(float("nan"))
        one(*xt, *o1)
        torch.cuda.synchronize()
        for a, b in zip(o1, o2):
            assert torch.equal(a, b), f"{case}: launch {rep} differs from the two-launch form"
    # and against the oracle (the graph update is the part that changed)
    ref, scale = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), *x, return_scale=True)
    U.assert_close(o1[2].cpu().numpy(), ref[2], scale[2], "gf")


def test_ticket_handoff_under_load_in_a_hipgraph(gn):
    """200 steps in ONE hipGraph over rotating inputs (every launch overlaps the tail of the previous one's last workgroups),
    replayed several times: every gf' equals the two-launch result of the same inputs bit for bit."""
    import torch
    for maker in (bench.make_c2, lambda: bench.make_hetero(3)):
        g = gn.GNGraphBatch.from_csc(*maker())
        dims = ((10, 5, 0), (3, 4, 5))
        p, one, two, _, _ = _plan_pair(gn, g, dims, 12)
        gen = torch.Generator(device=g.device); gen.manual_seed(3)
        nset, steps = 4, 200
        sets = [(torch.rand((1, g.n_edges, 10), generator=gen, device=g.device), torch.rand((1, g.n_nodes, 5), generator=gen, device=g.device))
                for _ in range(nset)]
        want = []
        for ef, nf in sets:
            o = two.outputs(); two(ef, nf, None, *o); want.append(o[2].clone())
        eo, no, _ = one.outputs()
        gfs = torch.zeros((steps, g.n_graphs, 5), device=g.device)
        ws = [one.new_workspace() for _ in range(nset)]
        for i in range(2):
            one(*sets[i], None, eo, no, gfs[i:i + 1], ws=ws[i])
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            for i in range(steps):
                one(*sets[i % nset], None, eo, no, gfs[i:i + 1], ws=ws[i % nset])
        for rep in range(5):
            gfs.fill_(float("nan"))
            cg.replay()
            torch.cuda.synchronize()
            for i in range(steps):
                assert torch.equal(gfs[i], want[i % nset][0]), f"replay {rep}, step {i}"


def test_default_policy_small_batches_single_launch_big_ones_two(gn, monkeypatch):
    monkeypatch.delenv("GNX_TICKET_MAX_ROWS")
    rng = np.random.default_rng(9)
    small = gn.GNGraphBatch.from_csc(*[[a] for a in U.er_csc(rng, 2000, 20000)], [2000])
    big = gn.GNGraphBatch.from_csc(*bench.make_c2())
    for g, want_graph_kernel in ((small, False), (big, True)):
        _, one, _, _, xt = _plan_pair(gn, g, ((10, 5, 0), (3, 4, 5)), 14)
        o = one.outputs()
        assert ("k_graph_t" in _kernels_of(gn, lambda: one(*xt, *o))) == want_graph_kernel


def test_uninitialised_workspace_is_only_used_without_the_flag(gn):
    """Without GNX_FLAG_WS_TICKETS a garbage-filled workspace is fine (two-launch form); gnx_block_workspace_init makes the same
    buffer eligible for the single-launch form."""
    import torch
    rng = np.random.default_rng(8)
    g = gn.GNGraphBatch.from_csc(*[[a] for a in U.er_csc(rng, 2000, 20000)], [2000])
    dims = ((10, 5, 0), (3, 4, 5))
    p, one, two, x, xt = _plan_pair(gn, g, dims, 13)
    junk = torch.full_like(one.ws, 0xAB)
    o = two.outputs()
    two(*xt, *o, ws=junk)
    ref, scale = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), *x, return_scale=True)
    U.assert_close(o[2].cpu().numpy(), ref[2], scale[2], "gf (two-launch, junk workspace)")
    lib = gn._lib.load()
    gn._lib.check(lib.gnx_block_workspace_init(g._h, C.byref(one.p), 1, junk.data_ptr(), junk.numel(), torch.cuda.current_stream().cuda_stream))
    o1 = one.outputs()
    one(*xt, *o1, ws=junk)
    torch.cuda.synchronize()
    assert torch.equal(o1[2], o[2])

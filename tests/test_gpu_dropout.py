"""Training-mode Dropout of a GNCore's FeedForwards (gnfeedforward.jl:27-31: Chain(Dense(d => 4d, relu), Dense(4d => d), Dropout(p)); Flux applies
the Dropout inside a gradient call and skips it in test mode): gnx_core_forward_train / gnx_core_backward_train / gnx_dropout_mask through the
mirror's GNCore(dims; dropout = p).

The reference's mask comes from Julia's random-number generator — nothing outside Julia reproduces those bits — so what is checked is the FORM:
  * the mask is independent per element, 0 with probability p and 1 / (1 - p) otherwise (Flux._dropout_kernel), differs per entity / per call;
  * the forward is x + block(gn1(x)) + mask .* ffwd(gn2(x)) in float64 with the library's own mask handed to the restatement;
  * the backward is torch float64 autograd of that same expression: the mask is regenerated, not stored;
  * outside a gradient call the layer is the identity (bit-identical to dropout = 0); testmode / trainmode force it as in Flux."""
import numpy as np
import pytest
import torch

from oracle import gn_oracle as O
from tests import util as U
from tests.test_gpu_backward import ACT, _graphs, _kink_free, _torch_block, _torch_ln

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


def _case(gn, dims, big, rng, pdrop, hidden_act, graphs=None):
    sizes, cps, rvs = graphs if graphs is not None else _graphs(rng, big)
    g = gn.GNGraphBatch.from_csc(cps, rvs, [int(n) for n in sizes])
    csc = (*g.csc(), g.node_off, g.edge_off)
    p = O.make_core_params(rng, dims)
    hidden_fn = {"tanh": torch.tanh, "relu": torch.relu, "gelu": ACT[4]}[hidden_act]
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims)
    # HIP: one differentiable call with Dropout(p) active
    core = U.core_from_params(gn, p)
    core.ffwd.dropout = pdrop
    for name, t in (("eff", "e"), ("nff", "n"), ("gff", "g")):
        fc1, fc2 = getattr(core.ffwd, name)
        setattr(core.ffwd, name, (gn.Dense.from_numpy(p[f"ff_{t}_W1"], p[f"ff_{t}_b1"], hidden_act, None), fc2))
    for q in core.parameters():
        q.requires_grad_(True)
    dev = g.device
    leaf = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    xt = [leaf(ef), leaf(nf), leaf(gf)]
    y = core(gn.NT(g, *(t.permute(2, 1, 0) for t in xt)))
    drop = core.last_dropout
    assert abs(drop.p - pdrop) < 1e-7
    masks = [gn.dropout_mask(drop, t, (dims[t], rows, 1), dev).permute(2, 1, 0)[0].double().cpu() for t, rows in enumerate((g.n_edges, g.n_nodes, g.n_graphs))]
    # float64 restatement with the call's masks
    T = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    W = {k: T(v) for k, v in p.items() if isinstance(v, np.ndarray)}
    Wb = {k: T(p["block"][k]) for k in ("We", "be", "Wn", "bn", "Wg", "bg")}
    xs = [T(ef[0]), T(nf[0]), T(gf[0])]
    l1 = [_torch_ln(x, W[f"ln1_{t}_gamma"], W[f"ln1_{t}_beta"], p["eps"], 0) for x, t in zip(xs, "eng")]
    l2 = [_torch_ln(x, W[f"ln2_{t}_gamma"], W[f"ln2_{t}_beta"], p["eps"], 0) for x, t in zip(xs, "eng")]
    blk = _torch_block(p["block"], csc, l1[0], l1[1], l1[2], Wb)
    outs_r, pre, scales = [], [], []
    for x, z, b, t, m in zip(xs, l2, blk, "eng", masks):
        zh = z @ W[f"ff_{t}_W1"].T + W[f"ff_{t}_b1"]
        pre.append(zh)
        f = hidden_fn(zh) @ W[f"ff_{t}_W2"].T + W[f"ff_{t}_b2"]
        outs_r.append(x + b + m * f)
        # error scale of an output element: the magnitudes that were added (the correction adds and subtracts f once more)
        scales.append((x.abs() + b.abs() + (1 + m) * (hidden_fn(zh).abs() @ W[f"ff_{t}_W2"].abs().T + W[f"ff_{t}_b2"].abs())).detach())
    if hidden_act == "relu" and not _kink_free(pre):
        return False
    # the bound every forward test uses, 1e-5 . S elementwise with the ORACLE's worst-case scale (|W| . |x| + |b| through every sum and through
    # LayerNorm's 1 / sigma: O.core_forward_sparse) — the FeedForward term's share counted (1 + 2 / (1 - p)) times: it enters scaled by the mask
    # (<= 1 / (1 - p)) and the correction y += (m - 1) . f adds and subtracts it once more.  (Until round 6 this compared against the summed output
    # magnitudes times a flat 8, which a graph-level sum over thousands of rows does not honour: 1.06e-4 on gf after an unrelated re-association.)
    _, scale_o = O.core_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    for name, got, ref, S, So in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), outs_r, scales, scale_o):
        err = (got.permute(2, 1, 0)[0].detach().double().cpu() - ref.detach()).abs().numpy()
        if err.size == 0:  # (the (de, 0) edge features of a batch without edges)
            continue
        bound = 1e-5 * (1.0 + 2.0 / (1.0 - pdrop)) * np.asarray(So[0], dtype=np.float64) + 1e-30
        assert float((err / bound).max()) <= 1.0, f"forward {name}: worst ratio to 1e-5 . S {float((err / bound).max()):.3f}"
        assert float(err.max()) <= 1e-5 * float(ref.detach().abs().max()), f"forward {name}: normwise {float(err.max()) / float(ref.detach().abs().max()):.3e}"
    cot = [torch.from_numpy(rng.standard_normal(tuple(o.shape))) for o in outs_r]
    sum((o * c).sum() for o, c in zip(outs_r, cot)).backward()
    loss = sum((o.permute(2, 1, 0)[0] * c.to(dev).float()).sum() for o, c in zip((y.ef, y.nf, y.gf), cot))
    loss.backward()

    def close(got, ref, what):
        ref = ref.detach().numpy(); got = got.detach().double().cpu().numpy()
        if ref.size == 0:
            assert got.shape == ref.shape, (what, got.shape, ref.shape)
            return
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.max(np.abs(got - ref)) <= 1e-3 * scale, f"{what}: max err {np.max(np.abs(got - ref)):.3e} (scale {scale:.3g})"

    for name, t, r in zip(("d_ef", "d_nf", "d_gf"), xt, xs):
        close(t.grad[0], r.grad, name)
    refs = [Wb[k].grad for k in ("We", "be", "Wn", "bn", "Wg", "bg")]
    for ln in ("ln1", "ln2"):
        for t in "eng":
            refs += [W[f"{ln}_{t}_gamma"].grad, W[f"{ln}_{t}_beta"].grad]
    for t in "eng":
        refs += [W[f"ff_{t}_W1"].grad, W[f"ff_{t}_b1"].grad, W[f"ff_{t}_W2"].grad, W[f"ff_{t}_b2"].grad]
    for i, (q, r) in enumerate(zip(core.parameters(), refs)):
        close(q.grad, r, f"param[{i}]")
    # a dropped element's FeedForward gradient really is zero: fc2's bias gradient = column sums of cot .* mask
    for t, (m, c) in enumerate(zip(masks, cot)):
        ref_b2 = (m * c).sum(dim=0)
        got_b2 = core.parameters()[18 + 4 * t + 3].grad.double().cpu()
        assert float((got_b2 - ref_b2).abs().max()) <= 1e-3 * max(1.0, float(ref_b2.abs().max()))
    return True


@pytest.mark.parametrize("pdrop", [0.1, 0.5])
@pytest.mark.parametrize("dims,big,act", [((3, 4, 5), False, "relu"), ((10, 5, 3), False, "relu"), ((40, 36, 33), True, "tanh"), ((64, 32, 16), True, "gelu"),
                                          ((128, 64, 32), True, "tanh")], ids=lambda v: str(v))
def test_core_training_forward_and_backward_with_dropout_match_float64(gn, dims, big, act, pdrop):
    for attempt in range(20):
        rng = np.random.default_rng(900 + sum(dims) + int(100 * pdrop) + 1000 * attempt)
        if _case(gn, dims, big, rng, pdrop, act):
            return
    pytest.fail("no kink-free draw in 20 attempts")


def test_mask_is_bernoulli_scaled_and_a_pure_function_of_seed_entity_and_element(gn):
    from graphnets_jl_amd import _lib
    dev = torch.device("cuda:0")
    n = (64, 50_000, 1)
    for p in (0.1, 0.5, 0.9):
        d = _lib.Dropout(p, 0, 12345)
        m = gn.dropout_mask(d, 0, n, dev).permute(2, 1, 0).contiguous().view(-1)
        keep = m != 0
        assert torch.all(m[keep] == float(np.float32(1.0) / (np.float32(1.0) - np.float32(p))))  # Flux._dropout_kernel: 1 / q
        N = m.numel()
        frac = float(keep.double().mean())
        assert abs(frac - (1 - p)) <= 5 * np.sqrt(p * (1 - p) / N), (p, frac)
        k = keep.double() - (1 - p)
        for lag in (1, 2, 3, 4, 64, 65):  # neighbouring elements (same Philox block, next block, next row) are uncorrelated
            c = float((k[:-lag] * k[lag:]).mean()) / (p * (1 - p))
            assert abs(c) <= 5 / np.sqrt(N), (p, lag, c)
        again = gn.dropout_mask(_lib.Dropout(p, 0, 12345), 0, n, dev).permute(2, 1, 0).contiguous().view(-1)
        assert torch.equal(m, again)
        for other in (gn.dropout_mask(_lib.Dropout(p, 0, 12346), 0, n, dev), gn.dropout_mask(d, 1, n, dev)):  # another call's seed / another entity
            o = other.permute(2, 1, 0).contiguous().view(-1) != 0
            agree = float((o == keep).double().mean())
            assert abs(agree - (p * p + (1 - p) * (1 - p))) <= 6 / np.sqrt(N)
    # a prefix of a longer mask is the shorter mask (element i depends on i only), odd lengths / unaligned ends included
    d = _lib.Dropout(0.3, 0, 7)
    full = gn.dropout_mask(d, 2, (1, 1003, 1), dev).view(-1)
    for k in (1, 2, 3, 5, 1001):
        assert torch.equal(gn.dropout_mask(d, 2, (1, k, 1), dev).view(-1), full[:k])
    # p = 1 drops everything (Flux: rand > 1 never holds)
    assert float(gn.dropout_mask(_lib.Dropout(1.0, 0, 7), 0, (4, 100, 1), dev).abs().max()) == 0.0
    with pytest.raises(gn.GnxError):
        gn.dropout_mask(_lib.Dropout(1.5, 0, 7), 0, (4, 100, 1), dev)


def test_dropout_follows_flux_modes(gn):
    """Automatic mode: identity outside a gradient call, active inside; testmode / trainmode force it (Flux.testmode! / trainmode!)."""
    rng = np.random.default_rng(77)
    dims = (10, 5, 3)
    sizes, cps, rvs = _graphs(rng, False)
    g = gn.GNGraphBatch.from_csc(cps, rvs, [int(n) for n in sizes])
    p = O.make_core_params(rng, dims)
    plain, core = U.core_from_params(gn, p), U.core_from_params(gn, p)
    core.ffwd.dropout = 0.5
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims)
    x = U.to_nt(gn, g, ef, nf, gf)
    same = lambda a, b: all(torch.equal(u, v) for u, v in zip((a.ef, a.nf, a.gf), (b.ef, b.nf, b.gf)))
    ref = plain(x)
    assert same(core(x), ref)                                   # no gradient: test mode
    for q in core.parameters():
        q.requires_grad_(True)
    with torch.no_grad():
        assert same(core(x), ref)
    torch.manual_seed(5)
    y1 = core(x)                                                # gradient call: active
    assert not same(y1, ref)
    torch.manual_seed(5)
    y2 = core(x)
    assert same(y1, y2)                                         # the seed comes from torch's generator: reproducible
    y3 = core(x)
    assert not same(y1, y3)                                     # a fresh mask per call
    gn.testmode(core)
    assert same(core(x), ref)                                   # forced off inside a gradient call
    gn.trainmode(core)
    with torch.no_grad():
        assert not same(core(x), ref)                           # forced on without one
    gn.testmode(core, None)                                     # back to automatic
    with torch.no_grad():
        assert same(core(x), ref)
    lst = gn.GNCoreList([core, U.core_from_params(gn, p)])
    lst.list[1].ffwd.dropout = 0.25
    gn.trainmode(lst)
    assert all(c._dropout_mode is True for c in lst.list)


def test_train_entry_points_without_dropout_are_the_plain_ones(gn):
    """dropout = NULL or p = 0: gnx_core_forward_train is gnx_core_forward bit for bit; argument checks."""
    import ctypes as C
    from graphnets_jl_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(78)
    dims = (128, 64, 32)
    sizes, cps, rvs = _graphs(rng, True)
    g = gn.GNGraphBatch.from_csc(cps, rvs, [int(n) for n in sizes])
    p = O.make_core_params(rng, dims)
    core = U.core_from_params(gn, p)
    ef, nf, gf = (torch.from_numpy(a).to(g.device) for a in U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims))
    ref = core(gn.NT(g, *(t.permute(2, 1, 0) for t in (ef, nf, gf))))
    keep = []
    cp = core._c(keep)
    nb = lib.gnx_core_train_workspace_bytes(g._h, C.byref(cp), 1)
    assert nb >= lib.gnx_core_workspace_bytes(g._h, C.byref(cp), 1) > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=g.device)
    stream = torch.cuda.current_stream(g.device).cuda_stream
    for drop in (None, _lib.Dropout(0.0, 0, 99)):
        outs = [torch.empty_like(t) for t in (ef, nf, gf)]
        gn._lib.check(lib.gnx_core_forward_train(g._h, C.byref(cp), None if drop is None else C.byref(drop), ef.data_ptr(), nf.data_ptr(), gf.data_ptr(), 1,
                                                 *(o.data_ptr() for o in outs), ws.data_ptr(), nb, 0, stream))
        for o, r in zip(outs, (ref.ef, ref.nf, ref.gf)):
            assert torch.equal(o, r.permute(2, 1, 0))
    bad = _lib.Dropout(-0.1, 0, 1)
    outs = [torch.empty_like(t) for t in (ef, nf, gf)]
    rc = lib.gnx_core_forward_train(g._h, C.byref(cp), C.byref(bad), ef.data_ptr(), nf.data_ptr(), gf.data_ptr(), 1, *(o.data_ptr() for o in outs), ws.data_ptr(), nb, 0, stream)
    assert rc == _lib.ERR_INVALID_ARG and b"Dropout" in lib.gnx_last_error()
    act = _lib.Dropout(0.5, 0, 1)
    rc = lib.gnx_core_forward_train(g._h, C.byref(cp), C.byref(act), ef.data_ptr(), nf.data_ptr(), gf.data_ptr(), 1, *(o.data_ptr() for o in outs), ws.data_ptr(),
                                    lib.gnx_core_workspace_bytes(g._h, C.byref(cp), 1), 0, stream)
    assert rc == _lib.ERR_WORKSPACE


def test_sort_example_trains_with_dropout(gn):
    """examples/train_sort.py --dropout 0.1: two GNCores whose FeedForwards drop 10 % in every training call (a fresh mask per core and call)
    still learn; evaluation calls (no gradient) run without the mask."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_sort.py")
    spec = importlib.util.spec_from_file_location("train_sort_dropout", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ["train_sort.py", "--iters", "150", "--graphs", "32", "--n", "6", "--width", "8", "--dropout", "0.1"]
    try:
        hist = mod.main()
    finally:
        sys.argv = argv
    assert np.isfinite(hist).all()
    assert np.mean(hist[-10:]) < 0.8 * np.mean(hist[:5]), (hist[:5], hist[-10:])


def test_dropout_on_a_shared_graph_batch_masks_every_replica_independently(gn):
    """A shared-adjacency batch (src/batch.jl:66: R = the data batch size) — the mask covers the packed [R][rows][width] tensor, so the replicas of one
    graph draw different masks; y_train - y_test = (m - 1) .* ffwd(gn2(x)) per replica, in float64 from the library's mask."""
    rng = np.random.default_rng(91)
    dims, R = (10, 5, 3), 3
    adj = (rng.random((30, 30)) < 0.2).astype(np.int64)
    p = O.make_core_params(rng, dims)
    core = U.core_from_params(gn, p)
    core.ffwd.dropout = 0.4
    E, N = int(adj.sum()), 30
    ef, nf, gf = rng.random((dims[0], E, R), dtype=np.float32), rng.random((dims[1], N, R), dtype=np.float32), rng.random((dims[2], R), dtype=np.float32)
    x = gn.batch(dict(graphs=adj, ef=ef, nf=nf, gf=gf))
    y0 = core(x)
    gn.trainmode(core)
    y1 = core(x)
    drop = core.last_dropout
    for t, (name, a) in enumerate((("ef", x.ef), ("nf", x.nf), ("gf", x.gf))):
        m = gn.dropout_mask(drop, t, tuple(a.shape), a.device).double().cpu()          # (d, rows, R)
        assert t == 2 or not torch.equal(m[:, :, 0], m[:, :, 1])              # (three elements per replica of gf can coincide)
        z = a.double().cpu().permute(2, 1, 0)                                           # [R][rows][d]
        k = "eng"[t]
        T = lambda v: torch.from_numpy(np.asarray(v, dtype=np.float64))
        l2 = _torch_ln(z.reshape(-1, z.shape[2]), T(p[f"ln2_{k}_gamma"]), T(p[f"ln2_{k}_beta"]), p["eps"], 0)
        f = torch.relu(l2 @ T(p[f"ff_{k}_W1"]).T + T(p[f"ff_{k}_b1"])) @ T(p[f"ff_{k}_W2"]).T + T(p[f"ff_{k}_b2"])
        f = f.reshape(z.shape).permute(2, 1, 0)
        got = (getattr(y1, name).double() - getattr(y0, name).double()).cpu()
        want = (m - 1) * f
        scale = float(getattr(y0, name).abs().max()) + float(f.abs().max())
        assert float((got - want).abs().max()) <= 5e-6 * scale, (name, float((got - want).abs().max()), scale)

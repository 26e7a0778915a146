"""Backward pass of the GNBlock: gnx_block_backward through the mirror's torch.autograd.Function against
torch CPU float64 autograd of a plain restatement of the forward (index_select / index_add)."""
import numpy as np
import pytest
import torch

from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
ACT = {0: lambda x: x, 1: torch.relu, 2: torch.tanh, 3: torch.sigmoid, 4: lambda x: torch.nn.functional.gelu(x, approximate="tanh")}  # 4: NNlib.gelu (tanh form)


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


def _torch_block(p, csc, ef, nf, gf, W, pre=None):
    """float64 torch restatement of SURVEY Appendix A (one replica); W = dict of leaf tensors.  `pre` (a list) receives the
    three pre-activations."""
    colptr, rowval, node_off, edge_off = (torch.from_numpy(np.asarray(a)) for a in csc)
    N, G = len(colptr) - 1, len(node_off) - 1
    dst = torch.repeat_interleave(torch.arange(N), colptr[1:] - colptr[:-1])
    ng = torch.repeat_interleave(torch.arange(G), node_off[1:] - node_off[:-1])
    eg = torch.repeat_interleave(torch.arange(G), edge_off[1:] - edge_off[:-1])
    cat = lambda parts: torch.cat([q for q in parts if q is not None], dim=1)
    Xe = cat([ef, None if nf is None else nf[rowval], None if nf is None else nf[dst], None if gf is None else gf[eg]])
    ze = Xe @ W["We"].T + W["be"]
    he = ACT[p["act_e"]](ze)
    agg = torch.zeros((N, he.shape[1]), dtype=torch.float64).index_add(0, dst, he)
    zn = cat([agg, nf, None if gf is None else gf[ng]]) @ W["Wn"].T + W["bn"]
    hn = ACT[p["act_n"]](zn)
    se = torch.zeros((G, he.shape[1]), dtype=torch.float64).index_add(0, eg, he)
    sn = torch.zeros((G, hn.shape[1]), dtype=torch.float64).index_add(0, ng, hn)
    zg = cat([se, sn, gf]) @ W["Wg"].T + W["bg"]
    hg = ACT[p["act_g"]](zg)
    if pre is not None:
        pre.extend([ze, zn, zg])
    return he, hn, hg


DIMS = [((10, 5, 0), (3, 4, 5)), ((3, 2, 4), (3, 4, 5)), ((0, 2, 0), (2, 2, 2)), ((4, 0, 3), (2, 3, 2)), ((6, 5, 0), (4, 3, 0)),
        ((40, 24, 8), (36, 20, 12))]


def _graphs(rng, big, many=False):
    """small: 6 graphs of 5..40 nodes (generic kernels); big: 3 graphs of 1500..2200 nodes, 4 edges per node — enough rows
    (>= 4096 nodes and edges) for the matrix-core dX / dW kernels of the backward to be selected."""
    if many:  # more graphs than a grid's y / z extent (65535): 70k graphs of 2 or 3 nodes
        sizes = rng.integers(2, 4, 70_000)
        cps, rvs = [], []
        for n in sizes:
            k = np.sort(rng.choice(int(n) * int(n), 2, replace=False))
            cp = np.zeros(int(n) + 1, dtype=np.int64)
            np.add.at(cp, k // int(n) + 1, 1)
            cps.append(np.cumsum(cp)); rvs.append((k % int(n)).astype(np.int64))
        return sizes, cps, rvs
    sizes = rng.integers(1500, 2200, 3) if big else rng.integers(5, 40, 6)
    cps, rvs = [], []
    for n in sizes:
        cp, rv = U.er_csc(rng, int(n), 4 * int(n) if big else int(0.15 * n * n) + 1)
        cps.append(cp); rvs.append(rv)
    return sizes, cps, rvs


BIG_DIMS = [((40, 24, 8), (36, 20, 12)), ((37, 22, 5), (35, 19, 7)), ((128, 64, 32), (128, 64, 32))]


def _kink_free(pre_acts, margin=5e-6):
    """relu is not differentiable at 0: an fp32 pre-activation within rounding distance of 0 can land on the other side
    of the kink than the float64 reference and legitimately flip one derivative from 0 to 1.  The comparisons below are
    only meaningful on data without such elements (small cases: re-draw; big cases use smooth activations)."""
    return all(float(t.detach().abs().min()) > margin for t in pre_acts if t.numel())


@pytest.mark.parametrize("dims", DIMS + BIG_DIMS, ids=[str(d) for d in DIMS] + ["mfma-" + str(d) for d in BIG_DIMS])
@pytest.mark.parametrize("act", [(0, 0, 0), (1, 2, 3), (4, 4, 4)], ids=["identity", "relu-tanh-sigmoid", "gelu"])
def test_block_backward_matches_torch_autograd(gn, dims, act, request):
    big = "mfma-" in request.node.callspec.id
    if big and act[0] == 1:
        act = (2, 2, 3)  # ~800k edge outputs: some pre-activation always sits on the relu kink; tanh keeps the check meaningful
    for attempt in range(20):
        rng = np.random.default_rng(200 + sum(dims[0]) + sum(act) + 1000 * attempt)
        if _block_backward_case(gn, dims, act, big, rng):
            return
    pytest.fail("no kink-free draw in 20 attempts")


def _block_backward_case(gn, dims, act, big, rng, many=False):
    sizes, cps, rvs = _graphs(rng, big, many)
    g = gn.GNGraphBatch.from_csc(cps, rvs, [int(n) for n in sizes])
    csc = (*g.csc(), g.node_off, g.edge_off)
    p = O.make_block_params(rng, *dims, act=act)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    # --- reference: torch float64 autograd
    W = {k: torch.tensor(p[k], dtype=torch.float64, requires_grad=True) for k in ("We", "be", "Wn", "bn", "Wg", "bg")}
    t64 = lambda a: None if a is None else torch.tensor(a[0], dtype=torch.float64, requires_grad=True)
    ef_r, nf_r, gf_r = t64(ef), t64(nf), t64(gf)
    pre = []
    outs_r = _torch_block(p, csc, ef_r, nf_r, gf_r, W, pre)
    if not _kink_free([z for z, a in zip(pre, act) if a == 1]):
        return False
    cot = [torch.from_numpy(rng.standard_normal(tuple(o.shape))) for o in outs_r]  # random cotangents
    loss_r = sum((o * c).sum() for o, c in zip(outs_r, cot) if o.shape[1] > 0)
    loss_r.backward()
    # --- HIP: forward + backward through the mirror's autograd function
    blk = U.block_from_params(gn, p)
    for layer in (blk.edgefn, blk.nodefn, blk.graphfn):
        layer.weight.requires_grad_(True)
        layer.bias.requires_grad_(True)
    dev = g.device
    leaf = lambda a: None if a is None else torch.from_numpy(a).to(dev).requires_grad_(True)
    ef_t, nf_t, gf_t = leaf(ef), leaf(nf), leaf(gf)
    jl = lambda t: None if t is None else t.permute(2, 1, 0)
    y = blk(gn.NT(g, jl(ef_t), jl(nf_t), jl(gf_t)))
    outs = [y.ef, y.nf, y.gf]
    loss = 0.0
    for o, c in zip(outs, cot):
        if o is not None:
            loss = loss + (o.permute(2, 1, 0)[0] * c.to(dev).float()).sum()
    loss.backward()

    def close(got, ref, what):
        ref = ref.detach().numpy()
        got = got.detach().double().cpu().numpy()
        if ref.size == 0:  # (the gradient of a (0, d) array: a batch without edges)
            assert got.shape == ref.shape, (what, got.shape, ref.shape)
            return
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.max(np.abs(got - ref)) <= 2e-4 * scale, f"{what}: max err {np.max(np.abs(got - ref)):.3e} (scale {scale:.3g})"

    for name, t, r in (("d_ef", ef_t, ef_r), ("d_nf", nf_t, nf_r), ("d_gf", gf_t, gf_r)):
        if t is not None:
            close(t.grad[0], r.grad, name)
    for name, layer, kw, kb in (("edge", blk.edgefn, "We", "be"), ("node", blk.nodefn, "Wn", "bn"), ("graph", blk.graphfn, "Wg", "bg")):
        if layer.weight.numel():
            close(layer.weight.grad, W[kw].grad, f"dW_{name}")
            close(layer.bias.grad, W[kb].grad, f"db_{name}")
    return True


def test_block_forward_backward_with_more_than_65535_graphs(gn):
    """70 000 tiny graphs: the graph index must not sit in a grid dimension that stops at 65535 (forward and backward)."""
    assert _block_backward_case(gn, ((3, 2, 4), (3, 4, 5)), (0, 0, 0), False, np.random.default_rng(777), many=True)


def test_backward_is_deterministic_and_trains(gn):
    """Two backward passes give identical bits, and a few SGD steps through the HIP forward/backward reduce a loss."""
    rng = np.random.default_rng(300)
    colptr, rowval = U.er_csc(rng, 300, 2500)
    g = gn.GNGraphBatch.from_csc([colptr], [rowval], [300])
    blk = gn.GNBlock((10, 5, 0), (3, 4, 5))
    params = []
    for layer in (blk.edgefn, blk.nodefn, blk.graphfn):
        layer.weight.requires_grad_(True); layer.bias.requires_grad_(True)
        params += [layer.weight, layer.bias]
    ef, nf, _ = U.packed_inputs(rng, 1, 2500, 300, 1, (10, 5, 0))
    x = U.to_nt(gn, g, ef, nf, None)
    target = torch.from_numpy(rng.random((4, 300), dtype=np.float32)).to(g.device)

    def loss_fn():
        y = blk(x)
        return ((y.nf[:, :, 0] - target) ** 2).mean() + 1e-6 * (y.gf ** 2).mean() + 1e-3 * (y.ef ** 2).mean()

    grads = []
    for _ in range(2):
        for q in params:
            q.grad = None
        loss_fn().backward()
        grads.append([q.grad.clone() for q in params])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    first = float(loss_fn().detach())
    opt = torch.optim.Adam(params, lr=1e-2)
    for _ in range(60):
        opt.zero_grad()
        loss = loss_fn()
        loss.backward()
        opt.step()
    last = float(loss_fn().detach())
    assert np.isfinite(last) and last < 0.7 * first, (first, last)


def _torch_ln(x, gamma, beta, eps, eps_mode):
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    xh = (x - mu) / (var.sqrt() + eps) if eps_mode == 0 else (x - mu) / (var + eps).sqrt()
    return xh * gamma + beta


@pytest.mark.parametrize("eps_mode", [0, 1])
@pytest.mark.parametrize("dims,big", [((3, 4, 5), False), ((10, 5, 3), False), ((40, 36, 33), False), ((40, 36, 33), True), ((64, 32, 16), True)],
                         ids=lambda v: str(v))
def test_core_backward_matches_torch_autograd(gn, dims, big, eps_mode):
    """gnx_core_backward (LayerNorm + FeedForward + block pullbacks, residual) against torch float64 autograd.  Small cases
    use the reference's relu FeedForward on kink-free draws; the big (matrix-core) cases use tanh as the hidden activation."""
    for attempt in range(20):
        rng = np.random.default_rng(400 + sum(dims) + eps_mode + 1000 * attempt)
        if _core_backward_case(gn, dims, big, eps_mode, rng):
            return
    pytest.fail("no kink-free draw in 20 attempts")


@pytest.mark.parametrize("dims,big", [((10, 5, 3), False), ((40, 36, 33), True), ((64, 32, 16), True)], ids=lambda v: str(v))
def test_core_backward_with_a_gelu_feedforward(gn, dims, big):
    """A FeedForward whose hidden activation is gelu (not a function of its output: the pullback keeps the recomputed
    pre-activation until delta1 is formed) — generic and matrix-core forms against torch float64 autograd."""
    assert _core_backward_case(gn, dims, big, 0, np.random.default_rng(450 + sum(dims)), hidden_act="gelu")


def _core_backward_case(gn, dims, big, eps_mode, rng, hidden_act=None, graphs=None):
    sizes, cps, rvs = graphs if graphs is not None else _graphs(rng, big)
    g = gn.GNGraphBatch.from_csc(cps, rvs, [int(n) for n in sizes])
    csc = (*g.csc(), g.node_off, g.edge_off)
    p = O.make_core_params(rng, dims, eps_mode=eps_mode)
    hidden_act = hidden_act or ("tanh" if big else "relu")
    hidden_fn = {"tanh": torch.tanh, "relu": torch.relu, "gelu": ACT[4]}[hidden_act]
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims)
    # reference
    T = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    W = {k: T(v) for k, v in p.items() if isinstance(v, np.ndarray)}
    Wb = {k: T(p["block"][k]) for k in ("We", "be", "Wn", "bn", "Wg", "bg")}
    xs = [T(ef[0]), T(nf[0]), T(gf[0])]
    l1 = [_torch_ln(x, W[f"ln1_{t}_gamma"], W[f"ln1_{t}_beta"], p["eps"], eps_mode) for x, t in zip(xs, "eng")]
    l2 = [_torch_ln(x, W[f"ln2_{t}_gamma"], W[f"ln2_{t}_beta"], p["eps"], eps_mode) for x, t in zip(xs, "eng")]
    blk = _torch_block(p["block"], csc, l1[0], l1[1], l1[2], Wb)
    outs_r, pre = [], []
    for x, z, b, t in zip(xs, l2, blk, "eng"):
        zh = z @ W[f"ff_{t}_W1"].T + W[f"ff_{t}_b1"]
        pre.append(zh)
        hdn = hidden_fn(zh)
        outs_r.append(x + b + hdn @ W[f"ff_{t}_W2"].T + W[f"ff_{t}_b2"])
    if hidden_act == "relu" and not _kink_free(pre):
        return False
    cot = [torch.from_numpy(rng.standard_normal(tuple(o.shape))) for o in outs_r]
    sum((o * c).sum() for o, c in zip(outs_r, cot)).backward()
    # HIP
    core = U.core_from_params(gn, p)
    for name, t in (("eff", "e"), ("nff", "n"), ("gff", "g")):
        fc1, fc2 = getattr(core.ffwd, name)
        setattr(core.ffwd, name, (gn.Dense.from_numpy(p[f"ff_{t}_W1"], p[f"ff_{t}_b1"], hidden_act, None), fc2))
    for q in core.parameters():
        q.requires_grad_(True)
    dev = g.device
    leaf = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    xt = [leaf(ef), leaf(nf), leaf(gf)]
    y = core(gn.NT(g, *(t.permute(2, 1, 0) for t in xt)))
    loss = sum((o.permute(2, 1, 0)[0] * c.to(dev).float()).sum() for o, c in zip((y.ef, y.nf, y.gf), cot))
    loss.backward()

    def close(got, ref, what):
        ref = ref.detach().numpy(); got = got.detach().double().cpu().numpy()
        if ref.size == 0:  # (the gradient of a (0, d) array: a batch without edges)
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
    return True


def test_readout_loss_is_differentiable(gn):
    rng = np.random.default_rng(500)
    yhat = torch.from_numpy(rng.standard_normal((5, 37)).astype(np.float32)).cuda().requires_grad_(True)
    tgt = torch.from_numpy(np.eye(5, dtype=np.float32)[rng.integers(0, 5, 37)].T.copy()).cuda()
    gn.logitcrossentropy(yhat, tgt).backward()
    ref = yhat.detach().double().cpu().requires_grad_(True)
    torch.nn.functional.cross_entropy(ref.t(), tgt.cpu().argmax(0)).backward()
    assert torch.allclose(yhat.grad.double().cpu(), ref.grad, rtol=1e-4, atol=1e-6)


def test_end_to_end_training_sort_example(gn):
    """examples/train_sort.py (analogue of the reference's examples/sort): the loss must go down when a
    GNBlock -> 2 x GNCore -> GNBlock model is trained with AdamW through the HIP forward and backward kernels."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_sort.py")
    spec = importlib.util.spec_from_file_location("train_sort", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ["train_sort.py", "--iters", "120", "--graphs", "32", "--n", "6", "--width", "8"]
    try:
        hist = mod.main()
    finally:
        sys.argv = argv
    assert np.isfinite(hist).all()
    assert np.mean(hist[-10:]) < 0.75 * np.mean(hist[:5]), (hist[:5], hist[-10:])

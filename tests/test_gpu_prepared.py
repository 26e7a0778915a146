"""Prepared parameters (include/gnx.h: gnx_block_prepare / gnx_core_prepare / gnx_prepared_refresh, gnx_model_refresh_weights): a layer's weight
blocks in the forms the six-term matrix-core kernels stage are made ONCE — the reference moves a model to the device once and then calls it
(`model |> device`, examples/sort/sort.jl:29,89) — instead of by preparation launches in front of every forward.  Bit-identical outputs, no
`*_prep` kernel in a prepared forward, in-place weight updates picked up through the refresh."""
import numpy as np
import pytest

from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn_
    # (these tests name the kernels of the DEFAULT forms: which preparation launches exist, that the one-launch core kernel runs)
    U.needs_default_forms(gn_, *U.ONE_LAUNCH_CORE_FORMS, "PROJ_FP32", "EDGE_NARROW_FP32")
    return gn_


def _profiled(gn, f):
    gn.profile_reset(); gn.profile_enable(True)
    y = f()
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    return y, names


def _prep_kernels(names):
    return {n for n in names if n.endswith("_prep")}


def _batch(gn, rng, n=4300, e=9500):
    colptr, rowval = U.er_csc(rng, n, e)
    return gn.GNGraphBatch.from_csc([colptr], [rowval], [n])


def test_prepared_core_launches_no_preparation_kernel_and_is_bit_identical(gn):
    import torch
    rng = np.random.default_rng(9100)
    dims = (128, 64, 32)
    g = _batch(gn, rng)
    p = O.make_core_params(rng, dims)
    core = U.core_from_params(gn, p)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, dims)
    x = U.to_nt(gn, g, ef, nf, gf)
    y0, names0 = _profiled(gn, lambda: core(x))
    assert {"k_edge_x6_prep", "k_ffn_x6_prep", "k_proj_x6_prep"} <= names0, names0  # the per-call preparation of the six-term kernels
    core.prepare()
    assert core._prepared.nbytes() > 900_000  # edge block 96 KB + projections 96 KB + two FeedForwards' planes
    y1, names1 = _profiled(gn, lambda: core(x))
    assert not _prep_kernels(names1), names1
    assert names1 == names0 - _prep_kernels(names0)  # the same kernels otherwise
    for a, b in zip((y0.ef, y0.nf, y0.gf), (y1.ef, y1.nf, y1.gf)):
        assert torch.equal(a, b)
    ref, scale = O.core_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y1.ef, y1.nf, y1.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name)
    # the forms of the call still apply: the fp32 matrix instruction needs no planes and must not be handed any
    y2 = core(x, flags=gn._lib.FLAG_FP32_MFMA)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y2.ef, y2.nf, y2.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, "fp32 " + name)


def test_in_place_weight_update_is_picked_up_through_the_refresh(gn):
    """an optimiser step writes the weights in place: the mirror sees the tensors' version counters move and refreshes the planes before the next
    forward (gnx_prepared_refresh, stream-ordered); the result equals a fresh, unprepared layer with the updated weights — bit for bit"""
    import torch
    rng = np.random.default_rng(9200)
    dims = (128, 64, 32)
    g = _batch(gn, rng)
    p = O.make_core_params(rng, dims)
    core = U.core_from_params(gn, p).prepare()
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, dims)
    x = U.to_nt(gn, g, ef, nf, gf)
    ya = core(x)
    with torch.no_grad():
        core.block.edgefn.weight.mul_(1.25)
        core.ffwd.eff[0].weight.add_(0.01)
        core.ffwd.nff[1].weight.mul_(0.5)
    yb, names = _profiled(gn, lambda: core(x))  # (the refresh runs the preparation kernels once — inside this call)
    assert not torch.equal(ya.ef, yb.ef)
    yc, names_c = _profiled(gn, lambda: core(x))
    assert not _prep_kernels(names_c), names_c
    p2 = dict(p)
    p2["block"] = dict(p["block"], We=p["block"]["We"] * np.float32(1.25))
    p2["ff_e_W1"] = p["ff_e_W1"] + np.float32(0.01)
    p2["ff_n_W2"] = p["ff_n_W2"] * np.float32(0.5)
    fresh = U.core_from_params(gn, p2)(x)
    for a, b, c in zip((yb.ef, yb.nf, yb.gf), (yc.ef, yc.nf, yc.gf), (fresh.ef, fresh.nf, fresh.gf)):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_in_place_update_of_the_folded_layernorm_parameters_and_bias_is_picked_up(gn):
    """the one-launch form of a core's edge rows folds gn1 / gn2 of the edges (scale into the weight planes, shift into constant vectors) and fc1's
    bias into its prepared planes: an in-place update of any of them refreshes the planes like a weight update does"""
    import torch
    rng = np.random.default_rng(9250)
    dims = (128, 64, 32)
    g = _batch(gn, rng)
    p = O.make_core_params(rng, dims)
    core = U.core_from_params(gn, p).prepare()
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, dims)
    x = U.to_nt(gn, g, ef, nf, gf)
    ya = core(x)
    with torch.no_grad():
        core.gn1.edgeln.gamma.mul_(1.5)
        core.gn1.edgeln.beta.add_(0.25)
        core.gn2.edgeln.gamma.mul_(0.75)
        core.gn2.edgeln.beta.sub_(0.125)
        core.ffwd.eff[0].bias.add_(0.5)
    yb, _ = _profiled(gn, lambda: core(x))
    assert not torch.equal(ya.ef, yb.ef)
    p2 = dict(p)
    p2["ln1_e_gamma"] = p["ln1_e_gamma"] * np.float32(1.5)
    p2["ln1_e_beta"] = p["ln1_e_beta"] + np.float32(0.25)
    p2["ln2_e_gamma"] = p["ln2_e_gamma"] * np.float32(0.75)
    p2["ln2_e_beta"] = p["ln2_e_beta"] - np.float32(0.125)
    p2["ff_e_b1"] = p["ff_e_b1"] + np.float32(0.5)
    fresh = U.core_from_params(gn, p2)(x)
    for a, b in zip((yb.ef, yb.nf, yb.gf), (fresh.ef, fresh.nf, fresh.gf)):
        assert torch.equal(a, b)
    ref, scale = O.core_forward_sparse(p2, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
    for name, got, r_, s_ in zip(("ef", "nf", "gf"), (yb.ef, yb.nf, yb.gf), ref, scale):
        U.assert_close(U.from_jl(got), r_, s_, name)


def test_in_place_update_of_the_node_weights_alone_is_picked_up(gn):
    """k_node_x6 stages planes made from nodefn.weight (4300 nodes: the six-term node update runs); a change of THAT tensor alone — frozen
    edge weights, a partial load_state_dict — must refresh them, for a block and for a core, outside and inside a gradient call (whose
    autograd.Function.forward runs with grad mode off: the mirror's version list is what the planes' freshness rests on)."""
    import torch
    rng = np.random.default_rng(9300)
    dims = (128, 64, 32)
    g = _batch(gn, rng)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, dims)
    x = U.to_nt(gn, g, ef, nf, gf)
    # a block
    pb = O.make_block_params(rng, dims, dims)
    blk = U.block_from_params(gn, pb).prepare()
    ya = blk(x)
    with torch.no_grad():
        blk.nodefn.weight.mul_(1.5)
        blk.graphfn.bias.add_(0.25)
    yb = blk(x)
    assert not torch.equal(ya.nf, yb.nf) and torch.equal(ya.ef, yb.ef)
    fresh = U.block_from_params(gn, dict(pb, Wn=pb["Wn"] * np.float32(1.5), bg=pb["bg"] + np.float32(0.25)))(x)
    for a, b in zip((yb.ef, yb.nf, yb.gf), (fresh.ef, fresh.nf, fresh.gf)):
        assert torch.equal(a, b)
    # a core, the second forward inside a gradient call
    p = O.make_core_params(rng, dims)
    core = U.core_from_params(gn, p).prepare()
    ya = core(x)
    with torch.no_grad():
        core.block.nodefn.weight.mul_(0.5)
        core.gn1.nodeln.gamma.mul_(1.25)
    for t in core.parameters():
        t.requires_grad_(True)
    yb = core(x)
    assert yb.nf.requires_grad and not torch.equal(ya.nf, yb.nf.detach())
    p2 = dict(p, ln1_n_gamma=p["ln1_n_gamma"] * np.float32(1.25))
    p2["block"] = dict(p["block"], Wn=p["block"]["Wn"] * np.float32(0.5))
    fresh = U.core_from_params(gn, p2)(x)
    for a, b in zip((yb.ef, yb.nf, yb.gf), (fresh.ef, fresh.nf, fresh.gf)):
        assert torch.equal(a.detach(), b)


@pytest.mark.parametrize("dout", [(128, 64, 32), (3, 4, 5)])
def test_prepared_block(gn, dout):
    """GNBlock (128,64,32) => (128,64,32) (edge block + both projection blocks) and => (3,4,5) (config 4's decoder: the narrow edge form)"""
    import torch
    rng = np.random.default_rng(9300 + dout[0])
    din = (128, 64, 32)
    g = _batch(gn, rng)
    p = O.make_block_params(rng, din, dout)
    blk = U.block_from_params(gn, p)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, din)
    x = U.to_nt(gn, g, ef, nf, gf)
    y0, names0 = _profiled(gn, lambda: blk(x))
    assert "k_edge_x6_prep" in names0 and (("k_proj_x6_prep" in names0) == (dout[0] == 128)), names0  # (the narrow form's projections are k_rows_gemm launches: no planes)
    blk.prepare()
    y1, names1 = _profiled(gn, lambda: blk(x))
    assert not _prep_kernels(names1), names1
    for a, b in zip((y0.ef, y0.nf, y0.gf), (y1.ef, y1.nf, y1.gf)):
        assert torch.equal(a, b)
    # narrow widths have nothing to prepare: an empty object, the forward unchanged
    pn = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    bn = U.block_from_params(gn, pn).prepare()
    assert bn._prepared.nbytes() == 0
    efn, nfn, _ = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, (10, 5, 0))
    yn = bn(U.to_nt(gn, g, efn, nfn, None))
    ref, scale = O.block_forward_sparse(pn, (*g.csc(), g.node_off, g.edge_off), efn, nfn, None, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (yn.ef, yn.nf, yn.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name)


def test_gnx_model_prepares_its_layers_once_and_refreshes_on_request(gn):
    """gnx_model_create prepares every layer; a forward of encoder -> core -> decoder at matrix-core widths launches no preparation kernel;
    after an in-place update of a weight the model keeps computing with the OLD planes until gnx_model_refresh_weights — then it equals the
    eager layers (which prepare per call)"""
    import torch
    rng = np.random.default_rng(9400)
    core_dims = (128, 64, 32)
    g = _batch(gn, rng)
    enc = U.block_from_params(gn, O.make_block_params(rng, (10, 5, 0), core_dims))
    core = U.core_from_params(gn, O.make_core_params(rng, core_dims))
    dec = U.block_from_params(gn, O.make_block_params(rng, core_dims, (3, 4, 5)))
    layers = [enc, core, dec]
    ef, nf, _ = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, 1, (10, 5, 0))
    x = U.to_nt(gn, g, ef, nf, None)

    def eager():
        y = x
        for l in layers:
            y = l(y)
        return y
    m = gn.Model(layers, x, flags=gn._lib.FLAG_NO_GRAPH)  # (eager launches inside the model: the profile sees every kernel)
    y_m, names = _profiled(gn, lambda: m(x))
    assert not _prep_kernels(names) and "k_core_edge_x6" in names, names
    y_e = eager()
    for a, b in zip((y_m.ef, y_m.nf, y_m.gf), (y_e.ef, y_e.nf, y_e.gf)):
        assert torch.equal(a, b)
    with torch.no_grad():
        core.ffwd.eff[1].weight.mul_(1.5)
    y_stale = m(x)
    y_new = eager()
    assert not torch.equal(y_stale.ef, y_new.ef)  # (the contract: the model's planes are a snapshot until it is told)
    m.refresh_weights()
    y_ref = m(x)
    for a, b in zip((y_ref.ef, y_ref.nf, y_ref.gf), (y_new.ef, y_new.nf, y_new.gf)):
        assert torch.equal(a, b)

"""The oracle against the reference's own fixtures and relational tests (CPU only).

Mirrors /root/reference/test/runtests.jl: known-answer broadcasters (:487-508, :659-680), batch invariance
(:62-116), batch/unbatch identity (:328-390), output shapes and `nothing` handling (:118-326, :627-735)."""
import itertools
import json
import os

import numpy as np
import pytest

from oracle import gn_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
README_ADJ = np.array([[1, 0, 1], [1, 1, 0], [0, 0, 1]])
README_ADJ2 = np.array([[1, 0, 1, 0], [1, 1, 0, 1], [0, 0, 1, 0], [1, 1, 0, 1]])


def _known():
    with open(os.path.join(HERE, "golden", "reference_known_answers.json")) as f:
        return json.load(f)


def test_known_answer_node2edge_broadcasters():
    k = _known()
    adj = [np.array(a) for a in k["adj_mats"]]
    padded = O.padadjmats(adj)
    src = O.getnode2edgebroadcaster(padded)
    dst = O.getnode2edgebroadcaster(padded, transpose=True)
    for b in range(2):
        assert np.array_equal(src[:, :, b], np.array(k["src_broadcaster"][b]))
        assert np.array_equal(dst[:, :, b], np.array(k["dst_broadcaster"][b]))


def test_known_answer_edge2node_broadcaster():
    k = _known()
    adj = [np.array(a) for a in k["adj_mats"]]
    e2n = O.getedge2nodebroadcaster(O.padadjmats(adj))
    for b in range(2):
        assert np.array_equal(e2n[:, :, b], np.array(k["edge2node_broadcaster"][b]))


def test_csc_matches_known_answer_broadcasters():
    """The sparse encoding (what the HIP path consumes) reproduces the reference's one-hot matrices: for the
    k-th active slot of graph b, rowval = the hot row of src_broadcaster and the colptr segment = the hot row
    of dst_broadcaster."""
    k = _known()
    for b, a in enumerate(k["adj_mats"]):
        a = np.array(a)
        colptr, rowval, node_off, edge_off = O.csc_from_adj([a])
        src_m, dst_m = np.array(k["src_broadcaster"][b]), np.array(k["dst_broadcaster"][b])
        slots = np.nonzero(src_m.sum(axis=0))[0]  # active slots in column-major order
        assert len(slots) == len(rowval) == edge_off[-1]
        dst = np.repeat(np.arange(a.shape[0]), np.diff(colptr))
        assert np.array_equal(src_m[:, slots].argmax(axis=0), rowval)
        assert np.array_equal(dst_m[:, slots].argmax(axis=0), dst)
        e2n = np.array(k["edge2node_broadcaster"][b])
        assert np.array_equal(e2n[slots].argmax(axis=1), dst)


def _rand_adj(rng, n, p=0.5):
    a = (rng.random((n, n)) < p).astype(np.int64)
    return a


def _rand_inputs_vector(rng, adjs, dims):
    de, dn, dg = dims
    ef = [rng.random((de, int(a.sum())), dtype=np.float32) for a in adjs] if de else None
    nf = [rng.random((dn, a.shape[0]), dtype=np.float32) for a in adjs] if dn else None
    gf = [rng.random((dg,), dtype=np.float32) for a in adjs] if dg else None
    return ef, nf, gf


def _rand_inputs_shared(rng, adj, dims, B):
    de, dn, dg = dims
    ef = rng.random((de, int(adj.sum()), B), dtype=np.float32) if de else None
    nf = rng.random((dn, adj.shape[0], B), dtype=np.float32) if dn else None
    gf = rng.random((dg, B), dtype=np.float32) if dg else None
    return ef, nf, gf


IN_COMBOS = [d for d in itertools.product((0, 3), (0, 2), (0, 4)) if any(d)]  # 7 edge-input forms


@pytest.mark.parametrize("in_dims", IN_COMBOS)
@pytest.mark.parametrize("out_dims", [(3, 4, 5), (2, 3, 0)])
def test_dense_equals_sparse_vector_mode(in_dims, out_dims):
    rng = np.random.default_rng(hash((in_dims, out_dims)) % 2**32)
    adjs = [_rand_adj(rng, n) for n in (3, 5, 1, 4)]
    adjs[2][:] = 1  # single self-loop graph
    p = O.make_block_params(rng, in_dims, out_dims)
    ef, nf, gf = _rand_inputs_vector(rng, adjs, in_dims)
    y = O.unbatch_dense(O.block_forward_dense(p, O.batch_dense(adjs, ef, nf, gf)))
    csc = O.csc_from_adj(adjs)
    se, sn, sg = O.block_forward_sparse(p, csc, O.packed_from_julia_vector(ef), O.packed_from_julia_vector(nf),
                                        O.packed_from_julia_vector(gf))
    np.testing.assert_allclose(np.concatenate([x.T for x in y["ef"]]), se[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np.concatenate([x.T for x in y["nf"]]), sn[0], rtol=1e-12, atol=1e-12)
    if out_dims[2]:
        np.testing.assert_allclose(np.stack(y["gf"]), sg[0], rtol=1e-12, atol=1e-12)
    else:
        assert y["gf"] is None and sg is None  # gnblock.jl:71-78, runtests.jl:156


@pytest.mark.parametrize("in_dims", IN_COMBOS)
def test_dense_equals_sparse_shared_mode(in_dims):
    rng = np.random.default_rng(7 + sum(in_dims))
    out_dims = (3, 4, 5)
    p = O.make_block_params(rng, in_dims, out_dims, act=(O.ACT_RELU, O.ACT_TANH, O.ACT_SIGMOID))
    ef, nf, gf = _rand_inputs_shared(rng, README_ADJ, in_dims, B=3)
    y = O.unbatch_dense(O.block_forward_dense(p, O.batch_dense(README_ADJ, ef, nf, gf)))
    assert y["ef"].shape == (3, 5, 3) and y["nf"].shape == (4, 3, 3) and y["gf"].shape == (5, 3)
    csc = O.csc_from_adj([README_ADJ])
    pk = O.packed_from_julia_shared
    se, sn, sg = O.block_forward_sparse(p, csc, pk(ef), pk(nf), None if gf is None else pk(gf[:, None, :]))
    np.testing.assert_allclose(pk(y["ef"]), se, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pk(y["nf"]), sn, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pk(y["gf"][:, None, :]), sg, rtol=1e-12, atol=1e-12)


def test_padded_slots_never_influence_real_outputs():
    """junk in padded slots (the reference leaves act(b) there) does not leak."""
    rng = np.random.default_rng(11)
    adjs = [_rand_adj(rng, 2), _rand_adj(rng, 5)]
    p = O.make_block_params(rng, (3, 2, 4), (3, 4, 5))
    ef, nf, gf = _rand_inputs_vector(rng, adjs, (3, 2, 4))
    x = O.batch_dense(adjs, ef, nf, gf)
    y0 = O.unbatch_dense(O.block_forward_dense(p, x))
    g = x["graphs"]
    x["ef"][:, ~g.flat_edge_unpadder.reshape(-1, len(adjs), order="F")[:, 0], 0] = 1e3  # junk in graph 0's pads
    x["nf"][:, adjs[0].shape[0]:, 0] = -1e3
    y1 = O.unbatch_dense(O.block_forward_dense(p, x))
    for k in ("ef", "nf", "gf"):
        for a, b in zip(y0[k], y1[k]):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-9)


def test_batch_invariance():
    """runtests.jl:62-116: graph A alone == graph A inside a padded heterogeneous batch [A, B], two stacked
    blocks (0,2,0)→(2,2,2)→(2,2,2), ef = gf = nothing."""
    rng = np.random.default_rng(3)
    enc = O.make_block_params(rng, (0, 2, 0), (2, 2, 2))
    dec = O.make_block_params(rng, (2, 2, 2), (2, 2, 2))
    A, Bm = np.ones((2, 2), dtype=int), np.ones((3, 3), dtype=int)
    nfs = [rng.random((2, 2), dtype=np.float32), rng.random((2, 3), dtype=np.float32)]
    run = lambda adjs, nf: O.unbatch_dense(O.block_forward_dense(dec, O.block_forward_dense(enc, O.batch_dense(adjs, None, nf, None))))
    y1, yn = run([A], nfs[:1]), run([A, Bm], nfs)
    # a 1-graph batch unbatches through the shared-adjacency branch (unbatch.jl:15-17)
    np.testing.assert_allclose(y1["nf"][:, :, 0], yn["nf"][0], rtol=1e-12)
    np.testing.assert_allclose(y1["ef"][:, :, 0], yn["ef"][0], rtol=1e-12)
    np.testing.assert_allclose(y1["gf"][:, 0], yn["gf"][0], rtol=1e-12)


def test_batch_inverse_2d_and_3d():
    """runtests.jl:328-390: unbatch(batch(x)) == x exactly, vector and shared modes."""
    rng = np.random.default_rng(5)
    adjs = [README_ADJ, README_ADJ2]
    ef, nf, gf = _rand_inputs_vector(rng, adjs, (10, 5, 3))
    y = O.unbatch_dense(O.batch_dense(adjs, ef, nf, gf))
    for k, ref in (("ef", ef), ("nf", nf), ("gf", gf)):
        for a, b in zip(y[k], ref):
            assert np.array_equal(a, b.astype(np.float64))
    ef, nf, gf = _rand_inputs_shared(rng, README_ADJ, (10, 5, 3), B=2)
    y = O.unbatch_dense(O.batch_dense(README_ADJ, ef, nf, gf))
    assert np.array_equal(y["ef"], ef) and np.array_equal(y["nf"], nf) and np.array_equal(y["gf"], gf)


def test_flatunpadded_is_packed_layout():
    """views.jl:80-98: flatunpaddednf/ef of a vector batch = graph-major packed (D, ΣT) = the HIP path's layout."""
    rng = np.random.default_rng(6)
    adjs = [README_ADJ, README_ADJ2]
    ef, nf, gf = _rand_inputs_vector(rng, adjs, (3, 4, 0))
    x = O.batch_dense(adjs, ef, nf, gf)
    np.testing.assert_array_equal(O.flat_from_dense(x, "ef").T, O.packed_from_julia_vector(ef)[0])
    np.testing.assert_array_equal(O.flat_from_dense(x, "nf").T, O.packed_from_julia_vector(nf)[0])


def test_core_dense_equals_sparse_and_shapes():
    """runtests.jl:685-735 shapes + value agreement of the two forms, both eps conventions."""
    for eps_mode in (0, 1):
        rng = np.random.default_rng(9 + eps_mode)
        dims = (3, 4, 5)
        p = O.make_core_params(rng, dims, eps_mode=eps_mode)
        ef, nf, gf = _rand_inputs_shared(rng, README_ADJ, dims, B=2)
        y = O.unbatch_dense(O.core_forward_dense(p, O.batch_dense(README_ADJ, ef, nf, gf)))
        assert y["ef"].shape == (3, 5, 2) and y["nf"].shape == (4, 3, 2) and y["gf"].shape == (5, 2)
        pk = O.packed_from_julia_shared
        se, sn, sg = O.core_forward_sparse(p, O.csc_from_adj([README_ADJ]), pk(ef), pk(nf), pk(gf[:, None, :]))
        np.testing.assert_allclose(pk(y["ef"]), se, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(pk(y["nf"]), sn, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(pk(y["gf"][:, None, :]), sg, rtol=1e-11, atol=1e-11)


def test_layernorm_eps_conventions():
    x = np.array([[1.0, 2.0, 4.0]]).T
    m0 = O.layernorm(x, np.ones(3), np.zeros(3), eps=1e-5, eps_mode=0)
    sigma = x.std()
    np.testing.assert_allclose(m0[:, 0], (x[:, 0] - x.mean()) / (sigma + 1e-5))
    m1 = O.layernorm(x, np.ones(3), np.zeros(3), eps=1e-5, eps_mode=1)
    np.testing.assert_allclose(m1[:, 0], (x[:, 0] - x.mean()) / np.sqrt(sigma**2 + 1e-5))


def _rand_csc(rng, sizes, p):
    adjs = [(rng.random((n, n)) < p).astype(np.int64) for n in sizes]
    return adjs, O.csc_from_adj(adjs)


@pytest.mark.parametrize("in_dims", IN_COMBOS)
def test_c_port_block_matches_float64_oracle(in_dims):
    from oracle import c_port
    rng = np.random.default_rng(21 + sum(in_dims))
    adjs, csc = _rand_csc(rng, (7, 1, 12, 30), 0.3)
    N, E, G = csc[2][-1], csc[3][-1], 4
    out_dims = (3, 4, 5)
    p = O.make_block_params(rng, in_dims, out_dims, act=(O.ACT_RELU, O.ACT_IDENTITY, O.ACT_TANH))
    R = 2
    ef = rng.random((R, E, in_dims[0]), dtype=np.float32) if in_dims[0] else None
    nf = rng.random((R, N, in_dims[1]), dtype=np.float32) if in_dims[1] else None
    gf = rng.random((R, G, in_dims[2]), dtype=np.float32) if in_dims[2] else None
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    got = c_port.block_forward(p, csc, ef, nf, gf)
    for g_, r_, s_ in zip(got, ref, scale):
        assert np.all(np.abs(g_ - r_) <= 1e-5 * s_)


def test_c_port_core_matches_float64_oracle():
    from oracle import c_port
    rng = np.random.default_rng(33)
    adjs, csc = _rand_csc(rng, (9, 17, 4), 0.4)
    N, E, G = csc[2][-1], csc[3][-1], 3
    dims = (6, 5, 3)
    for eps_mode in (0, 1):
        p = O.make_core_params(rng, dims, eps_mode=eps_mode)
        ef, nf, gf = (rng.random((1, T, d), dtype=np.float32) for T, d in zip((E, N, G), dims))
        ref = O.core_forward_sparse(p, csc, ef, nf, gf)
        got = c_port.core_forward(p, csc, ef, nf, gf)
        for g_, r_ in zip(got, ref):
            np.testing.assert_allclose(g_, r_, rtol=2e-4, atol=2e-4)  # LayerNorm amplifies fp32 rounding by 1/sigma


def test_three_independent_restatements_agree():
    """Absolute numerics cannot be pinned to the reference (Julia, no golden values): what CAN be checked on the CPU is
    that three restatements written independently of each other agree to float64 rounding — the literal one-hot
    batched-matmul form (oracle i), the sparse CSC form (oracle ii) and a torch index_select / index_add form (the one
    the backward tests differentiate) — on a heterogeneous batch with every activation."""
    import torch
    from tests.test_gpu_backward import _torch_block
    rng = np.random.default_rng(77)
    adjs = [(rng.random((n, n)) < 0.4).astype(np.int64) for n in (1, 4, 7, 3)]
    din, dout = (5, 3, 2), (4, 6, 3)
    for act in [(0, 0, 0), (1, 2, 3), (4, 1, 2)]:
        p = O.make_block_params(rng, din, dout, act=act)
        ef = [rng.random((din[0], int(a.sum())), dtype=np.float32) for a in adjs]
        nf = [rng.random((din[1], a.shape[0]), dtype=np.float32) for a in adjs]
        gf = [rng.random(din[2], dtype=np.float32) for _ in adjs]
        dense = O.unbatch_dense(O.block_forward_dense(p, O.batch_dense(adjs, ef, nf, gf)))
        csc = O.csc_from_adj(adjs)
        pk = lambda parts: np.concatenate([np.atleast_2d(q.T) if q.ndim == 2 else q[None, :] for q in parts], axis=0)[None]
        sp = O.block_forward_sparse(p, csc, pk(ef), pk(nf), pk(gf))
        if act[0] != 4 and act[1] != 4 and act[2] != 4:  # the torch form has no gelu
            W = {k: torch.tensor(p[k], dtype=torch.float64) for k in ("We", "be", "Wn", "bn", "Wg", "bg")}
            t = lambda a: torch.tensor(a[0], dtype=torch.float64)
            th = _torch_block(p, csc, t(pk(ef)), t(pk(nf)), t(pk(gf)), W)
            for a, b in zip(th, sp):
                np.testing.assert_allclose(a.numpy(), b[0], rtol=1e-12, atol=1e-12)
        off_n = np.concatenate([[0], np.cumsum([a.shape[0] for a in adjs])])
        off_e = np.concatenate([[0], np.cumsum([int(a.sum()) for a in adjs])])
        for i in range(len(adjs)):
            np.testing.assert_allclose(dense["ef"][i].T, sp[0][0, off_e[i]:off_e[i + 1]], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(dense["nf"][i].T, sp[1][0, off_n[i]:off_n[i + 1]], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(dense["gf"][i], sp[2][0, i], rtol=1e-12, atol=1e-12)


def test_propagated_error_scale_bounds_the_effect_of_input_perturbations():
    """`return_scale` / `in_scale` (the parity tests' tolerance scale): perturb every input by at most eta * s and the outputs
    move by at most eta * scale_out (first order), for a block, a core (LayerNorm's 1/σ!) and a core fed by a block."""
    rng = np.random.default_rng(77)
    adjs = [(rng.random((n, n)) < 0.4).astype(np.int64) for n in (5, 9, 3, 12)]
    csc = O.csc_from_adj(adjs)
    E, N, G = len(csc[1]), len(csc[0]) - 1, len(adjs)
    dims = (6, 5, 3)
    pb, pc = O.make_block_params(rng, dims, dims), O.make_core_params(rng, dims)
    x = [rng.random((1, T, d)) for T, d in zip((E, N, G), dims)]
    eta = 1e-7
    for trial in range(5):
        xp = [a + eta * np.abs(a) * rng.uniform(-1, 1, a.shape) for a in x]
        for fwd in (lambda z, sc=None: O.block_forward_sparse(pb, csc, *z, return_scale=True, in_scale=sc),
                    lambda z, sc=None: O.core_forward_sparse(pc, csc, *z, return_scale=True, in_scale=sc)):
            (y, s), (yp, _) = fwd(x), fwd(xp)
            for a, b, sc in zip(y, yp, s):
                assert np.all(np.abs(a - b) <= 1.01 * eta * sc + 1e-13)
        # chained: the block's scale is the core's in_scale
        (y1, s1), (y1p, _) = O.block_forward_sparse(pb, csc, *x, return_scale=True), O.block_forward_sparse(pb, csc, *xp, return_scale=True)
        (y2, s2) = O.core_forward_sparse(pc, csc, *y1, return_scale=True, in_scale=s1)
        y2p = O.core_forward_sparse(pc, csc, *y1p)
        for a, b, sc in zip(y2, y2p, s2):
            assert np.all(np.abs(a - b) <= 1.01 * eta * sc + 1e-13)


def test_chain_oracle_reduces_to_the_block_for_one_layer_chains():
    """chain_block_forward_sparse (Chain update functions, gnblock.jl:1-6) with one Dense per chain IS block_forward_sparse."""
    rng = np.random.default_rng(31)
    adjs = [(rng.random((n, n)) < 0.35).astype(np.int64) for n in (4, 9, 6)]
    csc = O.csc_from_adj(adjs)
    E, N, G = len(csc[1]), len(csc[0]) - 1, len(adjs)
    pb = O.make_block_params(rng, (3, 2, 1), (4, 3, 2), act=(1, 2, 3))
    pc = dict(in_dims=(3, 2, 1), edge=[(pb["We"], pb["be"], 1)], node=[(pb["Wn"], pb["bn"], 2)], graph=[(pb["Wg"], pb["bg"], 3)])
    x = [rng.random((2, T, d)) for T, d in zip((E, N, G), (3, 2, 1))]
    (a, sa), (b, sb) = O.block_forward_sparse(pb, csc, *x, return_scale=True), O.chain_block_forward_sparse(pc, csc, *x, return_scale=True)
    for u, v, su, sv in zip(a, b, sa, sb):
        np.testing.assert_allclose(u, v, rtol=0, atol=1e-13)
        np.testing.assert_allclose(su, sv, rtol=1e-5)


def test_edge_collapsing_restatement_satisfies_the_reference_test():
    """test/runtests.jl:4-59 on the oracle's literal restatement (edge_collapser matmul, gngraphbatch.jl:56-111): the relations the
    reference asserts between flatunpaddedcollapsedef and the raw padded slots of two stacked blocks on fully connected graphs."""
    rng = np.random.default_rng(91)
    enc, dec = O.make_block_params(rng, (0, 2, 0), (2, 2, 2)), O.make_block_params(rng, (2, 2, 2), (2, 2, 2))
    A, B = np.ones((2, 2)), np.ones((3, 3))
    x = O.batch_dense([A, B], None, [rng.random((2, 2)), rng.random((2, 3))], None)
    y = O.block_forward_dense(dec, O.block_forward_dense(enc, x))
    flat = np.concatenate(O.unpaddedcollapsedef_dense(y), axis=1)
    ef = y["ef"]
    s = lambda slot, g: ef[:, slot - 1, g]
    expect = [s(1, 0), (s(2, 0) + s(4, 0)) / 2, s(5, 0),
              s(1, 1), (s(2, 1) + s(4, 1)) / 2, (s(3, 1) + s(7, 1)) / 2, s(5, 1), (s(6, 1) + s(8, 1)) / 2, s(9, 1)]
    assert flat.shape == (2, 9)
    for c, e in enumerate(expect):
        np.testing.assert_allclose(flat[:, c], e, rtol=1e-12, atol=1e-12)
    assert O.collapsef_dense(y).shape == (2, 6, 2)
    assert O.getedgecollapser(3).shape == (9, 6) and O.getedgecollapser(3)[:, 0].tolist() == [2, 0, 0, 0, 0, 0, 0, 0, 0]

"""Host-side behaviour of the Python mirror that needs no GPU: refusals that mirror what the reference would do differently."""
import pytest
import torch

import graphnets_jl_amd as gn


def _cpu_core(dropout):
    return gn.GNCore((4, 3, 2), dropout=dropout, device="cpu")


def test_dropout_is_active_inside_a_gradient_call_and_the_identity_outside():
    """gnfeedforward.jl:27-31: Chain(Dense, Dense, Dropout(p)).  Flux applies the Dropout inside a gradient call and skips it in test mode;
    testmode! / trainmode! force it.  Host logic only (no GPU): which calls get a gnx_dropout, with which seed."""
    core = _cpu_core(0.1)
    assert core._dropout_now(False) is None                      # test mode: identity
    d = core._dropout_now(True)                                  # gradient call: active, p as given
    assert abs(d.p - 0.1) < 1e-7 and core.last_dropout is d
    torch.manual_seed(3); a = core._dropout_now(True).seed
    torch.manual_seed(3); b = core._dropout_now(True).seed
    c = core._dropout_now(True).seed
    assert a == b and a != c and 0 <= a < 2 ** 63                # reproducible under torch.manual_seed, fresh per call
    gn.testmode(core)
    assert core._dropout_now(True) is None
    gn.trainmode(core)
    assert core._dropout_now(False) is not None
    gn.trainmode(core, None)                                     # automatic again
    assert core._dropout_now(False) is None and core._dropout_now(True) is not None
    assert _cpu_core(0)._dropout_now(True) is None               # p = 0: never
    lst = gn.GNCoreList([_cpu_core(0.2), _cpu_core(0.3)])
    assert gn.testmode(lst) is lst and all(c._dropout_mode is False for c in lst.list)
    # the call goes on to the ordinary argument checks in either mode
    for t in core.parameters():
        t.requires_grad_(True)
    with pytest.raises(AssertionError, match="ef, nf and gf"):
        core(dict(graphs=None, ef=None, nf=None, gf=None))
    with torch.no_grad(), pytest.raises(AssertionError, match="ef, nf and gf"):
        core(dict(graphs=None, ef=None, nf=None, gf=None))


def test_chain_folds_the_layer_values_that_fold_exactly_and_refuses_the_rest():
    """gnblock.jl:1-6 allows any Flux chain as an update function.  Row-wise Dense layers run; an activation function as a layer folds into the
    Dense in front of it, identity is dropped, Dropout is the identity in test mode; everything else is an explicit error."""
    d = gn.Dense(4, 3, device="cpu")
    assert len(gn.Chain(d, gn.Dense(3, 2, device="cpu"))) == 2
    ch = gn.Chain(d, torch.relu, gn.Dense(3, 2, device="cpu"), "identity", "tanh", gn.Dropout(0.25), None)
    assert len(ch) == 2 and [l.act for l in ch.layers] == ["relu", "tanh"] and ch.dropout_p == 0.25
    assert ch.layers[0].weight is d.weight and ch.layers[0].bias is d.bias and d.act == "identity"   # the caller's Dense is not modified
    assert gn.Chain([d, "gelu"]).layers[0].act == "gelu"
    # round 6: a LayerNorm(d) layer value is a layer of its own (gnx_dense.kind = GNX_LAYER_LAYERNORM: gamma / beta in the weight / bias slots) ...
    ln = gn.LayerNorm(3, device="cpu")
    chl = gn.Chain(d, ln, gn.Dense(3, 2, device="cpu"))
    assert len(chl) == 3 and chl.layers[1] is ln and chl.out_width == 2 and ln.weight is ln.gamma and ln.bias is ln.beta
    keep = []
    c = ln._c_layer(keep)
    assert c.kind == gn._lib.LAYER_LAYERNORM and c.act == gn._lib.ACT["identity"] and c.weight and c.bias
    assert gn.Chain(d, ln).out_width == 3                          # (as a chain's last layer it keeps the width in front of it)
    with pytest.raises(NotImplementedError, match="does not follow"):
        gn.Chain(d, ln, "relu")                                   # ... but an activation behind it has no Dense to fold into
    with pytest.raises(NotImplementedError, match="BatchNorm"):
        gn.Chain(d, type("BatchNorm", (), {})())
    with pytest.raises(NotImplementedError, match="function"):
        gn.Chain(d, lambda x: x * 2)
    with pytest.raises(NotImplementedError, match="does not follow"):
        gn.Chain(torch.relu, d)                                   # an activation in front of the first Dense does not fold
    with pytest.raises(NotImplementedError, match="does not follow"):
        gn.Chain(gn.Dense(4, 3, "relu", device="cpu"), "tanh")    # nor do two activations in a row
    with pytest.raises(AssertionError):
        gn.Dropout(1.5)


def test_packed_csc_constructor_checks_lengths_before_anything_is_read():
    """the packed constructor takes both array lengths and the index width; a short colptr / rowval is an argument error (no
    read past the buffers), in Python a ValueError (not an assert that `python -O` strips).  No GPU needed: the checks precede any device work."""
    import ctypes as C
    import numpy as np
    lib = gn._lib.load()
    cp = np.array([0, 1, 3, 0, 2], dtype=np.int64)      # two graphs: 2 nodes / 3 edges, 1 node / 2 edges (the second is malformed on purpose below)
    rv = np.array([0, 0, 1, 0, 0], dtype=np.int64)
    nn = np.array([2, 1], dtype=np.int64)
    h = C.c_void_p()
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    call = lambda cpl, rvl, bits=64, base=0: lib.gnx_graphs_create_csc_cat(cp.ctypes.data, cpl, rv.ctypes.data, rvl, p64(nn), 2, base, bits, C.byref(h))
    assert call(4, 5) == gn._lib.ERR_INVALID_ARG and not h.value            # colptr shorter than sum(n) + G
    assert call(6, 5) == gn._lib.ERR_INVALID_ARG                            # ... longer
    assert call(5, 4) == gn._lib.ERR_INVALID_ARG                            # rowval shorter than the colptr arrays announce
    assert call(5, 6) == gn._lib.ERR_INVALID_ARG                            # ... longer
    assert call(5, 5, bits=16) == gn._lib.ERR_INVALID_ARG
    assert call(5, 5, base=2) == gn._lib.ERR_INVALID_ARG
    assert call(-1, 5) == gn._lib.ERR_INVALID_ARG
    assert call(5, 5) == gn._lib.ERR_CSC and not h.value                    # a 1-node graph with 2 in-edges: more than N per column (host pass, before the device)
    with pytest.raises(ValueError, match="n_nodes \\+ 1 entries"):
        gn.GNGraphBatch.from_csc_packed(cp[:-1], rv, nn)


def test_chained_forward_argument_checks():
    """gnx_block_forward_chained without a record to fill / with the deferral flag is refused before any device work."""
    import ctypes as C
    lib = gn._lib.load()
    pend = gn._lib.PendingUpdate()
    args = [None, None, None, None, None, 1, None, None, None, None, 0]
    assert lib.gnx_block_forward_chained(*args, 0, None, None, None) == gn._lib.ERR_INVALID_ARG
    assert lib.gnx_block_forward_chained(*args, gn._lib.FLAG_DEFER_GRAPH_UPDATE, None, None, C.byref(pend)) == gn._lib.ERR_INVALID_ARG
    assert lib.gnx_block_forward_chained(*args, 0, None, None, C.byref(pend)) == gn._lib.ERR_INVALID_ARG  # NULL handle / params

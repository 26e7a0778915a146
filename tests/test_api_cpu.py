"""Host-side behaviour of the Python mirror that needs no GPU: refusals that mirror what the reference would do differently."""
import pytest
import torch

import graphnets_jl_amd as gn


def _cpu_core(dropout):
    return gn.GNCore((4, 3, 2), dropout=dropout, device="cpu")


def test_dropout_in_training_mode_is_refused_not_ignored():
    """gnfeedforward.jl:27-31: Chain(Dense, Dense, Dropout(p)).  Flux applies the Dropout inside a gradient call; the HIP path has
    none, so a differentiable call with p > 0 raises instead of silently training another model."""
    core = _cpu_core(0.1)
    for t in core.parameters():
        t.requires_grad_(True)
    with pytest.raises(NotImplementedError, match="Dropout"):
        core(dict(graphs=None, ef=None, nf=None, gf=None))
    # test mode (no gradient): Dropout is the identity in Flux too — the call goes on to the ordinary argument checks
    with torch.no_grad(), pytest.raises(AssertionError, match="ef, nf and gf"):
        core(dict(graphs=None, ef=None, nf=None, gf=None))
    # p = 0 trains as before (reaches the argument checks)
    core0 = _cpu_core(0)
    for t in core0.parameters():
        t.requires_grad_(True)
    with pytest.raises(AssertionError, match="ef, nf and gf"):
        core0(dict(graphs=None, ef=None, nf=None, gf=None))


def test_non_dense_layer_in_a_chain_is_an_explicit_error():
    """gnblock.jl:1-6 allows any Flux chain as an update function; only Dense layers are supported here — and say so."""
    d = gn.Dense(4, 3, device="cpu")
    assert len(gn.Chain(d, gn.Dense(3, 2, device="cpu"))) == 2
    with pytest.raises(NotImplementedError, match="LayerNorm"):
        gn.Chain(d, gn.LayerNorm(3, device="cpu"))
    with pytest.raises(NotImplementedError, match="function"):
        gn.Chain(d, torch.relu)

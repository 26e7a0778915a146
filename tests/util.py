"""Shared helpers of the parity tests: build product layers from oracle parameter dicts, seeded inputs, tolerance."""
import numpy as np

from oracle import gn_oracle as O

ACT_NAMES = {0: "identity", 1: "relu", 2: "tanh", 3: "sigmoid", 4: "gelu"}
RTOL = 1e-5  # north_star: "outputs match the reference Julia CPU path within 1e-5 fp32" — relative to the magnitude
             # bound |W|·|x|+|b| of each output (oracle `return_scale`), i.e. 1e-5 of the sum of absolute terms.


def block_from_params(gn, p, device=None):
    b = gn.GNBlock(p["in_dims"], p["out_dims"], device=device)
    b.edgefn = gn.Dense.from_numpy(p["We"], p["be"], ACT_NAMES[p["act_e"]], device)
    b.nodefn = gn.Dense.from_numpy(p["Wn"], p["bn"], ACT_NAMES[p["act_n"]], device)
    b.graphfn = gn.Dense.from_numpy(p["Wg"], p["bg"], ACT_NAMES[p["act_g"]], device)
    return b


def core_from_params(gn, p, device=None):
    import torch
    c = gn.GNCore(p["dims"], device=device, eps=p["eps"], eps_mode=p["eps_mode"])
    c.block = block_from_params(gn, p["block"], device)
    dev = c.gn1.edgeln.gamma.device
    for t, l1, l2 in zip("eng", (c.gn1.edgeln, c.gn1.nodeln, c.gn1.graphln), (c.gn2.edgeln, c.gn2.nodeln, c.gn2.graphln)):
        l1.gamma = torch.from_numpy(p[f"ln1_{t}_gamma"]).to(dev); l1.beta = torch.from_numpy(p[f"ln1_{t}_beta"]).to(dev)
        l2.gamma = torch.from_numpy(p[f"ln2_{t}_gamma"]).to(dev); l2.beta = torch.from_numpy(p[f"ln2_{t}_beta"]).to(dev)
    mk = lambda t: (gn.Dense.from_numpy(p[f"ff_{t}_W1"], p[f"ff_{t}_b1"], "relu", device),
                    gn.Dense.from_numpy(p[f"ff_{t}_W2"], p[f"ff_{t}_b2"], "identity", device))
    c.ffwd.eff, c.ffwd.nff, c.ffwd.gff = mk("e"), mk("n"), mk("g")
    return c


def random_graphs(rng, sizes, p):
    return [(rng.random((n, n)) < p).astype(np.int64) for n in sizes]


def er_csc(rng, N, E):
    """SURVEY §8d C2 recipe: E distinct directed pairs (self-loops allowed) of an N-node graph, reference edge
    order (sorted by dst then src).  Returns per-graph CSC (colptr, rowval) 0-based."""
    k = np.unique(rng.integers(0, N * N, int(E * 1.1) + 16))
    k = np.sort(rng.permutation(k)[:E])
    dst, src = k // N, k % N
    colptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(colptr, dst + 1, 1)
    return np.cumsum(colptr), src.astype(np.int64)


def packed_inputs(rng, R, E, N, G, dims):
    de, dn, dg = dims
    ef = rng.random((R, E, de), dtype=np.float32) if de else None
    nf = rng.random((R, N, dn), dtype=np.float32) if dn else None
    gf = rng.random((R, G, dg), dtype=np.float32) if dg else None
    return ef, nf, gf


def to_nt(gn, g, ef, nf, gf):
    """packed numpy [R][T][D] → the batched tuple the layers take (Julia-shaped views of device tensors)."""
    import torch
    mk = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(g.device).permute(2, 1, 0)
    return gn.NT(g, mk(ef), mk(nf), mk(gf))


def from_jl(a):
    """Julia-shaped device view (D, T, R) → packed numpy [R][T][D]."""
    return None if a is None else a.permute(2, 1, 0).contiguous().cpu().numpy()


PARITY_LOG = {}  # label -> {tensor: (worst |diff| / (1e-5·S), plain max|diff| / max|ref|)}: written by the full-size tests, dumped by conftest.py


def assert_close(got, ref, scale, what="", log=None, normwise=None):
    """|got - ref| <= 1e-5 · scale elementwise.  `log`: a label under which the two summary figures of this comparison are recorded —
    the worst ratio to that bound and the plain normwise error max|diff| / max|ref| (no scale involved) — for the session report.
    `normwise` (default: on for the logged = full-size comparisons): ALSO assert the plain bound of BASELINE.json's "within 1e-5 fp32" with no
    error scale in it — max|got - ref| <= 1e-5 · max|ref| over the tensor."""
    if ref is None:
        assert got is None, f"{what}: expected nothing"
        return
    assert got is not None and got.shape == ref.shape, f"{what}: shape {None if got is None else got.shape} vs {ref.shape}"
    err = np.abs(got.astype(np.float64) - ref)
    bad = err > RTOL * scale + 1e-30
    if log is not None:
        PARITY_LOG.setdefault(log, {})[what] = (float(np.max(err / (RTOL * scale + 1e-30))), float(np.max(err) / max(float(np.max(np.abs(ref))), 1e-30)))
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} outside 1e-5·scale; worst ratio {np.max(err / (scale + 1e-30)):.3e}"
    if (log is not None) if normwise is None else normwise:
        d, m = (float(np.max(err)), float(np.max(np.abs(ref)))) if err.size else (0.0, 0.0)
        assert d <= RTOL * m + 1e-30, f"{what}: max|diff| = {d:.3e} above the plain bound 1e-5 · max|ref| = {RTOL * m:.3e}"


def assert_same_formula(a, b, what="", k=2e-6):
    """Two fp32 evaluations of the SAME formula in different associations (e.g. a LayerNorm's scale applied to the rows, or folded into the weight
    planes they are multiplied with): finite, and normwise within a few fp32 roundings — max|a - b| <= k · max|b|.  (The bound against the float64
    oracle, 1e-5·S elementwise, is asserted separately by the callers.)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.isfinite(a).all() and np.isfinite(b).all(), f"{what}: not finite"
    d, m = float(np.abs(a - b).max()) if a.size else 0.0, float(np.abs(b).max()) if b.size else 0.0
    assert d <= k * m + 1e-30, f"{what}: max|a - b| = {d:.3e} against {k:g} · max|b| = {k * m:.3e}"


def oracle_layer(kind, p, csc, x, in_scale=None):
    """One layer of the float64 oracle on packed (ef, nf, gf) with its worst-case error scale."""
    if kind == "block":
        return O.block_forward_sparse(p, csc, *x, return_scale=True, in_scale=in_scale)
    return O.core_forward_sparse(p, csc, *x, return_scale=True, in_scale=in_scale)


def check_chain(gn, g, csc, layers, x0, what="chain", normwise=1e-5):
    """Parity of a multi-layer model (list of (kind, oracle params, HIP layer)) against the oracle, two ways:

    (1) LAYER BY LAYER, elementwise, at the 1e-5 bar: the reference stores float32 arrays between layers, so the oracle chain
        does too — every layer is computed in float64 from the float32-rounded output of the previous oracle layer, the HIP
        layer gets the very same float32 input, and |hip - oracle| <= 1e-5 * (|W|·|x| + |b|) propagated through that ONE layer
        (LayerNorm's 1/σ included: O.layernorm_scale).  This is where a wrong kernel shows, at any depth, with a tight bound.
    (2) END TO END, free-running HIP chain vs that oracle chain, normwise per tensor: max|diff| <= `normwise` * max|ref|.
        (A worst-case elementwise bound propagated through several layers of million-term sums and 1/σ is vacuous — ~1e8 x
        the value at BASELINE config 4 — and errors of shared rows are correlated, so no quadrature bound holds either.)
    Returns {tensor: (worst layer-wise ratio to the 1e-5 bound, end-to-end max|diff| / max|ref|)}."""
    import torch
    names = ("ef", "nf", "gf")
    dev = g.device
    to_dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev).permute(2, 1, 0)
    # free-running HIP chain
    y = gn.NT(g, *(to_dev(a) for a in x0))
    for _, _, layer in layers:
        y = layer(y)
    free = [from_jl(a) for a in (y.ef, y.nf, y.gf)]
    del y
    worst = {n: 0.0 for n in names}
    x = tuple(None if a is None else np.asarray(a, dtype=np.float32) for a in x0)
    for li, (kind, p, layer) in enumerate(layers):
        ref, scale = oracle_layer(kind, p, csc, x)
        yt = layer(gn.NT(g, *(to_dev(a) for a in x)))
        for n, got, r, s in zip(names, (yt.ef, yt.nf, yt.gf), ref, scale):
            if r is None:
                assert got is None
                continue
            err = np.abs(from_jl(got).astype(np.float64) - r)
            if err.size == 0:  # (a batch without edges: (de, 0) edge features)
                assert from_jl(got).shape == r.shape
                continue
            ratio = float(np.max(err / (RTOL * s + 1e-30)))
            worst[n] = max(worst[n], ratio)
            assert ratio <= 1.0, f"{what}: layer {li} ({kind}) {n}: {int((err > RTOL * s + 1e-30).sum())} of {err.size} outside 1e-5·scale; worst ratio {ratio:.3f}"
        del yt
        x = tuple(None if r is None else r.astype(np.float32) for r in ref)  # float32 storage between layers, as in the reference
    out = {}
    for n, got, r in zip(names, free, ref):
        if r is None:
            continue
        if r.size == 0:
            continue
        rel = float(np.max(np.abs(got.astype(np.float64) - r)) / max(float(np.max(np.abs(r))), 1e-30))
        assert rel <= normwise, f"{what}: end-to-end {n}: max|diff| / max|ref| = {rel:.3e} > {normwise:g}"
        out[n] = (worst[n], rel)
    return out


import contextlib


@contextlib.contextmanager
def with_flags(layer, flags):
    """`layer` (GNBlock / GNCore of the mirror) runs with these GNX_FLAG_* forms added to its default flags inside the block — the per-call
    selection of include/gnx.h (round 5: the environment variables of the same names are process-wide defaults read ONCE, no longer per call)."""
    old = layer.flags
    layer.flags = old | int(flags)
    try:
        yield layer
    finally:
        layer.flags = old


def default_flags(gn):
    """forms the environment switched on for this process (gnx_default_flags)"""
    return int(gn._lib.load().gnx_default_flags())


def needs_default_forms(gn, *names):
    """Skips a test that asserts WHICH kernels run (or compares forms that only differ under the default selection) when the process-wide default
    (the GNX_* environment variables, read once: gnx_default_flags) switches one of the named forms on: the suite stays green under
    `GNX_FFN_FP32=1 pytest …`, `GNX_CORE_EDGE_SPLIT=1 pytest …` etc., where the remaining tests check that form against the oracle."""
    import pytest
    on = [n for n in names if default_flags(gn) & getattr(gn._lib, "FLAG_" + n)]
    if on:
        pytest.skip("switched on for the whole run: " + ", ".join("GNX_" + n for n in on))


ONE_LAUNCH_CORE_FORMS = ("FFN_FP32", "EDGE_FP32", "LN_STATS_PASS", "CORE_EDGE_SPLIT", "EDGE_N", "NO_LN_FUSE")  # any of these: no k_core_edge_x6

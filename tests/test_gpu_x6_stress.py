"""Stress of the six-term arithmetic and of the index arithmetic at scale (VERDICT r4 "next" 6).

The wide kernels evaluate every fp32 product on the bf16 matrix cores as six terms of an exact three-way split of both operands (include/gnx.h:
"as accurate as the fp32 matrix instruction").  That is a claim about ARITHMETIC, so it is tested on inputs that stress the split — magnitudes
over twelve decades with random signs, rows whose terms cancel to 1e-4 of their size, a weight column scaled by 1e20 — against the float64
oracle at 1e-5·S and against the fp32-MFMA kernels of the same call (GNX_FLAG_FP32_MFMA), whose mean error the six-term form must match;
the documented edge behaviours (non-finite in -> NaN out; operands below ~1e-33 keep 16 mantissa bits) are asserted; and tensors with more
than 2^31 elements run through the wide path (17M edges x 128) and the narrow path (72M edges x 32) with sampled rows against the oracle."""
import numpy as np
import pytest

from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
DIMS = (128, 64, 32)


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn_
    return gn_


def _graph(gn, rng, n=4400, e=12000):
    colptr, rowval = U.er_csc(rng, n, e)
    return gn.GNGraphBatch.from_csc([colptr], [rowval], [n])


def _errors(y, ref, scale):
    """per tensor: (worst, mean) of |got - ref| / scale"""
    out = {}
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        err = np.abs(U.from_jl(got).astype(np.float64) - r) / (s + 1e-300)
        out[name] = (float(err.max()), float(err.mean()))
    return out


def _log_uniform(rng, shape, lo=-6.0, hi=6.0):
    return (10.0 ** rng.uniform(lo, hi, size=shape) * rng.choice([-1.0, 1.0], size=shape)).astype(np.float32)


@pytest.mark.parametrize("case", ["log_uniform_inputs", "log_uniform_weights", "cancellation", "column_times_1e20"])
def test_six_term_block_against_float64_and_the_fp32_matrix_instruction(gn, case):
    """GNBlock (128,64,32) => (128,64,32) from 4096 nodes on: k_proj_x6 + k_edge_x6 (six bf16 terms) and, on the same call with
    GNX_FLAG_FP32_MFMA, k_rows_gemm (fp32 matrix instruction).  Both within 1e-5·S of float64; the six-term form's mean error within 1.1 x the
    fp32 instruction's and its worst within 1.5 x, per tensor that goes through the six-term kernels (ef, and nf / gf which read its sums)."""
    rng = np.random.default_rng({"log_uniform_inputs": 1, "log_uniform_weights": 2, "cancellation": 3, "column_times_1e20": 4}[case] + 6000)
    g = _graph(gn, rng)
    E, N = g.n_edges, g.n_nodes
    p = O.make_block_params(rng, DIMS, DIMS)
    ef, nf, gf = U.packed_inputs(rng, 1, E, N, 1, DIMS)
    if case == "log_uniform_inputs":  # every element its own magnitude, 1e-6 .. 1e6, random sign
        ef, nf = _log_uniform(rng, ef.shape), _log_uniform(rng, nf.shape)
    elif case == "log_uniform_weights":
        p["We"] = (p["We"] * 10.0 ** rng.uniform(-4, 4, size=p["We"].shape)).astype(np.float32)
        ef, nf = (ef * 2 - 1).astype(np.float32), (nf * 2 - 1).astype(np.float32)
    elif case == "cancellation":
        # a common mode of 1e4 on every input against weight columns that sum to ~0 over each input block: sum|terms| >= 1e4 |result|
        We = p["We"].astype(np.float64)  # (out, in): in = [ef 128 | nf_src 64 | nf_dst 64 | gf 32]
        for a, b in ((0, 128), (128, 192), (192, 256)):
            We[:, a:b] -= We[:, a:b].mean(axis=1, keepdims=True)
        p["We"] = We.astype(np.float32)
        ef, nf = (ef + 1e4).astype(np.float32), (nf + 1e4).astype(np.float32)
    else:  # one output column of the edge function 1e20 times the others
        p["We"][7, :] *= np.float32(1e20)
        p["be"][7] *= np.float32(1e20)
    csc = (*g.csc(), g.node_off, g.edge_off)
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    if case == "cancellation":
        he = np.abs(ref[0])
        assert np.median(scale[0] / (he + 1e-300)) >= 1e4, "the case must cancel: sum|terms| >= 1e4 |result| for the typical element"
    blk = U.block_from_params(gn, p)
    x = U.to_nt(gn, g, ef, nf, gf)
    gn.profile_reset(); gn.profile_enable(True)
    y6 = blk(x)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    if not U.default_flags(gn) & gn._lib.FLAG_EDGE_FP32:
        assert "k_edge_x6_prep" in names and "k_proj_x6_prep" in names, names
    y32 = blk(x, flags=gn._lib.FLAG_FP32_MFMA)
    e6, e32 = _errors(y6, ref, scale), _errors(y32, ref, scale)
    edge_n = bool(U.default_flags(gn) & gn._lib.FLAG_EDGE_N)
    print(case, "six terms", e6, "fp32 MFMA", e32)
    for name in ("ef", "nf", "gf"):
        assert np.isfinite(U.from_jl(getattr(y6, name))).all()
        assert e6[name][0] <= 1e-5 and e32[name][0] <= 1e-5, (case, name, e6[name], e32[name])
        # "as accurate as the fp32 instruction": on ef (E x 128 elements go through the six-term products: the statistics mean something) mean within
        # 1.1 x and worst within 1.5 x; nf (sums of ef' rows in another order) 1.5 x / 2 x; gf is ONE row of 32 sums of 12 000 terms — bound only
        k_mean, k_max = {"ef": (1.1, 1.5), "nf": (1.5, 2.0), "gf": (None, None)}[name]
        if k_mean and not edge_n:  # (the opt-in k_edge_n sums on the matrix cores in another order: mean 1.3-1.45 x, worst 2.1 x the fp32 form's — bound only)
            assert e6[name][1] <= k_mean * e32[name][1] + 1e-12 and e6[name][0] <= k_max * e32[name][0] + 1e-12, (case, name, e6[name], e32[name])


@pytest.mark.parametrize("case", ["wide_gamma", "log_uniform_ffn_weights", "cancelling_hidden"])
def test_six_term_core_feedforward_against_float64_and_the_fp32_matrix_instruction(gn, case):
    """GNCore(128,64,32): k_ffn_x6 (edge FeedForward at 128, node FeedForward at 64; the edge update rides in the edge launch) against float64 at
    1e-5·S and against k_ffn_fused / k_rows_gemm on the fp32 matrix instruction (GNX_FLAG_FP32_MFMA on the same call).  LayerNorm scales over six
    decades (the FeedForward's input then spans them), FeedForward weights over eight decades, and a second layer that cancels the hidden units
    pairwise."""
    rng = np.random.default_rng({"wide_gamma": 1, "log_uniform_ffn_weights": 2, "cancelling_hidden": 3}[case] + 6100)
    g = _graph(gn, rng)
    E, N = g.n_edges, g.n_nodes
    p = O.make_core_params(rng, DIMS)
    ef, nf, gf = U.packed_inputs(rng, 1, E, N, 1, DIMS)
    ef = (ef * 3 - 1).astype(np.float32)
    if case == "wide_gamma":
        for t, d in zip("en", DIMS):
            p[f"ln2_{t}_gamma"] = _log_uniform(rng, (d,), -3, 3)
            p[f"ln2_{t}_beta"] = _log_uniform(rng, (d,), -3, 3)
    elif case == "log_uniform_ffn_weights":
        for t in "en":
            p[f"ff_{t}_W1"] = (p[f"ff_{t}_W1"] * 10.0 ** rng.uniform(-4, 4, size=p[f"ff_{t}_W1"].shape)).astype(np.float32)
            p[f"ff_{t}_W2"] = (p[f"ff_{t}_W2"] * 10.0 ** rng.uniform(-4, 4, size=p[f"ff_{t}_W2"].shape)).astype(np.float32)
    else:  # hidden units in identical pairs whose second-layer weights are opposite up to 1e-4: the second product cancels
        for t in "en":
            W1, W2 = p[f"ff_{t}_W1"], p[f"ff_{t}_W2"]  # (4d, d), (d, 4d)
            W1[1::2] = W1[0::2]
            p[f"ff_{t}_b1"][1::2] = p[f"ff_{t}_b1"][0::2]
            W2[:, 1::2] = -W2[:, 0::2] * np.float32(1.0001)
            p[f"ff_{t}_b1"] = (p[f"ff_{t}_b1"] + 1.0).astype(np.float32)  # (most units active)
    csc = (*g.csc(), g.node_off, g.edge_off)
    ref, scale = O.core_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    core = U.core_from_params(gn, p)
    x = U.to_nt(gn, g, ef, nf, gf)
    gn.profile_reset(); gn.profile_enable(True)
    y6 = core(x)
    gn.profile_enable(False)
    names = set(gn.profile_read()); gn.profile_reset()
    if not U.default_flags(gn) & (gn._lib.FLAG_FFN_FP32 | gn._lib.FLAG_EDGE_FP32):
        assert "k_ffn_x6_prep" in names, names
    y32 = core(x, flags=gn._lib.FLAG_FP32_MFMA)
    e6, e32 = _errors(y6, ref, scale), _errors(y32, ref, scale)
    edge_n = bool(U.default_flags(gn) & gn._lib.FLAG_EDGE_N)
    print(case, "six terms", e6, "fp32 MFMA", e32)
    for name in ("ef", "nf", "gf"):
        assert np.isfinite(U.from_jl(getattr(y6, name))).all()
        assert e6[name][0] <= 1e-5 and e32[name][0] <= 1e-5, (case, name, e6[name], e32[name])
        # "as accurate as the fp32 instruction": on ef (E x 128 elements go through the six-term products: the statistics mean something) mean within
        # 1.1 x and worst within 1.5 x; nf (sums of ef' rows in another order) 1.5 x / 2 x; gf is ONE row of 32 sums of 12 000 terms — bound only.
        # LayerNorm scales over six decades: the core's one-launch form has them in the weight planes (one more fp32 rounding per weight, and the
        # shift's contribution W^T beta summed apart from the scaled rows') — measured 1.2 x the fp32 form's mean at 8e-10 of the scale; bound 1.5 x
        k_mean, k_max = {"ef": (1.5 if case == "wide_gamma" else 1.1, 1.5), "nf": (1.5, 2.0), "gf": (None, None)}[name]
        if k_mean and not edge_n:  # (the opt-in k_edge_n sums on the matrix cores in another order: mean 1.3-1.45 x, worst 2.1 x the fp32 form's — bound only)
            assert e6[name][1] <= k_mean * e32[name][1] + 1e-12 and e6[name][0] <= k_max * e32[name][0] + 1e-12, (case, name, e6[name], e32[name])


def test_documented_edge_behaviours_of_the_six_term_arithmetic(gn):
    """include/gnx.h: "inputs that are not finite (or within 0.4 % of the largest finite float) produce NaN where the fp32 instruction may
    produce an infinity, and operands below ~1e-33 in magnitude may keep only 16 of their 24 mantissa bits" — asserted: an infinite and a
    near-maximal element turn exactly the rows they touch into NaN (never a finite wrong number), everything else stays within 1e-5·S; inputs of
    magnitude 1e-36 come out finite with a relative error of at most 2^-14 of their scale (and 1e-30 inputs at full accuracy)."""
    U.needs_default_forms(gn, "EDGE_FP32")  # (the fp32 instruction: an infinity where the six-term form gives NaN — the header's sentence)
    rng = np.random.default_rng(6200)
    g = _graph(gn, rng)
    E, N = g.n_edges, g.n_nodes
    p = O.make_block_params(rng, DIMS, DIMS, random_bias=False)
    blk = U.block_from_params(gn, p)
    ef, nf, gf = U.packed_inputs(rng, 1, E, N, 1, DIMS)
    csc = (*g.csc(), g.node_off, g.edge_off)
    # (1) non-finite / near-maximal elements
    bad = ef.copy()
    bad[0, 100, 5] = np.inf
    bad[0, 2000, 77] = np.float32(3.4e38)  # within 0.4 % of FLT_MAX: its high part rounds to bf16 infinity
    bad[0, 3000, 1] = np.nan
    y = blk(U.to_nt(gn, g, bad, nf, gf))
    out = U.from_jl(y.ef)[0]
    touched = np.zeros(E, dtype=bool); touched[[100, 2000, 3000]] = True
    assert np.isnan(out[touched]).all(), "a non-finite input must give NaN in every output of its row (never a finite number)"
    assert np.isfinite(out[~touched]).all()
    ref, scale = O.block_forward_sparse(p, csc, ef, nf, gf, return_scale=True)
    U.assert_close(out[~touched][None], ref[0][0][~touched][None], scale[0][0][~touched][None], "rows without a non-finite input")
    # (2) tiny operands: the low bf16 parts are subnormal below ~1e-33 (16 of 24 mantissa bits survive); from 1e-30 up the accuracy is full
    for mag, tol in ((1e-36, 2.0 ** -14), (1e-30, 1e-5)):
        e_t = (ef * mag).astype(np.float32)
        n_t = (nf * mag).astype(np.float32)
        g_t = (gf * mag).astype(np.float32)
        y = blk(U.to_nt(gn, g, e_t, n_t, g_t))
        ref, scale = O.block_forward_sparse(p, csc, e_t, n_t, g_t, return_scale=True)
        got = U.from_jl(y.ef)
        assert np.isfinite(got).all()
        err = np.abs(got.astype(np.float64) - ref[0]) / (scale[0] + 1e-300)
        assert err.max() <= tol, (mag, float(err.max()))


def _ring_graph(n, deg, rng):
    """n nodes, node j receives an edge from (j + o) mod n for `deg` fixed distinct offsets o: CSC with strictly increasing rows per column,
    built without a sort over the edges (17M - 72M edges on the host in seconds)"""
    offs = np.sort(rng.choice(np.arange(1, min(n, 100_000)), deg, replace=False)).astype(np.int64)
    rows = (np.arange(n, dtype=np.int64)[:, None] + offs[None, :]) % n
    rows.sort(axis=1)
    colptr = np.arange(n + 1, dtype=np.int64) * deg
    return colptr, rows.reshape(-1)


def _free_gb():
    import torch
    return torch.cuda.mem_get_info()[0] / 2 ** 30


def test_more_than_2_to_31_elements_through_the_wide_path(gn):
    """17M edges x 128 = 2.2e9 elements per edge tensor (8.7 GB in, 8.7 GB out): the matrix-core block (128,64,32) => (128,64,32).  Sampled rows —
    the first and last edges / nodes and 4000 random ones, past the 2^31-element mark included — against a float64 evaluation of the block's
    definition (edgefninput.jl:2-7, nodefninput.jl:2-6); gf' against float64 sums of the device's own ef' / nf'."""
    import torch
    if _free_gb() < 60:
        pytest.skip("needs ~45 GB of device memory")
    rng = np.random.default_rng(6300)
    n, deg = 1_700_000, 10
    colptr, rowval = _ring_graph(n, deg, rng)
    E = n * deg
    assert E * 128 > 2 ** 31
    g = gn.GNGraphBatch.from_csc_packed(colptr, rowval, [n])
    p = O.make_block_params(rng, DIMS, DIMS)
    blk = U.block_from_params(gn, p)
    dev = blk.edgefn.weight.device
    tg = torch.Generator(device=dev); tg.manual_seed(63)
    ef = torch.rand((1, E, 128), generator=tg, device=dev) * 2 - 1
    nf = torch.rand((1, n, 64), generator=tg, device=dev) * 2 - 1
    gf = torch.rand((1, 1, 32), generator=tg, device=dev)
    y = blk(gn.NT(g, ef.permute(2, 1, 0), nf.permute(2, 1, 0), gf.permute(2, 1, 0)))
    torch.cuda.synchronize()
    eo, no, go = (t.permute(2, 1, 0)[0] for t in (y.ef, y.nf, y.gf))  # [E][128], [n][64], [1][32]
    assert eo.numel() > 2 ** 31
    We, be = p["We"].astype(np.float64), p["be"].astype(np.float64)
    Wn, bn = p["Wn"].astype(np.float64), p["bn"].astype(np.float64)
    gfv = gf[0, 0].double().cpu().numpy()
    dst = lambda e: e // deg

    def edge_rows(ids):
        ids_t = torch.as_tensor(ids, device=dev)
        x = np.concatenate([ef[0, ids_t].double().cpu().numpy(), nf[0, torch.as_tensor(rowval[ids], device=dev)].double().cpu().numpy(),
                            nf[0, torch.as_tensor(dst(ids), device=dev)].double().cpu().numpy(), np.broadcast_to(gfv, (len(ids), 32))], axis=1)
        return x @ We.T + be, np.abs(x) @ np.abs(We).T + np.abs(be)
    ids = np.unique(np.concatenate([np.arange(300), np.arange(E - 300, E), rng.integers(0, E, 4000), rng.integers(2 ** 31 // 128 - 64, 2 ** 31 // 128 + 64, 64)]))
    ref, sc = edge_rows(ids)
    got = eo[torch.as_tensor(ids, device=dev)].double().cpu().numpy()
    assert (np.abs(got - ref) <= 1e-5 * sc).all(), float((np.abs(got - ref) / sc).max())
    nodes = np.unique(np.concatenate([np.arange(100), np.arange(n - 100, n), rng.integers(0, n, 600)]))
    in_e = (nodes[:, None] * deg + np.arange(deg)[None, :]).reshape(-1)  # the in-edges of node j are edges j*deg .. j*deg + deg - 1
    he, se = edge_rows(in_e)
    agg, sagg = he.reshape(len(nodes), deg, 128).sum(1), se.reshape(len(nodes), deg, 128).sum(1)
    xn = np.concatenate([agg, nf[0, torch.as_tensor(nodes, device=dev)].double().cpu().numpy(), np.broadcast_to(gfv, (len(nodes), 32))], axis=1)
    sn = np.concatenate([sagg, np.abs(xn[:, 128:192]), np.broadcast_to(np.abs(gfv), (len(nodes), 32))], axis=1)
    refn, scn = xn @ Wn.T + bn, sn @ np.abs(Wn).T + np.abs(bn)
    gotn = no[torch.as_tensor(nodes, device=dev)].double().cpu().numpy()
    assert (np.abs(gotn - refn) <= 1e-5 * scn).all(), float((np.abs(gotn - refn) / scn).max())
    xg = np.concatenate([eo.double().sum(0).cpu().numpy(), no.double().sum(0).cpu().numpy(), gfv])
    sg = np.concatenate([eo.double().abs().sum(0).cpu().numpy(), no.double().abs().sum(0).cpu().numpy(), np.abs(gfv)])
    refg = p["Wg"].astype(np.float64) @ xg + p["bg"]
    scg = np.abs(p["Wg"].astype(np.float64)) @ sg + np.abs(p["bg"])
    assert (np.abs(go[0].double().cpu().numpy() - refg) <= 1e-5 * scg).all()


def test_more_than_2_to_31_elements_through_the_narrow_path(gn):
    """72M edges x 32 = 2.3e9 elements of ef on the fused narrow kernel (widths (32, 4, 0) => (3, 4, 5)): sampled rows against float64, the
    last edges (element offsets beyond 2^31) included."""
    import torch
    if _free_gb() < 30:
        pytest.skip("needs ~15 GB of device memory")
    rng = np.random.default_rng(6400)
    n, deg = 6_000_000, 12
    colptr, rowval = _ring_graph(n, deg, rng)
    E = n * deg
    assert E * 32 > 2 ** 31
    g = gn.GNGraphBatch.from_csc_packed(colptr, rowval, [n])
    din, dout = (32, 4, 0), (3, 4, 5)
    p = O.make_block_params(rng, din, dout)
    blk = U.block_from_params(gn, p)
    dev = blk.edgefn.weight.device
    tg = torch.Generator(device=dev); tg.manual_seed(64)
    ef = torch.rand((1, E, 32), generator=tg, device=dev) * 2 - 1
    nf = torch.rand((1, n, 4), generator=tg, device=dev) * 2 - 1
    y = blk(gn.NT(g, ef.permute(2, 1, 0), nf.permute(2, 1, 0), None))
    torch.cuda.synchronize()
    eo, no, go = (t.permute(2, 1, 0)[0] for t in (y.ef, y.nf, y.gf))
    We, be = p["We"].astype(np.float64), p["be"].astype(np.float64)
    Wn, bn = p["Wn"].astype(np.float64), p["bn"].astype(np.float64)

    def edge_rows(ids):
        x = np.concatenate([ef[0, torch.as_tensor(ids, device=dev)].double().cpu().numpy(), nf[0, torch.as_tensor(rowval[ids], device=dev)].double().cpu().numpy(),
                            nf[0, torch.as_tensor(ids // deg, device=dev)].double().cpu().numpy()], axis=1)
        return x @ We.T + be, np.abs(x) @ np.abs(We).T + np.abs(be)
    ids = np.unique(np.concatenate([np.arange(300), np.arange(E - 300, E), rng.integers(0, E, 4000), rng.integers(2 ** 31 // 32 - 64, 2 ** 31 // 32 + 64, 64)]))
    ref, sc = edge_rows(ids)
    got = eo[torch.as_tensor(ids, device=dev)].double().cpu().numpy()
    assert (np.abs(got - ref) <= 1e-5 * sc).all(), float((np.abs(got - ref) / sc).max())
    nodes = np.unique(np.concatenate([np.arange(100), np.arange(n - 100, n), rng.integers(0, n, 600)]))
    in_e = (nodes[:, None] * deg + np.arange(deg)[None, :]).reshape(-1)
    he, se = edge_rows(in_e)
    agg, sagg = he.reshape(len(nodes), deg, 3).sum(1), se.reshape(len(nodes), deg, 3).sum(1)
    xn = np.concatenate([agg, nf[0, torch.as_tensor(nodes, device=dev)].double().cpu().numpy()], axis=1)
    sn = np.concatenate([sagg, np.abs(xn[:, 3:])], axis=1)
    refn, scn = xn @ Wn.T + bn, sn @ np.abs(Wn).T + np.abs(bn)
    gotn = no[torch.as_tensor(nodes, device=dev)].double().cpu().numpy()
    assert (np.abs(gotn - refn) <= 1e-5 * scn).all(), float((np.abs(gotn - refn) / scn).max())
    xg = np.concatenate([eo.double().sum(0).cpu().numpy(), no.double().sum(0).cpu().numpy()])
    sg = np.concatenate([eo.double().abs().sum(0).cpu().numpy(), no.double().abs().sum(0).cpu().numpy()])
    refg = p["Wg"].astype(np.float64) @ xg + p["bg"]
    scg = np.abs(p["Wg"].astype(np.float64)) @ sg + np.abs(p["bg"])
    assert (np.abs(go[0].double().cpu().numpy() - refg) <= 1e-5 * scg).all()

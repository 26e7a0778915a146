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


def assert_close(got, ref, scale, what=""):
    if ref is None:
        assert got is None, f"{what}: expected nothing"
        return
    assert got is not None and got.shape == ref.shape, f"{what}: shape {None if got is None else got.shape} vs {ref.shape}"
    err = np.abs(got.astype(np.float64) - ref)
    bad = err > RTOL * scale + 1e-30
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} outside 1e-5·scale; worst ratio {np.max(err / (scale + 1e-30)):.3e}"

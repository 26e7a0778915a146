"""
gn_oracle — CPU restatement of GraphNets.jl's GNBlock / GNCore forward pass.

*** TEST INFRASTRUCTURE ONLY. ***  Nothing under `oracle/` is part of the product.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and only as the
checker.  The product path (`graphnets.jl_amd/`, `libgnx.so`) never calls into this directory.

*** PARITY UNPINNED (absolute numerics). ***  The reference is pure Julia (Flux 0.14 / NNlib); Julia is not
installed here or on the GPU box, so the reference can be neither compiled nor imported, and its test-suite
(`/root/reference/test/runtests.jl`) holds no numerical golden values for this path.  What IS pinned:
  * the index semantics (src = row, dst = column, column-major edge slots, aggregation at dst), against the
    known-answer matrices the reference's tests hold (`test/runtests.jl:487-508`, `:659-680`), transcribed
    as data in `tests/golden/reference_known_answers.json`;
  * the relational properties the reference tests assert (batch invariance `:62-116`, batch∘unbatch identity
    `:328-390`, output shapes / `nothing` handling `:118-326`).
The arithmetic of `Dense`, `LayerNorm`, `batched_mul` lives in un-vendored Flux/NNlib/BLAS (Project.toml:6-15,
no Manifest); their assumed semantics are restated below from the published definitions.

Two forms are provided, both float64:
  (i)  DENSE form  — a line-by-line restatement of the reference's padded N^2 edge grid + one-hot
       "broadcaster" batched matmuls.  Only usable for tiny graphs.  Arrays use the reference's Julia
       shapes `(D, T, B)`; "column-major flatten" is spelled `order='F'`.
  (ii) SPARSE form — the same mathematics on packed CSC data, arrays in the C-ABI layout `[R][T][D]`
       (byte-identical to Julia's column-major `(D, T, R)`).
`tests/test_oracle.py` proves (i) == (ii) and checks both against the reference's fixtures.
"""
from __future__ import annotations

import numpy as np

F64 = np.float64

# ----------------------------------------------------------------------------------------------------------
# Activations (Flux/NNlib definitions)
# ----------------------------------------------------------------------------------------------------------
ACT_IDENTITY, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_GELU = 0, 1, 2, 3, 4


def apply_act(x, act):
    if act == ACT_IDENTITY:
        return x
    if act == ACT_RELU:
        return np.maximum(x, 0.0)
    if act == ACT_TANH:
        return np.tanh(x)
    if act == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-x))
    if act == ACT_GELU:  # NNlib.gelu = tanh approximation
        return 0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))
    raise ValueError(f"unknown activation {act}")


# ----------------------------------------------------------------------------------------------------------
# Parameters
# ----------------------------------------------------------------------------------------------------------
def glorot_uniform(rng, out_dim, in_dim):
    """Flux.glorot_uniform: U(-s, s), s = sqrt(6 / (fan_in + fan_out)); weight is (out, in)."""
    if out_dim == 0 or in_dim == 0:
        return np.zeros((out_dim, in_dim), dtype=np.float32)
    s = np.sqrt(6.0 / (in_dim + out_dim))
    return rng.uniform(-s, s, size=(out_dim, in_dim)).astype(np.float32)


def block_input_widths(in_dims, out_dims):
    """gnblock.jl:52-54."""
    de, dn, dg = in_dims
    oe, on, og = out_dims
    return de + 2 * dn + dg, dn + oe + dg, on + oe + dg


def make_block_params(rng, in_dims, out_dims, random_bias=True, act=(0, 0, 0)):
    """GNBlock(in => out) (gnblock.jl:47-61): three single-Dense chains.  Flux's default bias is zero; tests
    draw U(-0.1, 0.1) biases so the bias path is exercised."""
    assert any(d > 0 for d in in_dims) and any(d > 0 for d in out_dims)  # gnblock.jl:48-49
    ke, kn, kg = block_input_widths(in_dims, out_dims)
    oe, on, og = out_dims

    def bias(n):
        return (rng.uniform(-0.1, 0.1, size=n) if random_bias else np.zeros(n)).astype(np.float32)

    return dict(
        in_dims=tuple(in_dims), out_dims=tuple(out_dims),
        We=glorot_uniform(rng, oe, ke), be=bias(oe), act_e=act[0],
        Wn=glorot_uniform(rng, on, kn), bn=bias(on), act_n=act[1],
        Wg=glorot_uniform(rng, og, kg), bg=bias(og), act_g=act[2],
    )


def make_core_params(rng, dims, random_bias=True, eps=1e-5, eps_mode=0):
    """GNCore(dims) (gncore.jl:46-54): block dims=>dims, GNFeedForward (gnfeedforward.jl:17-31),
    two GNGraphNorm (gngraphnorm.jl:9-17).  LayerNorm affine params are drawn randomly (Flux init is 1/0)."""
    assert all(d > 0 for d in dims)  # gnfeedforward.jl:18, gngraphnorm.jl:10
    p = dict(dims=tuple(dims), block=make_block_params(rng, dims, dims, random_bias), eps=eps, eps_mode=eps_mode)
    for tag, d in zip("eng", dims):
        for ln in ("ln1", "ln2"):
            p[f"{ln}_{tag}_gamma"] = rng.uniform(0.5, 1.5, size=d).astype(np.float32)
            p[f"{ln}_{tag}_beta"] = rng.uniform(-0.1, 0.1, size=d).astype(np.float32)
        p[f"ff_{tag}_W1"] = glorot_uniform(rng, 4 * d, d)
        p[f"ff_{tag}_b1"] = (rng.uniform(-0.1, 0.1, size=4 * d) if random_bias else np.zeros(4 * d)).astype(np.float32)
        p[f"ff_{tag}_W2"] = glorot_uniform(rng, d, 4 * d)
        p[f"ff_{tag}_b2"] = (rng.uniform(-0.1, 0.1, size=d) if random_bias else np.zeros(d)).astype(np.float32)
    return p


# ----------------------------------------------------------------------------------------------------------
# (i) DENSE form — literal restatement of the reference
# ----------------------------------------------------------------------------------------------------------
def padadjmats(adj_mats):
    """pad.jl:1-10 → (PN, PN, B)."""
    B = len(adj_mats)
    PN = max(a.shape[0] for a in adj_mats)
    out = np.zeros((PN, PN, B), dtype=F64)
    for b, a in enumerate(adj_mats):
        n = a.shape[0]
        out[:n, :n, b] = a
    return out


def getnode2edgebroadcaster(padded, transpose=False):
    """gngraphbatch.jl:197-211.  idx[i, j] = i (src) or j (dst, `transpose`), 1-based; one-hot of the masked
    index scattered into column `slot` of a (PN, PN^2) matrix."""
    PN, _, B = padded.shape
    idx = np.repeat(np.arange(1, PN + 1)[:, None], PN, axis=1).astype(F64)  # repeat(1:PN, 1, PN)
    if transpose:
        idx = idx.T
    dst = np.zeros((PN, PN * PN, B), dtype=F64)
    for b in range(B):
        flat = (padded[:, :, b] * idx).flatten(order="F")
        active_idx = np.nonzero(flat)[0]
        active = flat[active_idx].astype(np.int64)
        for slot, node in zip(active_idx, active):  # onehotbatch + scatter!(+)
            dst[node - 1, slot, b] += 1.0
    return dst


def getgraph2edgebroadcaster(padded):
    """gngraphbatch.jl:183-192 → (1, PN^2, B)."""
    PN, _, B = padded.shape
    m = np.zeros((1, PN * PN, B), dtype=F64)
    for b in range(B):
        m[0, :, b] = (padded[:, :, b].flatten(order="F") == 1.0)
    return m


def getedge2nodebroadcaster(padded):
    """gngraphbatch.jl:158-170 → (PN^2, PN, B)."""
    PN, _, B = padded.shape
    m = np.zeros((PN * PN, PN, B), dtype=F64)
    for b in range(B):
        for col in range(PN):
            m[PN * col: PN * col + PN, col, b] = padded[:, col, b]
    return m


def getgraph2nodebroadcaster(adj_mats, PN):
    """gngraphbatch.jl:172-181 → (1, PN, B)."""
    m = np.zeros((1, PN, len(adj_mats)), dtype=F64)
    for b, a in enumerate(adj_mats):
        m[0, : a.shape[0], b] = 1.0
    return m


def getedge2graphbroadcaster(padded):
    """gngraphbatch.jl:136-146 → (PN^2, 1, B)."""
    PN, _, B = padded.shape
    m = np.zeros((PN * PN, 1, B), dtype=F64)
    for b in range(B):
        m[:, 0, b] = padded[:, :, b].flatten(order="F")
    return m


def getnode2graphbroadcaster(adj_mats, PN):
    """gngraphbatch.jl:148-156 → (PN, 1, B)."""
    m = np.zeros((PN, 1, len(adj_mats)), dtype=F64)
    for b, a in enumerate(adj_mats):
        m[: a.shape[0], 0, b] = 1.0
    return m


class DenseGraphBatch:
    """GNGraphBatch (gngraphbatch.jl:1-54) without the edge-collapse members."""

    def __init__(self, adj_mats):
        self.adj_mats = [np.asarray(a, dtype=F64) for a in adj_mats]
        self.padded_adj_mats = padadjmats(self.adj_mats)
        PN = self.padded_adj_mats.shape[0]
        self.node_block_size = PN
        self.edge_block_size = PN * PN
        self.srcnode2edge = getnode2edgebroadcaster(self.padded_adj_mats)
        self.dstnode2edge = getnode2edgebroadcaster(self.padded_adj_mats, transpose=True)
        self.graph2edge = getgraph2edgebroadcaster(self.padded_adj_mats)
        self.edge2node = getedge2nodebroadcaster(self.padded_adj_mats)
        self.graph2node = getgraph2nodebroadcaster(self.adj_mats, PN)
        self.edge2graph = getedge2graphbroadcaster(self.padded_adj_mats)
        self.node2graph = getnode2graphbroadcaster(self.adj_mats, PN)
        # gngraphbatch.jl:113-134
        B = len(self.adj_mats)
        self.flat_node_unpadder = np.zeros(B * PN, dtype=bool)
        self.flat_edge_unpadder = np.zeros(B * PN * PN, dtype=bool)
        for b, a in enumerate(self.adj_mats):
            self.flat_node_unpadder[b * PN: b * PN + a.shape[0]] = True
            self.flat_edge_unpadder[b * PN * PN: (b + 1) * PN * PN] = self.padded_adj_mats[:, :, b].flatten(order="F") == 1.0


def getlowertriangularcoords(n):
    """gngraphbatch.jl:56-58: CartesianIndices of an n x n matrix with i >= j, in column-major order (0-based pairs)."""
    return [(i, j) for j in range(n) for i in range(n) if i >= j]


def getedgecollapser(n):
    """gngraphbatch.jl:67-82: (n^2, n(n+1)/2) matrix; column of coordinate (i, j) has a one at slots (i,j) and (j,i) of the
    column-major n x n grid — a two on the diagonal, since the reference adds both."""
    coords = getlowertriangularcoords(n)
    m = np.zeros((n * n, len(coords)), dtype=F64)
    for c, (i, j) in enumerate(coords):
        m[i + n * j, c] += 1
        m[j + n * i, c] += 1
    return m


def getcollapsededgeidxs(padded):
    """gngraphbatch.jl:60-65: per graph, positions (0-based) within the lower-triangle coordinate list whose adjacency entry is one."""
    coords = getlowertriangularcoords(padded.shape[0])
    return [np.array([c for c, (i, j) in enumerate(coords) if padded[i, j, b] == 1.0], dtype=np.int64) for b in range(padded.shape[2])]


def collapsef_dense(x):
    """gngraphbatch.jl:83-85: batched_mul(graph.ef, graph.graphs.edge_collapser) / Float32(2) on the padded (DE, PN^2, B) array."""
    return batched_mul(x["ef"], getedgecollapser(x["graphs"].node_block_size)) / 2.0


def unpaddedcollapsedef_dense(x):
    """gngraphbatch.jl:87-107: per batch element, the collapsed columns of the real lower-triangle edges."""
    c = collapsef_dense(x)
    idxs = getcollapsededgeidxs(x["graphs"].padded_adj_mats)
    if len(idxs) == 1:  # shared adjacency: one index list for every batch element
        return [c[:, idxs[0], b] for b in range(c.shape[2])]
    return [c[:, idxs[b], b] for b in range(c.shape[2])]


def batched_mul(A, Bm):
    """NNlib.batched_mul: C[:,:,k] = A[:,:,k] * B[:,:,k]; a size-1 batch on either side broadcasts."""
    if A.ndim == 2:
        A = A[:, :, None]
    if Bm.ndim == 2:
        Bm = Bm[:, :, None]
    nb = max(A.shape[2], Bm.shape[2])
    assert A.shape[2] in (1, nb) and Bm.shape[2] in (1, nb)
    out = np.zeros((A.shape[0], Bm.shape[1], nb), dtype=F64)
    for k in range(nb):
        out[:, :, k] = A[:, :, k if A.shape[2] > 1 else 0] @ Bm[:, :, k if Bm.shape[2] > 1 else 0]
    return out


def padef_shared(adj, ef):
    """pad.jl:26-39: packed (DE, E, B) scattered into (DE, N^2, B) at the column-major positions of the ones."""
    n = adj.shape[0]
    edge_idx = np.nonzero(np.asarray(adj).flatten(order="F") == 1)[0]
    out = np.zeros((ef.shape[0], n * n, ef.shape[2]), dtype=F64)
    out[:, edge_idx, :] = ef
    return out


def padef_vector(adj_mats, efs):
    """pad.jl:48-64."""
    padded = padadjmats(adj_mats)
    PN = padded.shape[0]
    out = np.zeros((efs[0].shape[0], PN * PN, len(adj_mats)), dtype=F64)
    for b, ef in enumerate(efs):
        edge_idx = np.nonzero(padded[:, :, b].flatten(order="F") == 1)[0]
        out[:, edge_idx, b] = ef
    return out


def padnf_vector(adj_mats, nfs):
    """pad.jl:14-24."""
    PN = max(a.shape[0] for a in adj_mats)
    out = np.zeros((nfs[0].shape[0], PN, len(adj_mats)), dtype=F64)
    for b, nf in enumerate(nfs):
        out[:, : nf.shape[1], b] = nf
    return out


def batch_dense(graphs, ef, nf, gf):
    """batch (batch.jl:53-64).  `graphs` is one adjacency matrix (shared; ef (DE,E,B), nf (DN,N,B), gf (DG,B))
    or a list (vector mode; lists of (DE,E_g), (DN,N_g), (DG,))."""
    assert not (ef is None and nf is None and gf is None)  # batch.jl:56
    if isinstance(graphs, (list, tuple)):
        adj_mats = [np.asarray(a) for a in graphs]
        g = DenseGraphBatch(adj_mats)
        bef = None if ef is None else padef_vector(adj_mats, [np.asarray(x, dtype=F64) for x in ef])
        bnf = None if nf is None else padnf_vector(adj_mats, [np.asarray(x, dtype=F64) for x in nf])
        bgf = None if gf is None else np.stack([np.asarray(x, dtype=F64) for x in gf], axis=1)[:, None, :]  # pad.jl:67
    else:
        adj = np.asarray(graphs)
        g = DenseGraphBatch([adj])
        bef = None if ef is None else padef_shared(adj, np.asarray(ef, dtype=F64))
        bnf = None if nf is None else np.asarray(nf, dtype=F64)  # pad.jl:12
        bgf = None if gf is None else np.asarray(gf, dtype=F64)[:, None, :]  # pad.jl:66
    return dict(graphs=g, ef=bef, nf=bnf, gf=bgf)


def dense_layer(W, b, act, x):
    """Flux.Dense on an N-D array: reshape to (in, :), y = act.(W*x .+ b), reshape back."""
    W = np.asarray(W, dtype=F64)
    b = np.asarray(b, dtype=F64)
    sh = x.shape
    y = W @ x.reshape(sh[0], -1, order="F") + b[:, None]
    return apply_act(y, act).reshape((W.shape[0],) + sh[1:], order="F")


def _vcat(parts):
    return np.concatenate([p for p in parts if p is not None], axis=0)


def getedgefninput_dense(g, ef, nf, gf):
    """edgefninput.jl:1-47 — segment order ef, nf⊗src, nf⊗dst, gf⊗g2e; absent inputs drop their segments."""
    parts = [ef]
    if nf is not None:
        parts += [batched_mul(nf, g.srcnode2edge), batched_mul(nf, g.dstnode2edge)]
    if gf is not None:
        parts += [batched_mul(gf, g.graph2edge)]
    return _vcat(parts)


def getnodefninput_dense(g, ef, nf, gf):
    """nodefninput.jl:1-24 — order agg, nf, gf."""
    parts = [batched_mul(ef, g.edge2node), nf]
    if gf is not None:
        parts += [batched_mul(gf, g.graph2node)]
    return _vcat(parts)


def getgraphfninput_dense(g, ef, nf, gf):
    """graphfninput.jl:1-13 — order edges, nodes, gf."""
    return _vcat([batched_mul(ef, g.edge2graph), batched_mul(nf, g.node2graph), gf])


def block_forward_dense(p, x):
    """(m::GNBlock)(x) (gnblock.jl:63-69) on the padded batched tuple; zero-row outputs → None (:71-78)."""
    g, ef, nf, gf = x["graphs"], x["ef"], x["nf"], x["gf"]
    h_ef = dense_layer(p["We"], p["be"], p["act_e"], getedgefninput_dense(g, ef, nf, gf))
    h_nf = dense_layer(p["Wn"], p["bn"], p["act_n"], getnodefninput_dense(g, h_ef, nf, gf))
    h_gf = dense_layer(p["Wg"], p["bg"], p["act_g"], getgraphfninput_dense(g, h_ef, h_nf, gf))
    z = lambda a: None if a.shape[0] == 0 else a
    return dict(graphs=g, ef=z(h_ef), nf=z(h_nf), gf=z(h_gf))


def layernorm(x, gamma, beta, eps=1e-5, eps_mode=0, axis=0):
    """Flux.LayerNorm(d) over the feature dim.  eps_mode 0: Flux 0.14 `normalise` = (x-μ)/(σ+ε) with the
    uncorrected std; eps_mode 1: (x-μ)/sqrt(σ²+ε).  Then γ .* x̂ .+ β."""
    mu = x.mean(axis=axis, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=axis, keepdims=True)
    xhat = (x - mu) / (np.sqrt(var) + eps) if eps_mode == 0 else (x - mu) / np.sqrt(var + eps)
    shape = [1] * x.ndim
    shape[axis] = -1
    return xhat * np.asarray(gamma, dtype=F64).reshape(shape) + np.asarray(beta, dtype=F64).reshape(shape)


def core_forward_dense(p, x):
    """(m::GNCore)(x) = x + block(gn1(x)) + ffwd(gn2(x)) (gncore.jl:56-68)."""
    g = x["graphs"]
    ln = lambda which: dict(
        graphs=g,
        **{k: layernorm(x[k], p[f"{which}_{t}_gamma"], p[f"{which}_{t}_beta"], p["eps"], p["eps_mode"])
           for k, t in (("ef", "e"), ("nf", "n"), ("gf", "g"))})
    blk = block_forward_dense(p["block"], ln("ln1"))
    x2 = ln("ln2")
    out = dict(graphs=g)
    for k, t in (("ef", "e"), ("nf", "n"), ("gf", "g")):
        h = dense_layer(p[f"ff_{t}_W1"], p[f"ff_{t}_b1"], ACT_RELU, x2[k])
        ff = dense_layer(p[f"ff_{t}_W2"], p[f"ff_{t}_b2"], ACT_IDENTITY, h)  # Dropout: identity in test mode
        out[k] = x[k] + blk[k] + ff
    return out


def unbatch_dense(x):
    """unbatch (unbatch.jl:6-39, unpad.jl:1-25)."""
    g, ef, nf, gf = x["graphs"], x["ef"], x["nf"], x["gf"]
    if len(g.adj_mats) == 1:
        adj = g.adj_mats[0]
        idx = np.nonzero(adj.flatten(order="F") == 1)[0]
        return dict(graphs=adj, ef=None if ef is None else ef[:, idx, :], nf=nf,
                    gf=None if gf is None else gf.reshape(gf.shape[0], -1, order="F"))
    efs = nfs = gfs = None
    if ef is not None:
        efs = [ef[:, np.nonzero(g.padded_adj_mats[:, :, b].flatten(order="F") == 1)[0], b] for b in range(len(g.adj_mats))]
    if nf is not None:
        nfs = [nf[:, : a.shape[0], b] for b, a in enumerate(g.adj_mats)]
    if gf is not None:
        gfs = [gf[:, 0, b] for b in range(len(g.adj_mats))]
    return dict(graphs=g.adj_mats, ef=efs, nf=nfs, gf=gfs)


# ----------------------------------------------------------------------------------------------------------
# (ii) SPARSE form — packed CSC, C-ABI layout [R][T][D]
# ----------------------------------------------------------------------------------------------------------
def csc_from_adj(adj_mats):
    """Edge k of graph g ↔ k-th `1` of vec(A_g) column-major (pad.jl:30): sorted by dst (column) then src (row).
    Returns global colptr[N+1], rowval[E] (global src node id), node_off[G+1], edge_off[G+1] (int64)."""
    node_off, edge_off, colptr, rowval = [0], [0], [0], []
    for a in adj_mats:
        a = np.asarray(a)
        n = a.shape[0]
        assert a.shape == (n, n)
        base = node_off[-1]
        for j in range(n):
            rows = np.nonzero(a[:, j] == 1)[0]
            rowval.extend((rows + base).tolist())
            colptr.append(colptr[-1] + len(rows))
        node_off.append(base + n)
        edge_off.append(colptr[-1])
    return (np.asarray(colptr, dtype=np.int64), np.asarray(rowval, dtype=np.int64),
            np.asarray(node_off, dtype=np.int64), np.asarray(edge_off, dtype=np.int64))


def _dense_rows(W, b, act, X):
    """Dense applied to row-major rows: X (T, K) → (T, out)."""
    return apply_act(X @ np.asarray(W, dtype=F64).T + np.asarray(b, dtype=F64)[None, :], act)


_CHUNK = 1 << 17  # rows per chunk of the row-wise stages: bounds the temporaries at BASELINE's 1M-edge / 128-wide sizes


def _segsum(v, seg, n):
    """out[seg[k]] += v[k] with np.bincount per column (vectorised; np.add.at is an order of magnitude slower)."""
    out = np.empty((n, v.shape[1]), dtype=v.dtype)
    for c in range(v.shape[1]):
        out[:, c] = np.bincount(seg, weights=v[:, c], minlength=n)
    return out


def _scale_rows(S, W, b):
    """Error scale of a Dense: S·|W|ᵀ + |b| in float32 (a bound only needs a few digits; sgemm is what makes the
    full-size scales affordable)."""
    return S.astype(np.float32, copy=False) @ np.abs(np.asarray(W, dtype=np.float32)).T + np.abs(np.asarray(b, dtype=np.float32))[None, :]


def block_forward_sparse(p, csc, ef, nf, gf, return_scale=False, in_scale=None):
    """SURVEY Appendix A on packed data.  ef (R,E,DE) | None, nf (R,N,DN) | None, gf (R,G,DG) | None.
    Returns (ef', nf', gf') with zero-width outputs as None.  With `return_scale` also returns, per output, the
    magnitude bound |W|·S + |b| that the parity tests use as the tolerance scale: S = |x| for exact inputs, or the
    `in_scale` triple (each >= |input|, same shapes) when the inputs are themselves computed results — the bound then
    also covers the first-order propagation of their error (a perturbation δ <= c·eps·S of the inputs moves the outputs
    by at most c·eps·(|W|·S)).  This is a WORST-CASE bound (every rounding error aligned): right for one layer, vacuous for a
    deep chain through million-term sums and LayerNorm's 1/σ — chains are checked layer by layer from the float32-rounded
    oracle input of each layer instead (tests/test_gpu_fullsize.py).  (A quadrature / probabilistic scale was tried and is
    wrong here: gf and the gathered node rows are shared by many edges, their errors are fully correlated in the next sum.)
    The edge stage runs in row chunks (1M edges x 288 inputs would be 2.3 GB at once)."""
    colptr, rowval, node_off, edge_off = csc
    N, E, G = len(colptr) - 1, len(rowval), len(node_off) - 1
    R = next(a.shape[0] for a in (ef, nf, gf) if a is not None)
    dst = np.repeat(np.arange(N), np.diff(colptr))
    node_graph = np.repeat(np.arange(G), np.diff(node_off))
    edge_graph = np.repeat(np.arange(G), np.diff(edge_off))
    oe, on, og = p["out_dims"]
    We, Wn, Wg = (np.asarray(p[k], dtype=F64) for k in ("We", "Wn", "Wg"))
    want = return_scale
    sin = in_scale if in_scale is not None else (None, None, None)
    outs, scales = ([], [], []), ([], [], [])
    for r in range(R):
        efr = None if ef is None else ef[r]
        nfr = None if nf is None else np.asarray(nf[r], dtype=F64)
        gfr = None if gf is None else np.asarray(gf[r], dtype=F64)
        s_ef = None if (ef is None or not want) else (np.abs(efr) if sin[0] is None else sin[0][r])
        s_nf = None if (nf is None or not want) else (np.abs(nfr) if sin[1] is None else sin[1][r]).astype(np.float32)
        s_gf = None if (gf is None or not want) else (np.abs(gfr) if sin[2] is None else sin[2][r]).astype(np.float32)
        he = np.empty((E, oe), dtype=F64)
        se = np.empty((E, oe), dtype=np.float32) if want else None
        for c0 in range(0, max(E, 1), _CHUNK):
            c1 = min(E, c0 + _CHUNK)
            if c1 <= c0:
                break
            parts, sparts = [], []
            if ef is not None:
                parts.append(np.asarray(efr[c0:c1], dtype=F64))
                if want:
                    sparts.append(np.asarray(s_ef[c0:c1], dtype=np.float32))
            if nf is not None:
                parts += [nfr[rowval[c0:c1]], nfr[dst[c0:c1]]]
                if want:
                    sparts += [s_nf[rowval[c0:c1]], s_nf[dst[c0:c1]]]
            if gf is not None:
                parts.append(gfr[edge_graph[c0:c1]])
                if want:
                    sparts.append(s_gf[edge_graph[c0:c1]])
            Xe = np.concatenate(parts, axis=1) if parts else np.zeros((c1 - c0, 0))
            he[c0:c1] = _dense_rows(We, p["be"], p["act_e"], Xe)
            if want:
                se[c0:c1] = _scale_rows(np.concatenate(sparts, axis=1), We, p["be"])

        parts = [_segsum(he, dst, N)]
        sparts = [_segsum(se, dst, N).astype(np.float32)] if want else None
        if nf is not None:
            parts.append(nfr)
            if want:
                sparts.append(s_nf)
        if gf is not None:
            parts.append(gfr[node_graph])
            if want:
                sparts.append(s_gf[node_graph])
        hn = _dense_rows(Wn, p["bn"], p["act_n"], np.concatenate(parts, axis=1))
        sn = _scale_rows(np.concatenate(sparts, axis=1), Wn, p["bn"]) if want else None

        parts = [_segsum(he, edge_graph, G), _segsum(hn, node_graph, G)]
        sparts = [_segsum(se, edge_graph, G).astype(np.float32), _segsum(sn, node_graph, G).astype(np.float32)] if want else None
        if gf is not None:
            parts.append(gfr)
            if want:
                sparts.append(s_gf)
        hg = _dense_rows(Wg, p["bg"], p["act_g"], np.concatenate(parts, axis=1))
        sg = _scale_rows(np.concatenate(sparts, axis=1), Wg, p["bg"]) if want else None
        for lst, v in zip(outs, (he, hn, hg)):
            lst.append(v)
        for lst, v in zip(scales, (se, sn, sg)):
            lst.append(v)
    res = tuple(None if d == 0 else np.stack(lst) for d, lst in zip((oe, on, og), outs))
    if return_scale:
        return res, tuple(None if d == 0 else np.stack(lst) for d, lst in zip((oe, on, og), scales))
    return res


def layernorm_scale(x, s, gamma, beta, eps=1e-5, eps_mode=0):
    """Error scale of LayerNorm(x) over the last axis for an input that carries the error scale s >= |x| (first order,
    worst case): δμ <= mean(s), δ(x-μ) <= s + mean(s) =: sc, δσ <= sqrt(mean(sc²)) =: q (Cauchy-Schwarz on
    mean((x-μ)·δ(x-μ))/σ), hence δx̂ <= (sc + |x̂|·q) / (σ+ε); the affine part adds its own rounding |γ·x̂| + |β|.
    LayerNorm divides by σ: for U[0,1) inputs (σ ≈ 0.29) the scale grows about 3.5x — that is arithmetic, not slack."""
    x = np.asarray(x, dtype=F64)
    s = np.asarray(s, dtype=F64)
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc ** 2).mean(axis=-1, keepdims=True)
    den = (np.sqrt(var) + eps) if eps_mode == 0 else np.sqrt(var + eps)
    xhat = xc / den
    sc = s + s.mean(axis=-1, keepdims=True)
    q = np.sqrt((sc ** 2).mean(axis=-1, keepdims=True))
    g = np.abs(np.asarray(gamma, dtype=F64))
    return (g * ((sc + np.abs(xhat) * q) / den + np.abs(xhat)) + np.abs(np.asarray(beta, dtype=F64))).astype(np.float32)


def _ffn_rows(p, t, x2, s2):
    """FeedForward (gnfeedforward.jl:27-31) on rows, in chunks (the 4d-wide hidden layer of 1M x 128 rows is 4 GB in float64);
    returns (ff, scale | None)."""
    W1, b1, W2, b2 = p[f"ff_{t}_W1"], p[f"ff_{t}_b1"], p[f"ff_{t}_W2"], p[f"ff_{t}_b2"]
    rows = x2.reshape(-1, x2.shape[-1])
    ff = np.empty_like(rows)
    sc = None if s2 is None else np.empty(rows.shape, dtype=np.float32)
    srows = None if s2 is None else s2.reshape(-1, s2.shape[-1])
    for c0 in range(0, rows.shape[0], _CHUNK):
        c1 = min(rows.shape[0], c0 + _CHUNK)
        h = _dense_rows(W1, b1, ACT_RELU, rows[c0:c1])
        ff[c0:c1] = _dense_rows(W2, b2, ACT_IDENTITY, h)  # Dropout: identity in test mode
        if sc is not None:
            sc[c0:c1] = _scale_rows(_scale_rows(srows[c0:c1], W1, b1), W2, b2)
    return ff.reshape(x2.shape), (None if sc is None else sc.reshape(x2.shape))


def core_forward_sparse(p, csc, ef, nf, gf, return_scale=False, in_scale=None):
    """GNCore on packed data (all three inputs required, gncore.jl:61-68): y = x + block(gn1(x)) + ffwd(gn2(x)).
    `return_scale` / `in_scale` as in block_forward_sparse: the scale of every output is |x| (or its incoming scale) + the
    block's scale for inputs carrying gn1's scale + the FeedForward's scale for inputs carrying gn2's scale, so a chain of
    layers can hand each layer's scale to the next (tests: `<= 1e-5 * scale` at every depth)."""
    xs = dict(e=np.asarray(ef, dtype=F64), n=np.asarray(nf, dtype=F64), g=np.asarray(gf, dtype=F64))
    sx = None
    if return_scale:
        sin = in_scale if in_scale is not None else (None, None, None)
        sx = {t: (np.abs(xs[t]).astype(np.float32) if si is None else np.asarray(si, dtype=np.float32)) for t, si in zip("eng", sin)}
    ln = lambda which: {t: layernorm(xs[t], p[f"{which}_{t}_gamma"], p[f"{which}_{t}_beta"], p["eps"], p["eps_mode"], axis=-1)
                        for t in "eng"}
    lns = lambda which: {t: layernorm_scale(xs[t], sx[t], p[f"{which}_{t}_gamma"], p[f"{which}_{t}_beta"], p["eps"], p["eps_mode"])
                         for t in "eng"}
    x1 = ln("ln1")
    if return_scale:
        s1 = lns("ln1")
        blk, sblk = block_forward_sparse(p["block"], csc, x1["e"], x1["n"], x1["g"], return_scale=True, in_scale=(s1["e"], s1["n"], s1["g"]))
        del s1
    else:
        blk, sblk = block_forward_sparse(p["block"], csc, x1["e"], x1["n"], x1["g"]), (None, None, None)
    del x1
    out, scales = [], []
    for i, t in enumerate("eng"):
        x2 = layernorm(xs[t], p[f"ln2_{t}_gamma"], p[f"ln2_{t}_beta"], p["eps"], p["eps_mode"], axis=-1)
        s2 = layernorm_scale(xs[t], sx[t], p[f"ln2_{t}_gamma"], p[f"ln2_{t}_beta"], p["eps"], p["eps_mode"]) if return_scale else None
        ff, sff = _ffn_rows(p, t, x2, s2)
        out.append(xs[t] + blk[i] + ff)
        if return_scale:
            scales.append(sx[t] + sblk[i] + sff)
    if return_scale:
        return tuple(out), tuple(scales)
    return tuple(out)


def make_chain_block_params(rng, in_dims, edge_widths, node_widths, graph_widths, acts=(ACT_RELU, ACT_TANH, ACT_IDENTITY)):
    """GNBlock whose update functions are Chains of Dense layers (gnblock.jl:1-6 allows any Chain; the default is one Dense):
    `*_widths` = output widths of the layers of each chain (empty = that output is `nothing`); hidden layers use `acts[i % 3]`,
    the last layer of a chain is linear (like the reference's default).  A width entry "ln" puts a Flux `LayerNorm(d)` layer value
    there (d = the width in front of it; random gamma / beta): the layer tuple is ("layernorm", gamma, beta)."""
    de, dn, dg = in_dims
    # (a LayerNorm keeps the width in front of it; a chain of LayerNorms alone keeps its input's)
    last = lambda ws, k: next((w for w in reversed(ws) if w != "ln"), k if ws else 0)
    oe = last(edge_widths, de + 2 * dn + dg)
    on = last(node_widths, oe + dn + dg)
    kin = dict(edge=de + 2 * dn + dg, node=oe + dn + dg, graph=oe + on + dg)
    p = dict(in_dims=tuple(in_dims))
    for name, widths in (("edge", edge_widths), ("node", node_widths), ("graph", graph_widths)):
        layers, k = [], kin[name]
        for i, w in enumerate(widths):
            if w == "ln":
                layers.append(("layernorm", rng.uniform(0.5, 1.5, size=k).astype(np.float32), rng.uniform(-0.2, 0.2, size=k).astype(np.float32)))
                continue
            act = acts[i % len(acts)] if i + 1 < len(widths) else ACT_IDENTITY
            layers.append((glorot_uniform(rng, w, k), rng.uniform(-0.1, 0.1, size=w).astype(np.float32), act))
            k = w
        p[name] = layers
    return p


def chain_block_forward_sparse(p, csc, ef, nf, gf, return_scale=False):
    """(m::GNBlock)(x) (gnblock.jl:63-69) with Chain update functions, on packed data: every chain is applied to the same
    function inputs as in block_forward_sparse (edgefninput.jl / nodefninput.jl / graphfninput.jl), layer after layer.
    Scales (worst case, as block_forward_sparse): |W|·S + |b| through every layer."""
    colptr, rowval, node_off, edge_off = csc
    N, E, G = len(colptr) - 1, len(rowval), len(node_off) - 1
    R = next(a.shape[0] for a in (ef, nf, gf) if a is not None)
    dst = np.repeat(np.arange(N), np.diff(colptr))
    node_graph = np.repeat(np.arange(G), np.diff(node_off))
    edge_graph = np.repeat(np.arange(G), np.diff(edge_off))

    def run(layers, X, S):
        for W, b, act in layers:
            if isinstance(W, str):  # ("layernorm", gamma, beta): a Flux LayerNorm(d) layer value between the Dense layers (gnblock.jl:1-6 admits any Chain)
                S = layernorm_scale(X, S, b, act)
                X = layernorm(X, b, act, axis=-1)
                continue
            S = np.abs(S) @ np.abs(np.asarray(W, dtype=F64)).T + np.abs(np.asarray(b, dtype=F64))[None, :]
            X = _dense_rows(W, b, act, X)
        return X, S

    outs, scales = ([], [], []), ([], [], [])
    for r in range(R):
        parts = []
        if ef is not None:
            parts.append(np.asarray(ef[r], dtype=F64))
        if nf is not None:
            nfr = np.asarray(nf[r], dtype=F64)
            parts += [nfr[rowval], nfr[dst]]
        if gf is not None:
            gfr = np.asarray(gf[r], dtype=F64)
            parts.append(gfr[edge_graph])
        Xe = np.concatenate(parts, axis=1)
        he, se = run(p["edge"], Xe, np.abs(Xe)) if p["edge"] else (None, None)
        hn = sn = hg = sg = None
        if p["node"]:
            # (a chain without layers = a zero-width output: an empty segment of the next input, as the reference's 0-row arrays, gnblock.jl:63-69)
            parts, sparts = ([_segsum(he, dst, N)], [_segsum(se, dst, N)]) if he is not None else ([], [])
            if nf is not None:
                parts.append(nfr); sparts.append(np.abs(nfr))
            if gf is not None:
                parts.append(gfr[node_graph]); sparts.append(np.abs(gfr[node_graph]))
            hn, sn = run(p["node"], np.concatenate(parts, axis=1), np.concatenate(sparts, axis=1))
        if p["graph"]:
            parts, sparts = [], []
            if he is not None:
                parts.append(_segsum(he, edge_graph, G)); sparts.append(_segsum(se, edge_graph, G))
            if hn is not None:
                parts.append(_segsum(hn, node_graph, G)); sparts.append(_segsum(sn, node_graph, G))
            if gf is not None:
                parts.append(gfr); sparts.append(np.abs(gfr))
            hg, sg = run(p["graph"], np.concatenate(parts, axis=1), np.concatenate(sparts, axis=1))
        for lst, v in zip(outs, (he, hn, hg)):
            lst.append(v)
        for lst, v in zip(scales, (se, sn, sg)):
            lst.append(v)
    pack = lambda ls: tuple(None if (lst[0] is None or lst[0].shape[1] == 0) else np.stack(lst) for lst in ls)
    return (pack(outs), pack(scales)) if return_scale else pack(outs)


# ----------------------------------------------------------------------------------------------------------
# Layout bridges between the two forms
# ----------------------------------------------------------------------------------------------------------
def packed_from_julia_shared(a):
    """Julia (D, T, B) → C-ABI (B, T, D).  Same bytes when `a` is column-major."""
    return None if a is None else np.ascontiguousarray(np.transpose(np.asarray(a), (2, 1, 0)))


def packed_from_julia_vector(items):
    """Vector of (D, T_g) → (1, ΣT, D); vector of (D,) → (1, G, D)."""
    if items is None:
        return None
    items = [np.asarray(x) for x in items]
    if items[0].ndim == 1:
        return np.stack(items)[None]
    return np.concatenate([x.T for x in items], axis=0)[None]


def flat_from_dense(x, key):
    """flatunpaddednf/ef (views.jl:80-98): (D, ΣT) in graph-major order from the padded arrays."""
    g = x["graphs"]
    a = x[key]
    flat = a.reshape(a.shape[0], -1, order="F")
    mask = g.flat_node_unpadder if key == "nf" else g.flat_edge_unpadder
    return flat[:, mask]

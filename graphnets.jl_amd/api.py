"""Host-side mirror of GraphNets.jl's public interface for the GNBlock / GNCore forward path.

Same names, argument meaning and error behaviour as the reference (`/root/reference/src/GraphNets.jl:12-50`), with
the compute done by libgnx.so (hand-written HIP, gfx950) through the C ABI in include/gnx.h.  Julia is not
available in this image, so this mirror is Python; the Julia `ccall` shim over the same ABI is julia/GraphNetsHIP.jl.

Conventions that make it read like the reference:
  * arrays keep Julia's shapes — ef (DE, E, B), nf (DN, N, B), gf (DG, B) — as torch tensors whose strides are
    column-major, i.e. the bytes are Julia's bytes (and the C ABI's packed [B][T][D] rows); indices are 0-based;
  * `nothing` is `None`; `@assert` failures are `AssertionError`;
  * the batched tuple is packed, not padded: `batch(x).ef` is (DE, ΣE, 1) for a vector of graphs instead of the
    reference's (DE, PN², B).  `unbatch`, `efview`/`nfview`/`gfview`, `flatunpaddednf/ef` return exactly what the
    reference returns; `padded(x)` materialises the reference's padded layout when it is wanted.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import GnxError, check

__all__ = ["GNGraphBatch", "NT", "batch", "unbatch", "efview", "nfview", "gfview", "flatunpaddednf", "flatunpaddedef",
           "Dense", "Chain", "LayerNorm", "GNBlock", "GNCore", "GNCoreList", "GNFeedForward", "GNGraphNorm", "zerodim2nothing",
           "padded", "unpadded", "GnxError", "getedgefninput", "getnodefninput", "getgraphfninput",
           "collapsef", "unpaddedcollapsedef", "flatunpaddedcollapsedef", "BlockPlan", "Graphed", "logitcrossentropy"]

_KEYS = ("graphs", "ef", "nf", "gf")


class NT:
    """The reference's NamedTuple `(graphs, ef, nf, gf)` (batch.jl:58-63)."""
    __slots__ = _KEYS

    def __init__(self, graphs=None, ef=None, nf=None, gf=None):
        self.graphs, self.ef, self.nf, self.gf = graphs, ef, nf, gf

    def keys(self):
        return _KEYS

    def __getitem__(self, k):
        return getattr(self, k)

    def __iter__(self):
        return iter((self.graphs, self.ef, self.nf, self.gf))

    def _replace(self, **kw):
        d = {k: getattr(self, k) for k in _KEYS}
        d.update(kw)
        return NT(**d)

    def __repr__(self):
        sh = lambda a: None if a is None else (tuple(a.shape) if hasattr(a, "shape") else f"list[{len(a)}]")
        return f"NT(graphs={type(self.graphs).__name__}, ef={sh(self.ef)}, nf={sh(self.nf)}, gf={sh(self.gf)})"


def _as_nt(t):
    if isinstance(t, NT):
        return t
    if isinstance(t, dict):
        assert set(t.keys()) == set(_KEYS), "keys must be (graphs, ef, nf, gf)"  # batch.jl:54
        return NT(**t)
    if hasattr(t, "_asdict"):
        d = t._asdict()
        assert set(d.keys()) == set(_KEYS)
        return NT(**d)
    return NT(*(getattr(t, k) for k in _KEYS))


def _device(device=None):
    if device is None:
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cuda", 0)
    return torch.device(device)


_NATIVE_ELEM = {"f": _lib.ELEM_F32, "d": _lib.ELEM_F64, "?": _lib.ELEM_U8, "B": _lib.ELEM_U8, "i": _lib.ELEM_I32, "l": _lib.ELEM_I64, "q": _lib.ELEM_I64}


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


# ------------------------------------------------------------------------------------------------------------
# GNGraphBatch — replaces the struct of dense broadcasters (gngraphbatch.jl:1-54) by a libgnx handle
# ------------------------------------------------------------------------------------------------------------
class GNGraphBatch:
    """`GNGraphBatch(adj_mats)`; `GNGraphBatch.from_csc(colptrs, rowvals, n_nodes)` is the sparse constructor
    (API extension: dense N×N input cannot hold BASELINE configs 2-5)."""

    def __init__(self, adj_mats=None, *, device=None, _csc=None, _dense_packed=None):
        lib = _lib.load()
        self.device = _device(device)
        self._h = C.c_void_p(None)
        self._ws = {}
        self._masks = None
        keep = []
        if _dense_packed is not None:
            buf, nn, kind, on_device, nbytes, ptr = _dense_packed
            with torch.cuda.device(self.device):
                if on_device:  # the library scans a device buffer on the NULL stream: whatever produced it on torch's current (possibly non-blocking) stream is complete first
                    torch.cuda.current_stream(self.device).synchronize()
                check(lib.gnx_graphs_create_dense_packed(ptr, nbytes, nn.ctypes.data_as(C.POINTER(C.c_int64)), int(nn.size), kind, 1, 1 if on_device else 0,
                                                         C.byref(self._h)))
            self._adj_mats = None
            self._packed_adj = (buf, nn)  # adj_mats: views of the packed buffer, made when somebody asks (unbatch on a matrix-form batch)
            self._finish_init(lib)
            return
        if _csc is not None:  # (array preparation and the length check need no device)
            colptrs, rowvals, n_nodes = _csc
            nn = np.ascontiguousarray(n_nodes, dtype=np.int64)
            G = int(nn.size)
            # ONE array per kind and one call (gnx_graphs_create_csc_cat): building 2 G ctypes pointers costs ~1 us each — 8 of the
            # 20 ms of a 4096-graph batch.  A caller that already holds the concatenated arrays (from_csc_packed) passes them as they
            # are; int32 indices stay int32 (half the bytes to validate and upload).
            def cat(parts):
                if isinstance(parts, np.ndarray) and parts.ndim == 1 and parts.dtype.kind in "iu":
                    a_ = parts
                else:
                    parts = list(parts)
                    a_ = np.concatenate(parts) if len(parts) > 1 else (np.asarray(parts[0]) if parts else np.zeros(0, np.int64))
                if a_.dtype != np.int32 and a_.dtype != np.int64:
                    a_ = a_.astype(np.int64)
                return np.ascontiguousarray(a_)
            cpc, rvc = cat(colptrs), cat(rowvals)
            if rvc.dtype != cpc.dtype:
                rvc = rvc.astype(cpc.dtype)
            if cpc.size != int(nn.sum()) + G:
                raise ValueError("every colptr must have n_nodes + 1 entries (colptr_cat: sum(n_nodes) + n_graphs)")
            keep += [cpc, rvc, nn]
        with torch.cuda.device(self.device):
            if _csc is not None:
                # (the C entry point checks both lengths against what the colptr arrays announce and never reads past them)
                check(lib.gnx_graphs_create_csc_cat(cpc.ctypes.data, cpc.size, rvc.ctypes.data if rvc.size else None, rvc.size,
                                                    nn.ctypes.data_as(C.POINTER(C.c_int64)), G, 0, cpc.dtype.itemsize * 8, C.byref(self._h)))
                self._adj_mats = None
            else:
                mats = [_np(a) for a in adj_mats]
                for a in mats:
                    assert a.ndim == 2, "adjacency matrix must be 2-D (checks.jl:11)"
                    assert a.shape[0] == a.shape[1], "adjacency matrix must be square"
                self._adj_mats = mats
                conv, kind = [], _lib.ELEM_I64
                native = _NATIVE_ELEM  # dtype.char -> element kind the ABI reads as it is (a bool / uint8 matrix is 8x fewer bytes than int64)
                for a in mats:  # (a dict lookup per matrix: the dtype comparisons of the first version cost 10 us each — 41 ms for 4096 graphs)
                    kind_a = native.get(a.dtype.char)
                    if kind_a is None:
                        kind_a, a = (_lib.ELEM_F64, a.astype(np.float64)) if a.dtype.kind == "f" else (_lib.ELEM_I64, a.astype(np.int64))
                    elif kind_a == _lib.ELEM_U8 and a.dtype.char == "?":
                        a = a.view(np.uint8)
                    conv.append((kind_a, a if a.flags.c_contiguous else np.ascontiguousarray(a)))
                kinds = {k for k, _ in conv}
                if len(kinds) > 1:  # mixed element types: promote to float64
                    conv = [(_lib.ELEM_F64, np.ascontiguousarray(b.astype(np.float64))) for _, b in conv]
                kind = conv[0][0] if conv else _lib.ELEM_I64
                keep += [b for _, b in conv]
                G = len(mats)
                ptrs = (C.c_void_p * max(G, 1))(*[b.ctypes.data for _, b in conv])
                nn = np.asarray([a.shape[0] for a in mats], dtype=np.int64)
                check(lib.gnx_graphs_create_dense(ptrs, nn.ctypes.data_as(C.POINTER(C.c_int64)), G, kind, 1, C.byref(self._h)))
        self._finish_init(lib)

    def _finish_init(self, lib):
        info = _lib.GraphsInfo()
        check(lib.gnx_graphs_get_info(self._h, C.byref(info)))
        self.n_graphs, self.n_nodes, self.n_edges = info.n_graphs, info.n_nodes, info.n_edges
        self.node_block_size, self.edge_block_size = info.node_block_size, info.edge_block_size  # gngraphbatch.jl:35-36
        self.n_tiles, self.max_in_degree = info.n_tiles, info.max_in_degree
        self.node_off = np.zeros(self.n_graphs + 1, dtype=np.int64)
        self.edge_off = np.zeros(self.n_graphs + 1, dtype=np.int64)
        check(lib.gnx_graphs_get_offsets(self._h, self.node_off.ctypes.data_as(C.POINTER(C.c_int64)),
                                         self.edge_off.ctypes.data_as(C.POINTER(C.c_int64))))

    @property
    def adj_mats(self):
        """the adjacency matrices the batch was made from (None for CSC input); for a packed dense batch: views of the packed buffer"""
        if self._adj_mats is None and getattr(self, "_packed_adj", None) is not None:
            buf, nn = self._packed_adj
            flat = buf.detach().cpu().numpy() if isinstance(buf, torch.Tensor) else buf
            offs = np.concatenate([[0], np.cumsum(nn * nn)])
            self._adj_mats = [flat[offs[i]:offs[i + 1]].reshape(int(n), int(n)) for i, n in enumerate(nn)]
        return self._adj_mats

    @adj_mats.setter
    def adj_mats(self, v):
        self._adj_mats = v

    @classmethod
    def from_dense_packed(cls, adj_cat, n_nodes, device=None):
        """`batch`'s input form (dense 0/1 matrices, src/batch.jl:53-64) as ONE buffer: adj_cat = the graphs' matrices one after the other, each
        row-major (numpy's order; A[i, j] = 1 <=> edge i -> j), as a 1-D numpy array or torch tensor of uint8 / bool / int32 / int64 / float32 /
        float64 with sum(n_g^2) elements.  No per-graph Python work (a list of 4096 matrices costs ~6 ms of attribute access and pointer
        extraction before the library is even called).  A CUDA tensor is read where it is (no copy); a pinned CPU tensor travels as one DMA;
        pageable memory goes through the library's pinned staging buffers."""
        nn = np.ascontiguousarray(n_nodes, dtype=np.int64)
        if isinstance(adj_cat, torch.Tensor):
            t = adj_cat.reshape(-1)
            if t.dtype == torch.bool:
                t = t.view(torch.uint8)
            kind = {torch.uint8: _lib.ELEM_U8, torch.int32: _lib.ELEM_I32, torch.int64: _lib.ELEM_I64, torch.float32: _lib.ELEM_F32, torch.float64: _lib.ELEM_F64}.get(t.dtype)
            if kind is None:
                raise ValueError(f"adjacency element type {t.dtype} is not supported")
            t = t.contiguous()
            packed = (t, nn, kind, t.is_cuda, t.numel() * t.element_size(), t.data_ptr())
            device = t.device if t.is_cuda and device is None else device
        else:
            a = np.asarray(adj_cat).reshape(-1)
            kind = _NATIVE_ELEM.get(a.dtype.char)
            if kind is None:
                kind, a = (_lib.ELEM_F64, a.astype(np.float64)) if a.dtype.kind == "f" else (_lib.ELEM_I64, a.astype(np.int64))
            elif a.dtype.char == "?":
                a = a.view(np.uint8)
            a = np.ascontiguousarray(a)
            packed = (a, nn, kind, False, a.nbytes, a.ctypes.data)
        numel = packed[0].numel() if isinstance(packed[0], torch.Tensor) else packed[0].size
        if nn.size == 0 or (nn <= 0).any() or numel != int((nn * nn).sum()):
            raise ValueError("adj_cat must hold sum(n_nodes^2) elements of n_nodes >= 1")
        return cls(device=device, _dense_packed=packed)

    @classmethod
    def from_csc(cls, colptrs, rowvals, n_nodes, device=None):
        """Per-graph 0-based CSC: colptrs[g] (N_g+1), rowvals[g] = local source index of every edge (sorted inside
        a column).  CSC nz order is the reference edge order (pad.jl:30)."""
        return cls(device=device, _csc=(colptrs, rowvals, n_nodes))

    @classmethod
    def from_csc_packed(cls, colptr_cat, rowval_cat, n_nodes, device=None):
        """The same batch from the CONCATENATED arrays (int32 or int64 numpy): colptr_cat = the graphs' 0-based colptr arrays one
        after the other (N_g + 1 entries each), rowval_cat = their rowval arrays.  No per-graph Python work."""
        return cls(device=device, _csc=(np.asarray(colptr_cat), np.asarray(rowval_cat), n_nodes))

    def __len__(self):
        return self.n_graphs

    def csc(self):
        """Global 0-based (colptr[N+1], rowval[E]) host copies."""
        colptr = np.zeros(self.n_nodes + 1, dtype=np.int64)
        rowval = np.zeros(max(self.n_edges, 1), dtype=np.int64)
        check(_lib.load().gnx_graphs_get_csc(self._h, colptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                             rowval.ctypes.data_as(C.POINTER(C.c_int64))))
        return colptr, rowval[: self.n_edges]

    def _unpadders(self):
        """flat_node_unpadder / flat_edge_unpadder Bool masks (gngraphbatch.jl:113-134), built on demand."""
        if self._masks is None:
            PN, B = self.node_block_size, self.n_graphs
            nm = np.zeros(B * PN, dtype=bool)
            em = np.zeros(B * PN * PN, dtype=bool)
            colptr, rowval = self.csc()
            dst = np.repeat(np.arange(self.n_nodes), np.diff(colptr))
            for g in range(B):
                n0, n1, e0, e1 = self.node_off[g], self.node_off[g + 1], self.edge_off[g], self.edge_off[g + 1]
                nm[g * PN: g * PN + (n1 - n0)] = True
                em[g * PN * PN + (rowval[e0:e1] - n0) + PN * (dst[e0:e1] - n0)] = True
            self._masks = (nm, em)
        return self._masks

    @property
    def flat_node_unpadder(self):
        return self._unpadders()[0]

    @property
    def flat_edge_unpadder(self):
        return self._unpadders()[1]

    def workspace(self, nbytes, layout):
        """Device scratch for one forward call on this handle, reused across calls with the same workspace LAYOUT (layer kind,
        widths, replicas) ON THE SAME STREAM (calls on one stream are ordered).  A buffer is never replaced, shrunk or handed
        to a call with another layout: a hipGraph that captured a call keeps replaying into the buffer it captured (the handle
        holds it for its whole life), and two streams never share scratch."""
        nbytes = max(int(nbytes), 256)
        key = (torch.cuda.current_stream(self.device).cuda_stream, layout, nbytes)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def __del__(self):
        try:
            if self._h:
                _lib.load().gnx_graphs_destroy(self._h)
                self._h = C.c_void_p(None)
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------
# layout helpers: Julia-shaped column-major views over packed [R][T][D] storage
# ------------------------------------------------------------------------------------------------------------
def _jl(c):
    """packed [R][T][D] contiguous tensor → Julia-shaped (D, T, R) view of the same bytes."""
    return None if c is None else c.permute(2, 1, 0)


def _packed(j):
    """Julia-shaped (D, T, R) tensor → contiguous [R][T][D] (no copy when it already is one of our views)."""
    return None if j is None else j.permute(2, 1, 0).contiguous()


def _dev_f32(a, device):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(_np(a), dtype=np.float32))
    return t.to(device=device, dtype=torch.float32)


def _ndim(a):
    return a.dim() if isinstance(a, torch.Tensor) else np.ndim(a)


def _size(a, d):
    return a.shape[d]


# ------------------------------------------------------------------------------------------------------------
# checks (src/checks.jl) — same assertions, AssertionError like Julia's @assert
# ------------------------------------------------------------------------------------------------------------
def _checks_shared(adj, ef, nf, gf):
    if ef is not None:
        assert _ndim(ef) == 3, "ndims(ef) != 3"  # checks.jl:63-84 (C, T, B)
    if nf is not None:
        assert _ndim(nf) == 3, "ndims(nf) != 3"
    if gf is not None:
        assert _ndim(gf) == 2, f"{_ndim(gf)} != 2"  # (C, B)
    bs = [_size(ef, 2)] if ef is not None else []
    bs += [_size(nf, 2)] if nf is not None else []
    bs += [_size(gf, 1)] if gf is not None else []
    assert len(set(bs)) <= 1, "batch sizes differ (checks.jl:169-183)"
    a = _np(adj)
    if ef is not None:
        ne = int((a == 1).sum())
        assert _size(ef, 1) == ne, f"{_size(ef, 1)} != num_edges"  # checks.jl:44
    if nf is not None:
        assert _size(nf, 1) == a.shape[0], f"{_size(nf, 1)} != {a.shape[0]}"  # checks.jl:45


def _checks_vector(adjs, ef, nf, gf):
    assert len(adjs) > 0  # checks.jl:8
    for name, v in (("ef", ef), ("nf", nf), ("gf", gf)):
        if v is not None:
            assert len(v) == len(adjs), f"length(adj_mats) != length({name}) (checks.jl:136-163)"
    for a in adjs:
        assert _ndim(a) == 2  # checks.jl:11
    for g, a in enumerate(adjs):
        a = _np(a)
        if ef is not None:
            assert _ndim(ef[g]) == 2
            assert _size(ef[g], 1) == int((a == 1).sum()), f"{_size(ef[g], 1)} != num_edges"
        if nf is not None:
            assert _ndim(nf[g]) == 2, f"{_ndim(nf[g])} != 2"
            assert _size(nf[g], 1) == a.shape[0], f"{_size(nf[g], 1)} != {a.shape[0]}"
        if gf is not None:
            assert _ndim(gf[g]) == 1


# ------------------------------------------------------------------------------------------------------------
# batch / unbatch / views (src/batch.jl, src/unbatch.jl, src/unpad.jl, src/views.jl)
# ------------------------------------------------------------------------------------------------------------
def _is_matrix(g):
    return isinstance(g, (np.ndarray, torch.Tensor)) and g.ndim == 2


def batch(t, device=None):
    """`batch(t::NamedTuple)` (batch.jl:53-64).  `graphs` is one adjacency matrix (shared by the whole data batch),
    a vector of adjacency matrices, or an existing GNGraphBatch (e.g. from_csc)."""
    if isinstance(t, dict):
        assert set(t.keys()) == set(_KEYS), "keys must be (graphs, ef, nf, gf)"
    t = _as_nt(t)
    graphs, ef, nf, gf = t
    assert not (ef is None and nf is None and gf is None), "ef, nf and gf are all nothing (batch.jl:56)"
    dev = _device(device)
    if isinstance(graphs, GNGraphBatch):
        shared = graphs.n_graphs == 1 and not isinstance(next(x for x in (ef, nf, gf) if x is not None), (list, tuple))
        g = graphs
        dev = g.device
    elif _is_matrix(graphs):
        shared = True
        _checks_shared(graphs, ef, nf, gf)
        g = GNGraphBatch([graphs], device=dev)
    else:
        shared = False
        graphs = list(graphs)
        _checks_vector(graphs, ef, nf, gf)
        g = GNGraphBatch(graphs, device=dev)
    if shared:
        bef = None if ef is None else _jl(_packed(_dev_f32(ef, dev)))
        bnf = None if nf is None else _jl(_packed(_dev_f32(nf, dev)))
        bgf = None if gf is None else _jl(_dev_f32(gf, dev).t().contiguous()[:, None, :])  # (DG,B) → [B][1][DG]
        for name, a, T in (("ef", bef, g.n_edges), ("nf", bnf, g.n_nodes)):
            if a is not None:
                assert a.shape[1] == T, f"size({name}, 2) = {a.shape[1]} != {T} (checks.jl:41-46)"
    else:
        cat = lambda items: torch.cat([_dev_f32(x, dev).t() for x in items], dim=0).contiguous()[None]  # [1][ΣT][D]
        bef = None if ef is None else _jl(cat(ef))
        bnf = None if nf is None else _jl(cat(nf))
        bgf = None if gf is None else _jl(torch.stack([_dev_f32(x, dev) for x in gf]).contiguous()[None])  # [1][G][DG]
        for name, a, T in (("ef", bef, g.n_edges), ("nf", bnf, g.n_nodes), ("gf", bgf, g.n_graphs)):
            if a is not None:
                assert a.shape[1] == T, f"{name}: {a.shape[1]} columns != {T} (checks.jl:41-46)"
    return NT(g, bef, bnf, bgf)


def _shared_like(g):
    return g.n_graphs == 1  # unbatch.jl:15-17: a 1-graph batch unbatches through the shared-adjacency branch


def unbatch(t):
    """`unbatch` (unbatch.jl:6-39): the inverse of `batch`; results alias the batched arrays (views, like
    unpad.jl's @view)."""
    t = _as_nt(t)
    g, ef, nf, gf = t
    assert not (ef is None and nf is None and gf is None)
    if _shared_like(g):
        graphs = g.adj_mats[0] if g.adj_mats is not None else g
        return NT(graphs, ef, nf, None if gf is None else gf[:, 0, :])  # gf → (DG, B) (unpad.jl:19-21)
    eo, no = g.edge_off, g.node_off
    efs = None if ef is None else [ef[:, eo[i]:eo[i + 1], 0] for i in range(g.n_graphs)]
    nfs = None if nf is None else [nf[:, no[i]:no[i + 1], 0] for i in range(g.n_graphs)]
    gfs = None if gf is None else [gf[:, i, 0] for i in range(g.n_graphs)]
    return NT(g.adj_mats if g.adj_mats is not None else g, efs, nfs, gfs)


def efview(t, d1, d2, d3):
    """`efview(t, d1, d2, d3)` (views.jl:6-31): d2 indexes the real edges of graph/batch element d3 in edge order."""
    t = _as_nt(t)
    if t.ef is None:
        return None
    if _shared_like(t.graphs):
        return t.ef[d1, d2, d3]
    eo = t.graphs.edge_off
    return t.ef[:, eo[d3]:eo[d3 + 1], 0][d1, d2]


def nfview(t, d1, d2, d3):
    """`nfview(t, d1, d2, d3)` (views.jl:38-61)."""
    t = _as_nt(t)
    if t.nf is None:
        return None
    if _shared_like(t.graphs):
        return t.nf[d1, d2, d3]
    no = t.graphs.node_off
    return t.nf[:, no[d3]:no[d3 + 1], 0][d1, d2]


def gfview(t, d1, d2):
    """`gfview(t, d1, d2)` (views.jl:68-78)."""
    t = _as_nt(t)
    if t.gf is None:
        return None
    return t.gf[d1, 0, d2] if _shared_like(t.graphs) else t.gf[d1, d2, 0]


def _flat(a):
    D, T, R = a.shape
    return a.permute(2, 1, 0).reshape(R * T, D).t()  # (D, T*R), column t + T*r — Julia's reshape(a, D, T*B)


def flatunpaddednf(t):
    """`flatunpaddednf` (views.jl:80-88): (DN, ΣN) over all real nodes, graph-major.  The packed layout already is it."""
    return _flat(_as_nt(t).nf)


def flatunpaddedef(t):
    """`flatunpaddedef` (views.jl:90-98)."""
    return _flat(_as_nt(t).ef)


def zerodim2nothing(t):
    """gnblock.jl:71-78."""
    t = _as_nt(t)
    z = lambda a: None if (a is None or a.shape[0] == 0) else a
    return NT(t.graphs, z(t.ef), z(t.nf), z(t.gf))


def padded(t):
    """The reference's padded batched arrays — ef (DE, PN², B), nf (DN, PN, B), gf (DG, 1, B) — with zeros in the
    pads (the reference leaves act(bias) junk there; SURVEY §8b 'raw padded arrays')."""
    t = _as_nt(t)
    g = t.graphs
    lib = _lib.load()
    stream = torch.cuda.current_stream(g.device).cuda_stream
    out = {}
    for key, kind, PT in (("ef", 0, g.edge_block_size), ("nf", 1, g.node_block_size)):
        a = getattr(t, key)
        if a is None:
            out[key] = None
            continue
        c = _packed(a)
        R, T, D = c.shape
        B = R if g.n_graphs == 1 else g.n_graphs
        p = torch.empty((B, PT, D), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(lib.gnx_pad_features(g._h, kind, c.data_ptr(), D, R, p.data_ptr(), stream))
        out[key] = _jl(p)
    gf = t.gf
    if gf is not None and not _shared_like(g):
        gf = gf.permute(0, 2, 1)  # (DG, G, 1) → (DG, 1, G)
    return NT(g, out["ef"], out["nf"], gf)


def unpadded(graphs, ef=None, nf=None, gf=None):
    """The inverse bridge of `padded()` (unpad.jl:1-17: `unpadef`, `unpadnf`, `unpadgf`): the reference's padded batched arrays —
    ef (DE, PN², B), nf (DN, PN, B), gf (DG, 1, B) — over a GNGraphBatch become the batch form this package computes on (what `batch`
    returns).  Whatever the pads hold is dropped."""
    g = graphs
    assert isinstance(g, GNGraphBatch), "graphs must be a GNGraphBatch (batch(...).graphs)"
    lib = _lib.load()
    stream = torch.cuda.current_stream(g.device).cuda_stream
    out = {}
    for key, a, kind, PT, T in (("ef", ef, 0, g.edge_block_size, g.n_edges), ("nf", nf, 1, g.node_block_size, g.n_nodes)):
        if a is None:
            out[key] = None
            continue
        p = _packed(_dev_f32(a, g.device))  # [B][PT][D]
        B, pt, D = p.shape
        assert pt == PT, f"{key}: {pt} padded slots != {PT} (pad.jl:12-64)"
        R = B if g.n_graphs == 1 else 1
        assert B == (R if g.n_graphs == 1 else g.n_graphs), f"{key}: {B} slices != {g.n_graphs} graphs"
        c = torch.empty((R, T, D), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(lib.gnx_unpad_features(g._h, kind, p.data_ptr(), D, R, c.data_ptr(), stream))
        out[key] = _jl(c)
    if gf is not None:
        gf = _dev_f32(gf, g.device)
        assert gf.dim() == 3 and gf.shape[1] == 1, "gf: the padded form is (DG, 1, B) (pad.jl:63-64)"
        if not _shared_like(g):
            gf = gf.permute(0, 2, 1)  # (DG, 1, G) → (DG, G, 1)
    return NT(g, out["ef"], out["nf"], gf)


def _collapse(t):
    t = _as_nt(t)
    g, ef = t.graphs, t.ef
    assert ef is not None, "collapsing needs edge features"
    lib = _lib.load()
    off = np.zeros(g.n_graphs + 1, dtype=np.int64)
    with torch.cuda.device(g.device):
        check(lib.gnx_collapse_offsets(g._h, off.ctypes.data_as(C.POINTER(C.c_int64))))
        c = _packed(ef)
        R, _, D = c.shape
        out = torch.empty((R, int(off[-1]), D), dtype=torch.float32, device=g.device)
        check(lib.gnx_collapse_edges(g._h, c.data_ptr(), D, R, out.data_ptr(), torch.cuda.current_stream(g.device).cuda_stream))
    return _jl(out), off


def collapsef(t):
    """`collapsef` (gngraphbatch.jl:83-85): `batched_mul(ef, edge_collapser) / 2` — the padded array form (DE, PN(PN+1)/2, B): for every
    coordinate (i, j), i >= j, of the padded grid in column-major order the symmetric average of slots i->j and j->i (the diagonal
    keeps its slot).  Non-edge slots count as 0 (the reference reads whatever its padded array holds there)."""
    t = _as_nt(t)
    g, ef = t.graphs, t.ef
    assert ef is not None, "collapsing needs edge features"
    lib = _lib.load()
    with torch.cuda.device(g.device):
        c = _packed(ef)
        R, _, D = c.shape
        B = R if _shared_like(g) else g.n_graphs
        PN = g.node_block_size
        out = torch.empty((B, PN * (PN + 1) // 2, D), dtype=torch.float32, device=g.device)
        check(lib.gnx_collapse_padded(g._h, c.data_ptr(), D, R, out.data_ptr(), torch.cuda.current_stream(g.device).cuda_stream))
    return _jl(out)


def unpaddedcollapsedef(t):
    """`unpaddedcollapsedef` (gngraphbatch.jl:87-107): per graph / batch element, the symmetric averages
    (ef[i->j] + ef[j->i]) / 2 over the real edges of the lower triangle (i >= j), in edge order.  A missing reverse edge
    contributes 0 (the reference reads the padded slot there)."""
    out, off = _collapse(t)
    g = _as_nt(t).graphs
    if _shared_like(g):
        return [out[:, :, b] for b in range(out.shape[2])]
    return [out[:, off[i]:off[i + 1], 0] for i in range(g.n_graphs)]


def flatunpaddedcollapsedef(t):
    """`flatunpaddedcollapsedef` (gngraphbatch.jl:109-111): hcat of `unpaddedcollapsedef`."""
    out, _ = _collapse(t)
    return _flat(out)


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):  # a, b: [cols][d] contiguous
        lib = _lib.load()
        cols, d = a.shape
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = torch.empty(max(int(lib.gnx_xent_workspace_bytes(cols)), 16), dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            check(lib.gnx_logit_cross_entropy(a.data_ptr(), b.data_ptr(), d, cols, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                              torch.cuda.current_stream(a.device).cuda_stream))
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        cols, d = a.shape
        dl = torch.empty_like(a)
        up = g.reshape(1).float().contiguous()
        with torch.cuda.device(a.device):
            check(_lib.load().gnx_logit_cross_entropy_backward(a.data_ptr(), b.data_ptr(), d, cols, up.data_ptr(), dl.data_ptr(),
                                                               torch.cuda.current_stream(a.device).cuda_stream))
        return dl, None


def logitcrossentropy(yhat, y):
    """`Flux.logitcrossentropy(ŷ, y)` on (d, cols) arrays such as `flatunpaddednf(ŷ)` (examples/sort/sort.jl:69-81):
    mean over columns of -sum(y .* logsoftmax(ŷ; dims=1); dims=1).  Returns a 0-d device tensor; differentiable w.r.t. ŷ."""
    assert yhat.dim() == 2 and tuple(yhat.shape) == tuple(y.shape), "ŷ and y must be (d, cols) arrays of the same size"
    a = yhat.t().contiguous().float()   # [cols][d] rows = the bytes of a column-major (d, cols) array
    b = y.to(yhat.device).t().contiguous().float()
    return _XentFn.apply(a, b)


def _fn_input(kind, graphs, ef, nf, gf):
    g = graphs
    assert isinstance(g, GNGraphBatch), "graphs must be the GNGraphBatch of a batched tuple"
    present = [a for a in (ef, nf, gf) if a is not None]
    assert present, "ef, nf and gf are all nothing"
    R = present[0].shape[2]
    c = [_packed(a) for a in (ef, nf, gf)]
    d = [0 if a is None else a.shape[2] for a in c]
    T = (g.n_edges, g.n_nodes, g.n_graphs)[kind]
    K = d[0] + (2 if kind == 0 else 1) * d[1] + d[2]
    out = torch.empty((R, T, K), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        check(_lib.load().gnx_fn_input(g._h, kind, _ptr(c[0]), d[0], _ptr(c[1]), d[1], _ptr(c[2]), d[2], R, out.data_ptr(),
                                       torch.cuda.current_stream(g.device).cuda_stream))
    return _jl(out)


def getedgefninput(graphs, edge_features, node_features, graph_features):
    """`getedgefninput` (edgefninput.jl:1-47): vcat(ef, nf⊗src, nf⊗dst, gf⊗g2e) per real edge — (K_e, ΣE, R), packed."""
    return _fn_input(0, graphs, edge_features, node_features, graph_features)


def getnodefninput(graphs, edge_features, node_features, graph_features):
    """`getnodefninput` (nodefninput.jl:1-24): vcat(Σ_{e→n} ef, nf, gf⊗g2n) per real node."""
    return _fn_input(1, graphs, edge_features, node_features, graph_features)


def getgraphfninput(graphs, edge_features, node_features, graph_features):
    """`getgraphfninput` (graphfninput.jl:1-13): vcat(Σ_e ef, Σ_n nf, gf) per graph."""
    return _fn_input(2, graphs, edge_features, node_features, graph_features)


# ------------------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------------------
def _colmajor(w):
    """(out, in) tensor whose storage is column-major — the bytes of Flux's Dense.weight."""
    return w.t().contiguous().t()


class Dense:
    """Flux `Dense(in => out, σ)`: weight (out, in) glorot-uniform, bias zeros."""

    def __init__(self, in_dim, out_dim, act="identity", device=None, generator=None):
        dev = _device(device)
        s = math.sqrt(6.0 / (in_dim + out_dim)) if in_dim + out_dim > 0 else 0.0
        w = (torch.rand((in_dim, out_dim), generator=generator) * 2 - 1) * s  # stored transposed → column-major
        self.weight = w.to(dev).t()
        self.bias = torch.zeros(out_dim, device=dev)
        self.act = act

    @classmethod
    def from_numpy(cls, W, b, act="identity", device=None):
        self = cls.__new__(cls)
        dev = _device(device)
        self.weight = _colmajor(torch.from_numpy(np.asarray(W, dtype=np.float32)).to(dev))
        self.bias = None if b is None else torch.from_numpy(np.asarray(b, dtype=np.float32)).to(dev)
        self.act = act if isinstance(act, str) else {v: k for k, v in _lib.ACT.items()}[act]
        return self

    def _c(self, keep):
        w = self.weight
        if w.dim() != 2 or (w.numel() > 0 and (w.stride(0) != 1 or w.stride(1) != w.shape[0])) or w.dtype != torch.float32:
            w = _colmajor(w.float())
        keep.append(w)
        b = self.bias
        if b is not None:
            b = b.float().contiguous()
            keep.append(b)
        return _lib.Dense(w.data_ptr() if w.numel() else None, None if b is None or b.numel() == 0 else b.data_ptr(),
                          _lib.ACT[self.act], 0)

    _c_layer = _c  # as an entry of a Chain (gnx_chain.layers)


class Dropout:
    """Flux `Dropout(p)` as a layer value: what ends a FeedForward chain (gnfeedforward.jl:27-31) and what `GNBlock` keeps in its `dropout`
    field without applying it (gnblock.jl:59, :63-69).  Inside a `Chain` it is the identity in test mode."""

    def __init__(self, p=0.0):
        assert 0 <= p <= 1, "Dropout: p must lie in [0, 1]"
        self.p = float(p)


_ACT_CALLABLES = {torch.relu: "relu", torch.tanh: "tanh", torch.sigmoid: "sigmoid", torch.nn.functional.relu: "relu"}  # (gelu by name: NNlib's tanh form, not torch's default)


class Chain:
    """Flux `Chain(layers...)` — a GNBlock's update functions may be any chain (gnblock.jl:1-6): `blk.edgefn = Chain(Dense(20, 64, "relu"),
    Dense(64, 3))`.  (A one-layer chain is the constructor's default, gnblock.jl:55-60.)  The HIP path runs row-wise `Dense` layers
    (gnx_chain_block_forward); the other layer values a Flux user writes between them are folded on the host where that is exact:
      * an activation function as a layer (`"relu"`, `torch.tanh`, ... — Flux: `Chain(Dense(a => b), relu)`) becomes the activation of the `Dense`
        in front of it when that one has none — `relu.(W x .+ b)` either way;
      * `"identity"` / `None` is dropped;
      * `Dropout(p)` is the identity in test mode and dropped; a differentiable call of a block whose chains hold a Dropout with p > 0 is refused
        (the training-mode Dropout of this library is the FeedForward's: GNCore(dims; dropout)).
      * a `LayerNorm(d)` layer value normalises the rows of the layer in front of it (`Chain(Dense(a => d, relu), LayerNorm(d), Dense(d => b))`):
        a row-wise launch of its own, differentiable (gamma / beta gradients); as the EDGE function's first layer it runs behind an identity Dense
        that the library puts in front (the fused first launch then writes getedgefninput itself).
    Anything else (BatchNorm, SkipConnection, closures) raises NotImplementedError: wrap such layers outside the block."""

    def __init__(self, *layers):
        given = list(layers[0]) if len(layers) == 1 and isinstance(layers[0], (list, tuple)) else list(layers)
        self.layers, self.dropout_p = [], 0.0
        for l in given:
            if isinstance(l, (Dense, LayerNorm)):
                self.layers.append(l)
            elif l is None or l == "identity":
                continue
            elif isinstance(l, Dropout):
                self.dropout_p = max(self.dropout_p, l.p)
            elif (isinstance(l, str) and l in _lib.ACT) or (callable(l) and l in _ACT_CALLABLES):
                name = l if isinstance(l, str) else _ACT_CALLABLES[l]
                if not self.layers or not isinstance(self.layers[-1], Dense) or self.layers[-1].act != "identity":
                    raise NotImplementedError(f"Chain: the activation layer {name!r} does not follow a Dense without activation — only that form folds into "
                                              "the row-wise Dense launches (gnx_chain_block_forward)")
                d = Dense.__new__(Dense)  # the same weight / bias tensors, with the activation
                d.weight, d.bias, d.act = self.layers[-1].weight, self.layers[-1].bias, name
                self.layers[-1] = d
            else:  # gnblock.jl:1-6 admits any Flux chain; the HIP path has row-wise Dense layers only
                raise NotImplementedError(f"Chain: layer of type {type(l).__name__} is not supported — the update functions of a GNBlock "
                                          "run as chains of Dense layers (gnx_chain_block_forward); wrap other layers outside the block")

    def __len__(self):
        return len(self.layers)

    @property
    def out_width(self):
        return int(self.layers[-1].weight.shape[0]) if self.layers else 0


class LayerNorm:
    """Flux `LayerNorm(d)`: γ = ones, β = zeros, ε = 1e-5.  GNGraphNorm's layers (gngraphnorm.jl:9-17) — and a layer value of a `Chain`."""

    def __init__(self, d, device=None):
        dev = _device(device)
        self.gamma = torch.ones(d, device=dev)
        self.beta = torch.zeros(d, device=dev)

    @classmethod
    def from_numpy(cls, gamma, beta, device=None):
        self = cls.__new__(cls)
        dev = _device(device)
        self.gamma = torch.from_numpy(np.asarray(gamma, dtype=np.float32)).to(dev)
        self.beta = torch.from_numpy(np.asarray(beta, dtype=np.float32)).to(dev)
        return self

    # as a layer of a Chain the (gamma, beta) pair takes the places of a Dense's (weight, bias): parameter lists, widths, gradient slots
    weight = property(lambda self: self.gamma)
    bias = property(lambda self: self.beta)
    act = "identity"

    def _c_layer(self, keep):
        g, b = self.gamma.float().contiguous(), self.beta.float().contiguous()
        keep += [g, b]
        return _lib.Dense(g.data_ptr(), b.data_ptr(), _lib.ACT["identity"], _lib.LAYER_LAYERNORM)

    def _c(self, keep):
        g, b = self.gamma.float().contiguous(), self.beta.float().contiguous()
        keep += [g, b]
        return _lib.LayerNorm(g.data_ptr(), b.data_ptr())


class _Prepared:
    """A layer's prepared parameters (`gnx_block_prepare` / `gnx_core_prepare`, include/gnx.h): the weight blocks in the forms the matrix-core
    kernels stage, made once — `model |> device` happens once in the reference (examples/sort/sort.jl:29,89).  The mirror remembers the
    version counter of EVERY parameter tensor of the layer (whatever subset the C side bakes into planes today): a descriptor built after an
    in-place update (an optimiser step, a `copy_`, a partial `load_state_dict`) refreshes them first — also inside a gradient call, whose
    autograd.Function.forward runs with grad mode off, so the version list is the one thing correctness rests on; writes through `tensor.data`
    bypass the counter (torch's own caveat): call `prepare()` again after those
    first (`gnx_prepared_refresh`, stream-ordered with the forward that follows); tensors that were REPLACED make the planes unreachable (they
    are looked up by the weight pointers), i.e. the forward prepares per call again until `prepare()` is called anew."""

    def __init__(self, handle, device, tensors, keep):
        self.handle, self.device, self.keep = handle, device, keep
        self.tensors = [t for t in tensors if t is not None]
        self.versions = [t._version for t in self.tensors]

    def current(self):
        """the handle, refreshed if a source tensor was written in place since the planes were made"""
        if any(t._version != v for t, v in zip(self.tensors, self.versions)):
            with torch.cuda.device(self.device):
                check(_lib.load().gnx_prepared_refresh(self.handle, torch.cuda.current_stream(self.device).cuda_stream))
            self.versions = [t._version for t in self.tensors]
        return self.handle

    def nbytes(self):
        return int(_lib.load().gnx_prepared_bytes(self.handle))

    def __del__(self):
        try:
            if self.handle:
                _lib.load().gnx_prepared_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _pair(in_dims, out_dims):
    if out_dims is None:
        in_dims, out_dims = in_dims  # GNBlock((in, out)) ~ `in => out`
    return tuple(int(d) for d in in_dims), tuple(int(d) for d in out_dims)


def _forward_common(x, in_dims):
    x = _as_nt(x)
    g, ef, nf, gf = x
    assert isinstance(g, GNGraphBatch), "x must come from batch()"
    present = [a for a in (ef, nf, gf) if a is not None]
    assert present, "ef, nf and gf are all nothing"
    R = present[0].shape[2]
    for name, a, d, T in (("ef", ef, in_dims[0], g.n_edges), ("nf", nf, in_dims[1], g.n_nodes), ("gf", gf, in_dims[2], g.n_graphs)):
        if a is None:
            assert d == 0, f"{name} is nothing but the layer expects width {d}"
        else:
            assert a.shape[0] == d, f"DimensionMismatch: {name} has {a.shape[0]} rows, layer expects {d}"
            assert a.shape[1] == T and a.shape[2] == R, f"{name} has shape {tuple(a.shape)}, expected (*, {T}, {R})"
    return g, _packed(ef), _packed(nf), _packed(gf), R


def _ptr(t):
    return None if t is None else t.data_ptr()


class GNBlock:
    """`GNBlock(in => out; dropout=0)` (gnblock.jl:47-61).  `block(x)` = gnblock.jl:63-69 via gnx_block_forward."""

    def __init__(self, in_dims, out_dims=None, dropout=0, device=None, generator=None, act=("identity",) * 3):
        in_dims, out_dims = _pair(in_dims, out_dims)
        assert any(d > 0 for d in in_dims)  # gnblock.jl:48
        assert any(d > 0 for d in out_dims)  # gnblock.jl:49
        de, dn, dg = in_dims
        oe, on, og = out_dims
        self.in_dims, self.out_dims = in_dims, out_dims
        self.edgefn = Dense(de + 2 * dn + dg, oe, act[0], device, generator)  # gnblock.jl:52,56
        self.nodefn = Dense(dn + oe + dg, on, act[1], device, generator)      # gnblock.jl:53,57
        self.graphfn = Dense(on + oe + dg, og, act[2], device, generator)     # gnblock.jl:54,58
        self.dropout = dropout  # stored, never applied by the forward (gnblock.jl:63-69)
        self.flags = 0

    def _c(self, keep):
        p = _lib.BlockParams()
        p.de, p.dn, p.dg = self.in_dims
        p.oe, p.on, p.og = self.out_dims
        p.edgefn, p.nodefn, p.graphfn = self.edgefn._c(keep), self.nodefn._c(keep), self.graphfn._c(keep)
        q = getattr(self, "_prepared", None)
        if q is not None:  # (a training step's in-place updates move the version counters of the tracked tensors: q.current() refreshes the planes first)
            p.prepared = q.current()
        return p

    def prepare(self):
        """`gnx_block_prepare`: split / transpose / slot-permute this block's weight blocks for the matrix-core kernels ONCE (what `gpu(model)` is
        in the Julia shim).  Forwards outside autograd then launch no preparation kernel; results are bit-identical.  Returns self."""
        if isinstance(self.edgefn, Chain) or isinstance(self.nodefn, Chain) or isinstance(self.graphfn, Chain):
            return self  # (Chain blocks run through gnx_chain_block_forward: nothing to prepare)
        self._prepared = None
        keep = []
        p = self._c(keep)
        dev = self.edgefn.weight.device
        h = C.c_void_p()
        with torch.cuda.device(dev):
            check(_lib.load().gnx_block_prepare(C.byref(p), torch.cuda.current_stream(dev).cuda_stream, C.byref(h)))
        # every parameter the C side may bake into a plane is a source (the edge planes, the projections, the node planes of k_node_x6, the
        # folded biases): an in-place change of ANY of them refreshes the planes (nodefn.weight alone used to go unnoticed)
        self._prepared = _Prepared(h, dev, [t for l in (self.edgefn, self.nodefn, self.graphfn) for t in (l.weight, l.bias)], keep)
        return self

    def _trainable(self, tensors):
        ps = [self.edgefn.weight, self.edgefn.bias, self.nodefn.weight, self.nodefn.bias, self.graphfn.weight, self.graphfn.bias]
        return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in list(tensors) + ps)

    def _sync_dims(self):
        """The update functions are plain fields (gnblock.jl:1-6: a struct of three functions) that a caller may replace after construction: the
        output widths are read off the layers, and a function with outputs must take what the block feeds it (Flux's DimensionMismatch)."""
        de, dn, dg = self.in_dims
        oe, on, og = (int(f.weight.shape[0]) for f in (self.edgefn, self.nodefn, self.graphfn))
        for name, f, k in (("edgefn", self.edgefn, de + 2 * dn + dg), ("nodefn", self.nodefn, oe + dn + dg), ("graphfn", self.graphfn, oe + on + dg)):
            assert f.weight.shape[0] == 0 or int(f.weight.shape[1]) == k, f"{name}: Dense({int(f.weight.shape[1])} => {int(f.weight.shape[0])}) in a block that feeds it {k} rows"
        assert oe + on + og > 0  # gnblock.jl:49
        self.out_dims = (oe, on, og)

    def _as_chain(self, fn):
        return fn if isinstance(fn, Chain) else Chain(fn)

    def _chain_params(self, keep):
        """gnx_chain_block_params of this block's update functions (each a Chain of Dense layers) + the chains themselves."""
        chains = [self._as_chain(f) for f in (self.edgefn, self.nodefn, self.graphfn)]
        p = _lib.ChainBlockParams()
        p.de, p.dn, p.dg = self.in_dims
        outw = []
        for name, ch in zip(("edgefn", "nodefn", "graphfn"), chains):
            widths = [int(l.weight.shape[0]) for l in ch.layers]
            arr = (_lib.Dense * max(len(ch.layers), 1))(*[l._c_layer(keep) for l in ch.layers])
            wid = (C.c_int32 * max(len(ch.layers), 1))(*widths)
            keep += [arr, wid]
            c = getattr(p, name)
            c.layers, c.widths, c.n_layers = arr, wid, len(ch.layers)
            outw.append(widths[-1] if widths else 0)
        return p, chains, outw

    def _chain_forward(self, g, R, flags, ef, nf, gf):
        lib = _lib.load()
        keep = []
        p, chains, outw = self._chain_params(keep)
        dev = g.device
        mk = lambda T, d: torch.empty((R, T, d), dtype=torch.float32, device=dev) if d > 0 else None
        eo, no, go = mk(g.n_edges, outw[0]), mk(g.n_nodes, outw[1]), mk(g.n_graphs, outw[2])
        with torch.cuda.device(dev):
            nbytes = lib.gnx_chain_block_workspace_bytes(g._h, C.byref(p), R)
            if nbytes == 0:
                raise GnxError(_lib.ERR_DIMS, lib.gnx_last_error().decode("utf-8", "replace"))
            ws = g.workspace(nbytes, ("chain", self.in_dims, tuple(tuple(int(l.weight.shape[0]) for l in ch.layers) for ch in chains), R))
            check(lib.gnx_chain_block_forward(g._h, C.byref(p), _ptr(ef), _ptr(nf), _ptr(gf), R, _ptr(eo), _ptr(no), _ptr(go),
                                              ws.data_ptr(), ws.numel(), flags, torch.cuda.current_stream(dev).cuda_stream))
        return eo, no, go

    def _call_chains(self, x, flags):
        """Update functions that are multi-layer Chains: gnx_chain_block_forward; differentiable through gnx_chain_block_backward."""
        g, ef, nf, gf, R = _forward_common(x, self.in_dims)
        flags = self.flags if flags is None else flags
        chains = [self._as_chain(f) for f in (self.edgefn, self.nodefn, self.graphfn)]
        params = [t for ch in chains for l in ch.layers for t in (l.weight, l.bias)]
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in [ef, nf, gf] + params):
            if any(ch.dropout_p > 0 for ch in chains):
                raise NotImplementedError("GNBlock: a Dropout(p > 0) inside an update-function Chain is the identity in test mode only; a differentiable "
                                          "call would train another model — use GNCore(dims; dropout = p) for the FeedForward's Dropout")
            outs = iter(_ChainBlockFn.apply(self, g, R, flags, ef, nf, gf, *params))
            eo, no, go = (next(outs) if ch.out_width > 0 else None for ch in chains)
        else:
            eo, no, go = self._chain_forward(g, R, flags, ef, nf, gf)
        return NT(g, _jl(eo), _jl(no), _jl(go))

    def __call__(self, x, flags=None):
        if any(isinstance(f, Chain) and (len(f) != 1 or isinstance(f.layers[0], LayerNorm)) for f in (self.edgefn, self.nodefn, self.graphfn)):
            return self._call_chains(x, flags)
        if any(isinstance(f, Chain) and f.dropout_p > 0 for f in (self.edgefn, self.nodefn, self.graphfn)):
            return self._call_chains(x, flags)
        if any(isinstance(f, Chain) for f in (self.edgefn, self.nodefn, self.graphfn)):  # one-layer chains are plain Dense layers
            self.edgefn, self.nodefn, self.graphfn = (f.layers[0] if isinstance(f, Chain) else f for f in (self.edgefn, self.nodefn, self.graphfn))
        self._sync_dims()
        g, ef, nf, gf, R = _forward_common(x, self.in_dims)
        if self._trainable((ef, nf, gf)):  # differentiable call: gnx_block_backward is the pullback
            outs = iter(_BlockFn.apply(self, g, R, self.flags if flags is None else flags, ef, nf, gf, self.edgefn.weight, self.edgefn.bias,
                                       self.nodefn.weight, self.nodefn.bias, self.graphfn.weight, self.graphfn.bias))
            eo, no, go = (next(outs) if d > 0 else None for d in self.out_dims)
            return NT(g, _jl(eo), _jl(no), _jl(go))
        lib = _lib.load()
        keep = []
        p = self._c(keep)
        oe, on, og = self.out_dims
        dev = g.device
        mk = lambda T, d: torch.empty((R, T, d), dtype=torch.float32, device=dev) if d > 0 else None
        eo, no, go = mk(g.n_edges, oe), mk(g.n_nodes, on), mk(g.n_graphs, og)
        with torch.cuda.device(dev):
            ws = g.workspace(lib.gnx_block_workspace_bytes(g._h, C.byref(p), R), ("block", self.in_dims, self.out_dims, R))
            check(lib.gnx_block_forward(g._h, C.byref(p), _ptr(ef), _ptr(nf), _ptr(gf), R, _ptr(eo), _ptr(no), _ptr(go),
                                        ws.data_ptr(), ws.numel(), (self.flags if flags is None else flags),
                                        torch.cuda.current_stream(dev).cuda_stream))
        return NT(g, _jl(eo), _jl(no), _jl(go))  # zero-width outputs are None (gnblock.jl:71-78)


class _BlockFn(torch.autograd.Function):
    """torch autograd node of one GNBlock call: forward = gnx_block_forward, backward = gnx_block_backward (the analogue of
    the Zygote `rrule` the Julia shim would define).  Tensors are packed [R][T][D]; weights (out, in) column-major."""

    @staticmethod
    def forward(ctx, block, g, R, flags, ef, nf, gf, We, be, Wn, bn, Wg, bg):
        lib = _lib.load()
        keep = []
        p = block._c(keep)
        oe, on, og = block.out_dims
        dev = g.device
        mk = lambda T, d: torch.empty((R, T, d), dtype=torch.float32, device=dev) if d > 0 else None
        eo, no, go = mk(g.n_edges, oe), mk(g.n_nodes, on), mk(g.n_graphs, og)
        with torch.cuda.device(dev):
            ws = g.workspace(lib.gnx_block_workspace_bytes(g._h, C.byref(p), R), ("block", block.in_dims, block.out_dims, R))
            check(lib.gnx_block_forward(g._h, C.byref(p), _ptr(ef), _ptr(nf), _ptr(gf), R, _ptr(eo), _ptr(no), _ptr(go), ws.data_ptr(),
                                        ws.numel(), flags, torch.cuda.current_stream(dev).cuda_stream))
        ctx.block, ctx.g, ctx.R = block, g, R
        # tensors go through save_for_backward (no ctx -> output -> grad_fn -> ctx cycle; in-place modification is detected);
        # ctx keeps only which of the six slots were present
        six = (ef, nf, gf, eo, no, go)
        ctx.slots = tuple(t is not None for t in six)
        ctx.save_for_backward(*[t for t in six if t is not None])
        outs = tuple(o for o in (eo, no, go) if o is not None)
        ctx.present = tuple(o is not None for o in (eo, no, go))
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        lib = _lib.load()
        block, g, R = ctx.block, ctx.g, ctx.R
        sv = iter(ctx.saved_tensors)
        ef, nf, gf, eo, no, go = (next(sv) if present else None for present in ctx.slots)
        it = iter(gouts)
        ge, gn_, gg = (next(it) if pr else None for pr in ctx.present)
        cont = lambda t: None if t is None else t.contiguous()
        ge, gn_, gg = cont(ge), cont(gn_), cont(gg)
        keep = []
        p = block._c(keep)
        dev = g.device
        need = ctx.needs_input_grad[1:]  # (g, R, flags, ef, nf, gf, We, be, Wn, bn, Wg, bg) -> ef is need[3]
        d_ef = torch.empty_like(ef) if ef is not None and need[3] else None
        d_nf = torch.empty_like(nf) if nf is not None and need[4] else None
        d_gf = torch.empty_like(gf) if gf is not None and need[5] else None
        layers = (block.edgefn, block.nodefn, block.graphfn)
        gW = [torch.empty((l.weight.shape[1], l.weight.shape[0]), dtype=torch.float32, device=dev) if need[6 + 2 * i] else None for i, l in enumerate(layers)]
        gb = [torch.empty_like(l.bias) if (l.bias is not None and need[7 + 2 * i]) else None for i, l in enumerate(layers)]
        grads = _lib.BlockGrads(*[_lib.DenseGrad(_ptr(w) if (w is not None and w.numel()) else None, _ptr(b) if (b is not None and b.numel()) else None)
                                  for w, b in zip(gW, gb)])
        with torch.cuda.device(dev):
            nb = lib.gnx_block_backward_workspace_bytes(g._h, C.byref(p), R)
            ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device=dev)
            check(lib.gnx_block_backward(g._h, C.byref(p), _ptr(ef), _ptr(nf), _ptr(gf), _ptr(eo), _ptr(no), _ptr(go), _ptr(ge), _ptr(gn_),
                                         _ptr(gg), R, _ptr(d_ef), _ptr(d_nf), _ptr(d_gf), C.byref(grads), ws.data_ptr(), ws.numel(),
                                         torch.cuda.current_stream(dev).cuda_stream))
        gWt = [None if w is None else w.t() for w in gW]  # (out, in) view with column-major storage, like the weights
        return (None, None, None, None, d_ef, d_nf, d_gf, gWt[0], gb[0], gWt[1], gb[1], gWt[2], gb[2])


class _ChainBlockFn(torch.autograd.Function):
    """autograd node of a GNBlock whose update functions are Chains: forward = gnx_chain_block_forward, backward =
    gnx_chain_block_backward (which recomputes every layer's output from the inputs)."""

    @staticmethod
    def forward(ctx, block, g, R, flags, ef, nf, gf, *params):
        eo, no, go = block._chain_forward(g, R, flags, ef, nf, gf)
        ctx.block, ctx.g, ctx.R = block, g, R
        ctx.slots = tuple(t is not None for t in (ef, nf, gf))
        ctx.save_for_backward(*[t for t in (ef, nf, gf) if t is not None])
        ctx.present = tuple(o is not None for o in (eo, no, go))
        return tuple(o for o in (eo, no, go) if o is not None)

    @staticmethod
    def backward(ctx, *gouts):
        lib = _lib.load()
        block, g, R = ctx.block, ctx.g, ctx.R
        sv = iter(ctx.saved_tensors)
        ef, nf, gf = (next(sv) if present else None for present in ctx.slots)
        it = iter(gouts)
        ge, gn_, gg = (next(it) if pr else None for pr in ctx.present)
        cont = lambda t: None if t is None else t.contiguous()
        ge, gn_, gg = cont(ge), cont(gn_), cont(gg)
        keep = []
        p, chains, _ = block._chain_params(keep)
        dev = g.device
        need = ctx.needs_input_grad  # (block, g, R, flags, ef, nf, gf, W, b, W, b, ...)
        d_ef = torch.empty_like(ef) if ef is not None and need[4] else None
        d_nf = torch.empty_like(nf) if nf is not None and need[5] else None
        d_gf = torch.empty_like(gf) if gf is not None and need[6] else None
        gW, gb, arrays, k = [], [], [], 7
        for ch in chains:
            entries = []
            for l in ch.layers:
                ln = isinstance(l, LayerNorm)  # (its "weight" slot is gamma [d])
                w = (torch.empty_like(l.gamma) if ln else torch.empty((l.weight.shape[1], l.weight.shape[0]), dtype=torch.float32, device=dev)) if need[k] else None
                b = torch.empty_like(l.bias) if (l.bias is not None and need[k + 1]) else None
                k += 2
                gW.append(w); gb.append(b)
                entries.append(_lib.DenseGrad(_ptr(w) if (w is not None and w.numel()) else None, _ptr(b) if (b is not None and b.numel()) else None))
            arrays.append((_lib.DenseGrad * max(len(entries), 1))(*entries))
        grads = _lib.ChainBlockGrads(*[C.cast(a, C.POINTER(_lib.DenseGrad)) for a in arrays])
        with torch.cuda.device(dev):
            nb = lib.gnx_chain_block_backward_workspace_bytes(g._h, C.byref(p), R)
            if nb == 0:
                raise GnxError(_lib.ERR_DIMS, lib.gnx_last_error().decode("utf-8", "replace"))
            ws = torch.empty(int(nb), dtype=torch.uint8, device=dev)
            check(lib.gnx_chain_block_backward(g._h, C.byref(p), _ptr(ef), _ptr(nf), _ptr(gf), _ptr(ge), _ptr(gn_), _ptr(gg), R,
                                               _ptr(d_ef), _ptr(d_nf), _ptr(d_gf), C.byref(grads), ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream(dev).cuda_stream))
        out = [None, None, None, None, d_ef, d_nf, d_gf]
        for w, b in zip(gW, gb):
            out += [None if w is None else (w if w.dim() == 1 else w.t()), b]  # (out, in) view with column-major storage, like the weights; a LayerNorm's gamma as it is
        return tuple(out)


class GNFeedForward:
    """gnfeedforward.jl:17-40: per entity `Chain(Dense(d => 4d, relu), Dense(4d => d), Dropout(p))`."""

    def __init__(self, dims, dropout=0, device=None, generator=None):
        assert all(d > 0 for d in dims)  # gnfeedforward.jl:18
        mk = lambda d: (Dense(d, 4 * d, "relu", device, generator), Dense(4 * d, d, "identity", device, generator))
        self.eff, self.nff, self.gff = mk(dims[0]), mk(dims[1]), mk(dims[2])
        self.dropout = dropout


class GNGraphNorm:
    """gngraphnorm.jl:9-17."""

    def __init__(self, dims, device=None):
        assert all(d > 0 for d in dims)  # gngraphnorm.jl:10
        self.edgeln, self.nodeln, self.graphln = (LayerNorm(d, device) for d in dims)


class GNCore:
    """`GNCore(dims; dropout=0)` (gncore.jl:46-54): `core(x) = x + block(gn1(x)) + ffwd(gn2(x))` (gncore.jl:56-59)
    via gnx_core_forward.  `dropout = p`: the Dropout(p) ending each FeedForward (gnfeedforward.jl:27-31) is active inside a gradient call
    (gnx_core_forward_train / gnx_core_backward_train, a fresh seed per call) and the identity otherwise — Flux's automatic mode;
    `testmode(core)` / `trainmode(core)` force it as `Flux.testmode!` / `trainmode!` do.
    `eps_mode` 0 = Flux 0.14 `normalise` (x-μ)/(σ+ε); 1 = (x-μ)/sqrt(σ²+ε)."""

    def __init__(self, dims, dropout=0, device=None, generator=None, eps=1e-5, eps_mode=0):
        dims = tuple(int(d) for d in dims)
        assert any(d > 0 for d in dims)  # gncore.jl:47
        self.dims = dims
        self.block = GNBlock(dims, dims, dropout=dropout, device=device, generator=generator)
        self.ffwd = GNFeedForward(dims, dropout=dropout, device=device, generator=generator)
        self.gn1 = GNGraphNorm(dims, device)
        self.gn2 = GNGraphNorm(dims, device)
        self.eps, self.eps_mode = eps, eps_mode
        self.flags = 0

    def _c(self, keep):
        p = _lib.CoreParams()
        p.block = self.block._c(keep)
        for i, (l1, l2, ff) in enumerate(zip((self.gn1.edgeln, self.gn1.nodeln, self.gn1.graphln),
                                             (self.gn2.edgeln, self.gn2.nodeln, self.gn2.graphln),
                                             (self.ffwd.eff, self.ffwd.nff, self.ffwd.gff))):
            p.ln1[i], p.ln2[i] = l1._c(keep), l2._c(keep)
            p.ff[i].fc1, p.ff[i].fc2 = ff[0]._c(keep), ff[1]._c(keep)
        p.eps, p.eps_mode = self.eps, self.eps_mode
        q = getattr(self, "_prepared", None)
        if q is not None:  # (see GNBlock._c)
            p.prepared = q.current()
        return p

    def prepare(self):
        """`gnx_core_prepare`: this core's weight blocks (block and FeedForwards) in the forms the matrix-core kernels stage, made ONCE.
        Forwards outside autograd then launch no preparation kernel; results are bit-identical.  Returns self."""
        self._prepared = None
        self.block._prepared = None  # (the core's object covers its block)
        keep = []
        p = self._c(keep)
        dev = self.block.edgefn.weight.device
        h = C.c_void_p()
        with torch.cuda.device(dev):
            check(_lib.load().gnx_core_prepare(C.byref(p), torch.cuda.current_stream(dev).cuda_stream, C.byref(h)))
        # every parameter of the core is a source: the block's weights (edge, projection AND node planes), the FeedForwards', and the LayerNorm
        # scales / shifts and biases the one-launch form folds into its planes
        self._prepared = _Prepared(h, dev, self._param_list(), keep)
        return self

    def _param_list(self):
        """Order = gnx_core_grads: block (edge, node, graph: weight, bias), ln1 x3 (gamma, beta), ln2 x3, ff x3 (W1, b1, W2, b2)."""
        ps = []
        for l in (self.block.edgefn, self.block.nodefn, self.block.graphfn):
            ps += [l.weight, l.bias]
        for gnorm in (self.gn1, self.gn2):
            for ln in (gnorm.edgeln, gnorm.nodeln, gnorm.graphln):
                ps += [ln.gamma, ln.beta]
        for ff in (self.ffwd.eff, self.ffwd.nff, self.ffwd.gff):
            ps += [ff[0].weight, ff[0].bias, ff[1].weight, ff[1].bias]
        return ps

    def parameters(self):
        return self._param_list()

    def _training(self, extra=()):
        return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in list(extra) + self._param_list())

    def _dropout_now(self, grad_call):
        """The Dropout(p) that ends every FeedForward chain (gnfeedforward.jl:27-31) for THIS call: None (identity) or a `_lib.Dropout` with a
        fresh seed.  Flux's rule: active inside a gradient call, identity otherwise, unless `testmode` / `trainmode` forced it."""
        p = float(self.ffwd.dropout or 0)
        forced = getattr(self, "_dropout_mode", None)
        if p <= 0 or not (grad_call if forced is None else forced):
            return None
        assert p <= 1, "Dropout: p must lie in [0, 1]"
        seed = int(torch.randint(0, 2 ** 63 - 1, (1,), dtype=torch.int64, generator=getattr(self, "rng", None)).item())  # torch.manual_seed reproduces it
        self.last_dropout = _lib.Dropout(p, 0, seed)  # (tests / debugging: `dropout_mask` turns it into the masks of the call)
        return self.last_dropout

    def __call__(self, x, flags=None):
        # gnfeedforward.jl:27-31: the FeedForward ends in Dropout(p), which Flux applies in training mode (inside a gradient call)
        # and skips in test mode: gnx_core_forward_train / gnx_core_backward_train with a per-call seed, gnx_core_forward otherwise.
        x = _as_nt(x)
        assert x.ef is not None and x.nf is not None and x.gf is not None, "GNCore needs ef, nf and gf (gncore.jl:61-68)"
        g, ef, nf, gf, R = _forward_common(x, self.dims)
        plist = self._param_list()
        grad_call = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in [ef, nf, gf] + plist)
        drop = self._dropout_now(grad_call)
        if grad_call:
            eo, no, go = _CoreFn.apply(self, g, R, self.flags if flags is None else flags, drop, ef, nf, gf, *plist)
            return NT(g, _jl(eo), _jl(no), _jl(go))
        eo, no, go = _core_forward(self, g, R, self.flags if flags is None else flags, drop, ef, nf, gf)
        return NT(g, _jl(eo), _jl(no), _jl(go))


def _core_forward(core, g, R, flags, drop, ef, nf, gf):
    """gnx_core_forward, or gnx_core_forward_train when a Dropout is active for the call."""
    lib = _lib.load()
    keep = []
    p = core._c(keep)
    dev = g.device
    eo, no, go = torch.empty_like(ef), torch.empty_like(nf), torch.empty_like(gf)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        if drop is None:
            ws = g.workspace(lib.gnx_core_workspace_bytes(g._h, C.byref(p), R), ("core", core.dims, R))
            check(lib.gnx_core_forward(g._h, C.byref(p), ef.data_ptr(), nf.data_ptr(), gf.data_ptr(), R, eo.data_ptr(), no.data_ptr(),
                                       go.data_ptr(), ws.data_ptr(), ws.numel(), flags, stream))
        else:
            nb = lib.gnx_core_train_workspace_bytes(g._h, C.byref(p), R)
            if nb == 0:
                raise GnxError(_lib.ERR_DIMS, lib.gnx_last_error().decode("utf-8", "replace"))
            ws = g.workspace(nb, ("core_train", core.dims, R))
            check(lib.gnx_core_forward_train(g._h, C.byref(p), C.byref(drop), ef.data_ptr(), nf.data_ptr(), gf.data_ptr(), R, eo.data_ptr(),
                                             no.data_ptr(), go.data_ptr(), ws.data_ptr(), ws.numel(), flags, stream))
    return eo, no, go


def dropout_mask(drop, entity, shape, device=None):
    """The mask `gnx_core_forward_train` applied to entity 0 / 1 / 2 (edges / nodes / graphs) of a call whose Dropout was `drop`
    (`core.last_dropout`): a (D, T, R) array of 0 and 1 / (1 - p), the layout of the features (gnx_dropout_mask)."""
    dev = _device(device)
    d, T, R = (int(v) for v in shape)
    out = torch.empty((R, T, d), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().gnx_dropout_mask(C.byref(drop), int(entity), out.numel(), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return _jl(out)


def testmode(m, mode=True):
    """`Flux.testmode!(m, mode)`: True = Dropout is the identity whatever the call; False = always active; None = automatic (active inside a
    gradient call).  Applies to a GNCore, a GNCoreList or any iterable of layers; returns m."""
    forced = None if mode is None else (not mode)
    for l in (m.list if isinstance(m, GNCoreList) else (m if isinstance(m, (list, tuple)) else [m])):
        if isinstance(l, GNCore):
            l._dropout_mode = forced
        elif isinstance(l, (GNCoreList, list, tuple)):
            testmode(l, mode)
    return m


def trainmode(m, mode=True):
    """`Flux.trainmode!(m, mode)` = testmode!(m, !mode)."""
    return testmode(m, None if mode is None else (not mode))


class _CoreFn(torch.autograd.Function):
    """torch autograd node of one GNCore call: forward = gnx_core_forward, backward = gnx_core_backward."""

    @staticmethod
    def forward(ctx, core, g, R, flags, drop, ef, nf, gf, *params):
        eo, no, go = _core_forward(core, g, R, flags, drop, ef, nf, gf)
        ctx.core, ctx.g, ctx.R, ctx.drop = core, g, R, drop
        ctx.save_for_backward(ef, nf, gf)
        return eo, no, go

    @staticmethod
    def backward(ctx, ge, gn_, gg):
        lib = _lib.load()
        core, g, R = ctx.core, ctx.g, ctx.R
        ef, nf, gf = ctx.saved_tensors
        dev = g.device
        cont = lambda t: None if t is None else t.contiguous()
        ge, gn_, gg = cont(ge), cont(gn_), cont(gg)
        keep = []
        p = core._c(keep)
        plist = core._param_list()
        out = [torch.empty((q.shape[1], q.shape[0]), dtype=torch.float32, device=dev).t() if q.dim() == 2 else torch.empty_like(q) for q in plist]
        gr = _lib.CoreGrads()
        wptr = lambda t: t.t().data_ptr() if t.dim() == 2 else t.data_ptr()  # the (in, out)-contiguous storage under the (out, in) view
        it = iter(out)
        for dst in (gr.block.edgefn, gr.block.nodefn, gr.block.graphfn):
            dst.weight, dst.bias = wptr(next(it)), wptr(next(it))
        for arr in (gr.ln1, gr.ln2):
            for i in range(3):
                arr[i].gamma, arr[i].beta = wptr(next(it)), wptr(next(it))
        for i in range(3):
            gr.ff[i].fc1.weight, gr.ff[i].fc1.bias = wptr(next(it)), wptr(next(it))
            gr.ff[i].fc2.weight, gr.ff[i].fc2.bias = wptr(next(it)), wptr(next(it))
        d_ef, d_nf, d_gf = torch.empty_like(ef), torch.empty_like(nf), torch.empty_like(gf)
        with torch.cuda.device(dev):
            nb = lib.gnx_core_backward_workspace_bytes(g._h, C.byref(p), R)
            ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device=dev)
            if ctx.drop is None:
                check(lib.gnx_core_backward(g._h, C.byref(p), ef.data_ptr(), nf.data_ptr(), gf.data_ptr(), _ptr(ge), _ptr(gn_), _ptr(gg), R,
                                            d_ef.data_ptr(), d_nf.data_ptr(), d_gf.data_ptr(), C.byref(gr), ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream(dev).cuda_stream))
            else:  # the forward's masks, regenerated from the call's seed
                check(lib.gnx_core_backward_train(g._h, C.byref(p), C.byref(ctx.drop), ef.data_ptr(), nf.data_ptr(), gf.data_ptr(), _ptr(ge), _ptr(gn_),
                                                  _ptr(gg), R, d_ef.data_ptr(), d_nf.data_ptr(), d_gf.data_ptr(), C.byref(gr), ws.data_ptr(), ws.numel(),
                                                  torch.cuda.current_stream(dev).cuda_stream))
        return (None, None, None, None, None, d_ef, d_nf, d_gf, *out)


class GNCoreList:
    """`GNCoreList(list)` (gncorelist.jl:37-45): left fold of the cores over x."""

    def __init__(self, cores):
        self.list = list(cores)

    def __call__(self, x):
        for fn in self.list:
            x = fn(x)
        return x


class BlockPlan:
    """A pre-bound `gnx_block_forward` call: parameter struct, workspace and handle are fixed, so one step is ONE
    ctypes call on packed [R][T][D] tensors — what hipGraph capture and the bench loop want.  No allocation here."""

    def __init__(self, block, g, R=1, flags=None):
        self.block, self.g, self.R = block, g, int(R)
        self.flags = block.flags if flags is None else flags
        self._keep = []
        self.p = block._c(self._keep)
        self.lib = _lib.load()
        with torch.cuda.device(g.device):
            nbytes = self.lib.gnx_block_workspace_bytes(g._h, C.byref(self.p), self.R)
        self.ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=g.device)

    def outputs(self):
        oe, on, og = self.block.out_dims
        g, R = self.g, self.R
        mk = lambda T, d: torch.empty((R, T, d), dtype=torch.float32, device=g.device) if d > 0 else None
        return mk(g.n_edges, oe), mk(g.n_nodes, on), mk(g.n_graphs, og)

    def __call__(self, ef, nf, gf, eo, no, go, stream=None, ws=None, defer_graph_update=False):
        """`defer_graph_update`: stop after the edge + node update (GNX_FLAG_DEFER_GRAPH_UPDATE); finish with
        `graph_update` — typically on a second stream so that it overlaps the next batch."""
        s = torch.cuda.current_stream(self.g.device).cuda_stream if stream is None else stream
        ws = self.ws if ws is None else ws
        flags = self.flags | (_lib.FLAG_DEFER_GRAPH_UPDATE if defer_graph_update else 0)
        check(self.lib.gnx_block_forward(self.g._h, C.byref(self.p), _ptr(ef), _ptr(nf), _ptr(gf), self.R, _ptr(eo), _ptr(no),
                                         _ptr(go), ws.data_ptr(), ws.numel(), flags, s))

    def chained(self, ef, nf, gf, eo, no, go, ws, prev=None, stream=None):
        """`gnx_block_forward_chained`: this call's edge + node update with the graph update of the PREVIOUS chained call (`prev`: the
        record that call returned, or None) at the front of the same launch.  Returns this call's pending record; `go` is valid after
        the next chained call on the stream or `flush(pending)`.  `ws` / `go` must differ from the pending call's."""
        s = torch.cuda.current_stream(self.g.device).cuda_stream if stream is None else stream
        pending = _lib.PendingUpdate()
        check(self.lib.gnx_block_forward_chained(self.g._h, C.byref(self.p), _ptr(ef), _ptr(nf), _ptr(gf), self.R, _ptr(eo), _ptr(no), _ptr(go),
                                                 ws.data_ptr(), ws.numel(), self.flags, s, C.byref(prev) if prev is not None else None, C.byref(pending)))
        return pending

    def steps(self, sets, stream=None):
        """`gnx_block_forward_steps`: a LOOP over batches as one call — `sets` is a sequence of dicts / tuples (ef, nf, gf, eo, no, go, ws), one per
        step, in order.  Exactly `len(sets)` forwards (bit-identical outputs); where the two-launch narrow form runs, step i's graph update rides at
        the front of step i + 1's launch and the last one is flushed inside the call: every output is complete when the enqueued work is.
        Consecutive steps must use different workspaces and gf outputs to be chained (else the step simply runs unchained)."""
        s = torch.cuda.current_stream(self.g.device).cuda_stream if stream is None else stream
        arr = (_lib.BlockStep * max(len(sets), 1))()
        for i, b in enumerate(sets):
            ef, nf, gf, eo, no, go, ws = (b["ef"], b["nf"], b["gf"], *b["out"], b["ws"]) if isinstance(b, dict) else b
            arr[i] = _lib.BlockStep(_ptr(ef), _ptr(nf), _ptr(gf), _ptr(eo), _ptr(no), _ptr(go), ws.data_ptr(), ws.numel())
        check(self.lib.gnx_block_forward_steps(self.g._h, C.byref(self.p), arr, len(sets), self.R, self.flags, s))

    def flush(self, pending, stream=None):
        """finishes a pending graph update (one plain `gnx_block_graph_update` launch); no-op when nothing is pending"""
        if pending is None or not pending.workspace:
            return
        s = torch.cuda.current_stream(self.g.device).cuda_stream if stream is None else stream
        check(self.lib.gnx_block_graph_update(self.g._h, C.byref(self.p), pending.gf, self.R, pending.gf_out, pending.workspace, pending.workspace_bytes,
                                              self.flags, s))

    def new_workspace(self):
        """A further workspace (one per buffer set when steps on different sets may overlap)."""
        return torch.empty_like(self.ws)

    def graph_update(self, gf, go, stream=None, ws=None):
        s = torch.cuda.current_stream(self.g.device).cuda_stream if stream is None else stream
        ws = self.ws if ws is None else ws
        check(self.lib.gnx_block_graph_update(self.g._h, C.byref(self.p), _ptr(gf), self.R, _ptr(go), ws.data_ptr(), ws.numel(),
                                              self.flags, s))


class Model:
    """A chain of GNBlock / GNCore layers as ONE hipGraph inside libgnx (`gnx_model_*`, include/gnx.h) — what a Julia or C
    caller uses instead of `Graphed` (which needs torch's graph capture).  `m = Model([enc, core1, core2, dec], x)`;
    `y = m(x2)`: feature values are copied into the model's static input buffers, the whole chain is replayed with one
    hipGraphLaunch, the returned tuple aliases the model's static output buffers (valid until the next call)."""

    def __init__(self, layers, x, flags=0):
        x = _as_nt(x)
        self.g = x.graphs
        self.layers = list(layers)
        self.flags = flags
        lib = _lib.load()
        dev = self.g.device
        present = [a for a in (x.ef, x.nf, x.gf) if a is not None]
        self.R = present[0].shape[2]
        self._keep = []
        self._params = [l._c(self._keep) for l in self.layers]
        arr = (_lib.Layer * len(self.layers))()
        for i, (l, p) in enumerate(zip(self.layers, self._params)):
            arr[i].kind = 1 if isinstance(l, GNCore) else 0
            arr[i].params = C.addressof(p)
        h = C.c_void_p()
        with torch.cuda.device(dev):
            check(lib.gnx_model_create(self.g._h, arr, len(self.layers), self.R, C.byref(h)))
        self.handle = h
        dims = (C.c_int32 * 3)()
        check(lib.gnx_model_out_dims(h, dims))
        rows = (self.g.n_edges, self.g.n_nodes, self.g.n_graphs)
        self._in = {k: None if getattr(x, k) is None else _packed(getattr(x, k)).clone() for k in ("ef", "nf", "gf")}
        self._out = [torch.empty((self.R, t, int(d)), dtype=torch.float32, device=dev) if d > 0 else None for t, d in zip(rows, dims)]

    def __call__(self, x):
        x = _as_nt(x)
        assert x.graphs is self.g, "a Model is bound to the GNGraphBatch it was created with"
        for k in ("ef", "nf", "gf"):
            a, buf = getattr(x, k), self._in[k]
            assert (a is None) == (buf is None), f"{k}: presence differs from the creating call"
            if a is not None:
                buf.copy_(a.permute(2, 1, 0))
        dev = self.g.device
        with torch.cuda.device(dev):
            check(_lib.load().gnx_model_forward(self.handle, _ptr(self._in["ef"]), _ptr(self._in["nf"]), _ptr(self._in["gf"]), _ptr(self._out[0]),
                                                _ptr(self._out[1]), _ptr(self._out[2]), self.flags, torch.cuda.current_stream(dev).cuda_stream))
        return NT(self.g, *(_jl(o) for o in self._out))

    def refresh_weights(self):
        """`gnx_model_refresh_weights`: the model prepared its layers' weight blocks when it was created; after the weights' VALUES changed in
        place (an optimiser step) call this before the next forward."""
        dev = self.g.device
        with torch.cuda.device(dev):
            check(_lib.load().gnx_model_refresh_weights(self.handle, torch.cuda.current_stream(dev).cuda_stream))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().gnx_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Graphed:
    """A model forward captured ONCE into a hipGraph and replayed: `g = Graphed(model, x); y = g(x2)`.

    At README-sized widths a GNBlock is ~25 us of GPU work while an eager call costs more than that on the host, so a
    multi-layer model (README ex.3: encoder -> GNCoreList -> decoder) is launch-bound; replaying one graph removes the
    per-layer host work.  Every `gnx_*` forward is capture-safe (no allocation, no synchronisation).  `model` is any
    callable on the batched tuple; the graph structure (the GNGraphBatch) and the feature shapes are fixed at capture,
    feature VALUES are copied into the captured input buffers at each call; the returned tuple aliases the captured output
    buffers (valid until the next call)."""

    def __init__(self, model, x, warmup=2):
        x = _as_nt(x)
        self.g = x.graphs
        dev = self.g.device
        self._in = {k: None if getattr(x, k) is None else _packed(getattr(x, k)).clone() for k in ("ef", "nf", "gf")}
        static = NT(self.g, *(_jl(self._in[k]) for k in ("ef", "nf", "gf")))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up off the capture: loads code objects, sizes the workspace
            for _ in range(max(warmup, 1)):
                model(static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # (thread_local: a "global" capture turns the event queries of torch's NCCL watchdog thread — any process with a process group — into errors
        #  that abort the process: profiles/r06_capture_vs_watchdog.log)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._out = model(static)

    def __call__(self, x):
        x = _as_nt(x)
        assert x.graphs is self.g, "a Graphed model is bound to the GNGraphBatch it was captured with"
        for k in ("ef", "nf", "gf"):
            a, buf = getattr(x, k), self._in[k]
            assert (a is None) == (buf is None), f"{k}: presence differs from the captured call"
            if a is not None:
                buf.copy_(a.permute(2, 1, 0))
        self.graph.replay()
        return self._out

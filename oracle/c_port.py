"""ctypes binding of oracle/gn_oracle_c.c (TEST INFRASTRUCTURE ONLY; see that file's header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libgn_oracle.so")
_lib = None

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int64)


class BlockParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("de", "dn", "dg", "oe", "on", "og")] + \
               [(n, _fp) for n in ("We", "be", "Wn", "bn", "Wg", "bg")] + \
               [(n, C.c_int) for n in ("act_e", "act_n", "act_g")]


class CoreParams(C.Structure):
    _fields_ = [("block", BlockParams)] + \
               [(n, _fp * 3) for n in ("ln1_gamma", "ln1_beta", "ln2_gamma", "ln2_beta", "W1", "b1", "W2", "b2")] + \
               [("eps", C.c_float), ("eps_mode", C.c_int)]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_DIR, "gn_oracle_c.c")):
        subprocess.check_call(["make", "-s", "-C", _DIR])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.gn_oracle_max_threads.restype = C.c_int
    return _lib


def _f(a, keep):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    keep.append(a)
    return a.ctypes.data_as(_fp)


def _wcol(W, keep):
    """(out, in) numpy matrix → column-major buffer (Flux Dense.weight bytes)."""
    a = np.asfortranarray(W, dtype=np.float32)
    keep.append(a)
    return a.ctypes.data_as(_fp)


def _block_struct(p, keep):
    de, dn, dg = p["in_dims"]
    oe, on, og = p["out_dims"]
    return BlockParams(de, dn, dg, oe, on, og, _wcol(p["We"], keep), _f(p["be"], keep), _wcol(p["Wn"], keep),
                       _f(p["bn"], keep), _wcol(p["Wg"], keep), _f(p["bg"], keep), p["act_e"], p["act_n"], p["act_g"])


def _csc_ptrs(csc, keep):
    out = []
    for a in csc:
        a = np.ascontiguousarray(a, dtype=np.int64)
        keep.append(a)
        out.append(a.ctypes.data_as(_ip))
    return out


def _run(fn, pstruct, csc, ef, nf, gf, out_dims, nthreads, keep):
    colptr, rowval, node_off, edge_off = csc
    N, E, G = len(colptr) - 1, len(rowval), len(node_off) - 1
    R = next(a.shape[0] for a in (ef, nf, gf) if a is not None)
    oe, on, og = out_dims
    eo = np.zeros((R, E, oe), dtype=np.float32)
    no = np.zeros((R, N, on), dtype=np.float32)
    go = np.zeros((R, G, og), dtype=np.float32)
    ptrs = _csc_ptrs(csc, keep)
    rc = fn(C.c_int64(N), C.c_int64(E), C.c_int64(G), *ptrs, C.byref(pstruct), _f(ef, keep), _f(nf, keep), _f(gf, keep),
            C.c_int64(R), eo.ctypes.data_as(_fp), no.ctypes.data_as(_fp), go.ctypes.data_as(_fp), C.c_int(nthreads))
    assert rc == 0
    return tuple(None if d == 0 else a for d, a in zip(out_dims, (eo, no, go)))


def block_forward(p, csc, ef, nf, gf, nthreads=0):
    keep = []
    return _run(lib().gn_oracle_block_forward_f32, _block_struct(p, keep), csc, ef, nf, gf, p["out_dims"], nthreads, keep)


def core_forward(p, csc, ef, nf, gf, nthreads=0):
    keep = []
    cp = CoreParams()
    cp.block = _block_struct(p["block"], keep)
    for i, t in enumerate("eng"):
        cp.ln1_gamma[i] = _f(p[f"ln1_{t}_gamma"], keep); cp.ln1_beta[i] = _f(p[f"ln1_{t}_beta"], keep)
        cp.ln2_gamma[i] = _f(p[f"ln2_{t}_gamma"], keep); cp.ln2_beta[i] = _f(p[f"ln2_{t}_beta"], keep)
        cp.W1[i] = _wcol(p[f"ff_{t}_W1"], keep); cp.b1[i] = _f(p[f"ff_{t}_b1"], keep)
        cp.W2[i] = _wcol(p[f"ff_{t}_W2"], keep); cp.b2[i] = _f(p[f"ff_{t}_b2"], keep)
    cp.eps, cp.eps_mode = p["eps"], p["eps_mode"]
    return _run(lib().gn_oracle_core_forward_f32, cp, csc, ef, nf, gf, p["dims"], nthreads, keep)


def max_threads():
    return lib().gn_oracle_max_threads()


class BlockRunner:
    """Pre-bound call of the C restatement for timing (bench.py's cpu_baseline leg): buffers allocated and touched once."""

    def __init__(self, p, csc, ef, nf, gf, nthreads=0):
        self.keep = []
        self.ps = _block_struct(p, self.keep)
        colptr, rowval, node_off, edge_off = csc
        self.N, self.E, self.G = len(colptr) - 1, len(rowval), len(node_off) - 1
        self.R = next(a.shape[0] for a in (ef, nf, gf) if a is not None)
        oe, on, og = p["out_dims"]
        self.out = [np.zeros((self.R, T, d), dtype=np.float32) for T, d in ((self.E, oe), (self.N, on), (self.G, og))]
        self.args = (C.c_int64(self.N), C.c_int64(self.E), C.c_int64(self.G), *_csc_ptrs(csc, self.keep), C.byref(self.ps),
                     _f(ef, self.keep), _f(nf, self.keep), _f(gf, self.keep), C.c_int64(self.R),
                     *[o.ctypes.data_as(_fp) for o in self.out], C.c_int(nthreads))
        self.fn = lib().gn_oracle_block_forward_f32

    def run(self):
        assert self.fn(*self.args) == 0
        return self.out

"""ctypes binding of libgnx.so — exactly the symbols include/gnx.h declares.

There is no fallback of any kind: if the shared library is missing or a symbol cannot be resolved the import
fails loudly; if no GPU is visible every compute entry point returns a HIP error that is raised as GnxError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNX_LIB_PATH") or os.path.join(_HERE, "libgnx.so")  # GNX_LIB_PATH: A/B runs of two builds in one session

# status codes (include/gnx.h)
GNX_OK = 0
ERR_INVALID_ARG, ERR_NO_GRAPHS, ERR_ADJ_SHAPE, ERR_ADJ_VALUE, ERR_ALL_NOTHING = -1, -2, -3, -4, -5
ERR_DIMS, ERR_CSC, ERR_WORKSPACE, ERR_TOO_LARGE, ERR_COUNT_MISMATCH = -6, -7, -8, -9, -10
ACT = dict(identity=0, relu=1, tanh=2, sigmoid=3, gelu=4)
ELEM_U8, ELEM_I32, ELEM_I64, ELEM_F32, ELEM_F64 = 0, 1, 2, 3, 4
FLAG_FORCE_GENERIC, FLAG_NO_MFMA, FLAG_DEFER_GRAPH_UPDATE, FLAG_NO_GRAPH, FLAG_DIST_NO_GATHER = 0x1, 0x2, 0x4, 0x8, 0x10
# forms of the forward selected per call (include/gnx.h); the environment variables of the same names (GNX_FFN_FP32=1 ...) are the process-wide
# defaults, read once by the library
FLAG_FFN_FP32, FLAG_EDGE_FP32, FLAG_PROJ_FP32, FLAG_EDGE_NARROW_FP32 = 0x20, 0x40, 0x80, 0x100
FLAG_FP32_MFMA = FLAG_FFN_FP32 | FLAG_EDGE_FP32
FLAG_NO_LN_FUSE, FLAG_LN_STATS_PASS, FLAG_CORE_EDGE_SPLIT, FLAG_NO_FORK = 0x200, 0x400, 0x800, 0x1000
FLAG_NO_PACK, FLAG_NO_FFE, FLAG_NO_JIT, FLAG_EDGE_N, FLAG_LN_ON_LOAD = 0x2000, 0x4000, 0x8000, 0x10000, 0x20000

_fp = C.c_void_p  # device float*


class GraphsInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_graphs", "n_nodes", "n_edges", "node_block_size", "edge_block_size",
                                         "n_tiles", "max_in_degree")] + [("device", C.c_int32), ("reserved", C.c_int32)]


class Dense(C.Structure):
    _fields_ = [("weight", _fp), ("bias", _fp), ("act", C.c_int32), ("kind", C.c_int32)]  # kind: LAYER_DENSE, or LAYER_LAYERNORM inside a Chain (weight = gamma, bias = beta)


LAYER_DENSE, LAYER_LAYERNORM, LAYER_LN_SQRT_EPS = 0, 1, 0x100


class BlockParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("de", "dn", "dg", "oe", "on", "og")] + \
               [("edgefn", Dense), ("nodefn", Dense), ("graphfn", Dense), ("prepared", C.c_void_p)]


class Chain(C.Structure):
    _fields_ = [("layers", C.POINTER(Dense)), ("widths", C.POINTER(C.c_int32)), ("n_layers", C.c_int32), ("reserved", C.c_int32)]


class ChainBlockParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("de", "dn", "dg", "reserved")] + [("edgefn", Chain), ("nodefn", Chain), ("graphfn", Chain)]


class DenseGrad(C.Structure):
    _fields_ = [("weight", _fp), ("bias", _fp)]


class ChainBlockGrads(C.Structure):
    _fields_ = [("edgefn", C.POINTER(DenseGrad)), ("nodefn", C.POINTER(DenseGrad)), ("graphfn", C.POINTER(DenseGrad))]


class BlockGrads(C.Structure):
    _fields_ = [("edgefn", DenseGrad), ("nodefn", DenseGrad), ("graphfn", DenseGrad)]


class LayerNormGrad(C.Structure):
    _fields_ = [("gamma", _fp), ("beta", _fp)]


class FfnGrad(C.Structure):
    _fields_ = [("fc1", DenseGrad), ("fc2", DenseGrad)]


class CoreGrads(C.Structure):
    _fields_ = [("block", BlockGrads), ("ln1", LayerNormGrad * 3), ("ln2", LayerNormGrad * 3), ("ff", FfnGrad * 3)]


class LayerNorm(C.Structure):
    _fields_ = [("gamma", _fp), ("beta", _fp)]


class Ffn(C.Structure):
    _fields_ = [("fc1", Dense), ("fc2", Dense)]


class CoreParams(C.Structure):
    _fields_ = [("block", BlockParams), ("ln1", LayerNorm * 3), ("ln2", LayerNorm * 3), ("ff", Ffn * 3),
                ("eps", C.c_float), ("eps_mode", C.c_int32), ("prepared", C.c_void_p)]


class Dropout(C.Structure):  # gnx_dropout
    _fields_ = [("p", C.c_float), ("reserved", C.c_uint32), ("seed", C.c_uint64)]


class Layer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("params", C.c_void_p)]


class PendingUpdate(C.Structure):
    _fields_ = [("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("gf", _fp), ("gf_out", _fp)]


class BlockStep(C.Structure):  # gnx_block_step: one step of gnx_block_forward_steps
    _fields_ = [("ef", _fp), ("nf", _fp), ("gf", _fp), ("ef_out", _fp), ("nf_out", _fp), ("gf_out", _fp), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double), ("kernels", C.c_int64)]


class GnxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gnx error {code}: {msg}")
        self.code = code


# every exported symbol of include/gnx.h with (restype, argtypes)
_pp = C.POINTER(C.c_void_p)
_i64p = C.POINTER(C.c_int64)
_FWD = [C.c_void_p, C.c_void_p, _fp, _fp, _fp, C.c_int64, _fp, _fp, _fp, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
SIGNATURES = {
    "gnx_version": (C.c_int32, []),
    "gnx_default_flags": (C.c_uint32, []),
    "gnx_last_error": (C.c_char_p, []),
    "gnx_graphs_create_dense": (C.c_int32, [_pp, _i64p, C.c_int64, C.c_int32, C.c_int32, _pp]),
    "gnx_graphs_create_dense_packed": (C.c_int32, [C.c_void_p, C.c_int64, _i64p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _pp]),
    "gnx_graphs_create_csc": (C.c_int32, [_pp, _pp, _i64p, C.c_int64, C.c_int32, _pp]),
    "gnx_graphs_create_csc_packed": (C.c_int32, [_i64p, _i64p, _i64p, C.c_int64, C.c_int32, _pp]),
    "gnx_graphs_create_csc_cat": (C.c_int32, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, _i64p, C.c_int64, C.c_int32, C.c_int32, _pp]),
    "gnx_graphs_destroy": (C.c_int32, [C.c_void_p]),
    "gnx_graphs_get_info": (C.c_int32, [C.c_void_p, C.POINTER(GraphsInfo)]),
    "gnx_graphs_get_offsets": (C.c_int32, [C.c_void_p, _i64p, _i64p]),
    "gnx_graphs_get_csc": (C.c_int32, [C.c_void_p, _i64p, _i64p]),
    "gnx_graphs_get_table": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, _i64p]),
    "gnx_block_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(BlockParams), C.c_int64]),
    "gnx_block_forward": (C.c_int32, [C.c_void_p, C.POINTER(BlockParams)] + _FWD[2:]),
    "gnx_chain_block_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(ChainBlockParams), C.c_int64]),
    "gnx_chain_block_forward": (C.c_int32, [C.c_void_p, C.POINTER(ChainBlockParams)] + _FWD[2:]),
    "gnx_chain_block_backward_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(ChainBlockParams), C.c_int64]),
    "gnx_chain_block_backward": (C.c_int32, [C.c_void_p, C.POINTER(ChainBlockParams)] + [_fp] * 6 + [C.c_int64] + [_fp] * 3 +
                                 [C.POINTER(ChainBlockGrads), C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnx_block_forward_chained": (C.c_int32, [C.c_void_p, C.POINTER(BlockParams)] + _FWD[2:] + [C.POINTER(PendingUpdate), C.POINTER(PendingUpdate)]),
    "gnx_block_forward_steps": (C.c_int32, [C.c_void_p, C.POINTER(BlockParams), C.POINTER(BlockStep), C.c_int64, C.c_int64, C.c_uint32, C.c_void_p]),
    "gnx_row_stats": (C.c_int32, [_fp, C.c_int64, C.c_int32, C.c_float, C.c_int32, _fp, C.c_void_p]),
    "gnx_block_graph_update": (C.c_int32, [C.c_void_p, C.POINTER(BlockParams), _fp, C.c_int64, _fp, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "gnx_block_backward_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(BlockParams), C.c_int64]),
    "gnx_block_backward": (C.c_int32, [C.c_void_p, C.POINTER(BlockParams)] + [_fp] * 9 + [C.c_int64] + [_fp] * 3 +
                           [C.POINTER(BlockGrads), C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnx_core_backward_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int64]),
    "gnx_core_backward": (C.c_int32, [C.c_void_p, C.c_void_p] + [_fp] * 6 + [C.c_int64] + [_fp] * 3 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnx_core_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(CoreParams), C.c_int64]),
    "gnx_core_forward": (C.c_int32, [C.c_void_p, C.POINTER(CoreParams)] + _FWD[2:]),
    "gnx_dropout_mask": (C.c_int32, [C.POINTER(Dropout), C.c_int32, C.c_int64, _fp, C.c_void_p]),
    "gnx_core_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(CoreParams), C.c_int64]),
    "gnx_core_forward_train": (C.c_int32, [C.c_void_p, C.POINTER(CoreParams), C.POINTER(Dropout)] + _FWD[2:]),
    "gnx_core_backward_train": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(Dropout)] + [_fp] * 6 + [C.c_int64] + [_fp] * 3 +
                                [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnx_fn_input": (C.c_int32, [C.c_void_p, C.c_int32, _fp, C.c_int32, _fp, C.c_int32, _fp, C.c_int32, C.c_int64, _fp, C.c_void_p]),
    "gnx_collapse_offsets": (C.c_int32, [C.c_void_p, _i64p]),
    "gnx_collapse_edges": (C.c_int32, [C.c_void_p, _fp, C.c_int32, C.c_int64, _fp, C.c_void_p]),
    "gnx_collapse_padded": (C.c_int32, [C.c_void_p, _fp, C.c_int32, C.c_int64, _fp, C.c_void_p]),
    "gnx_xent_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnx_logit_cross_entropy": (C.c_int32, [_fp, _fp, C.c_int32, C.c_int64, _fp, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnx_logit_cross_entropy_backward": (C.c_int32, [_fp, _fp, C.c_int32, C.c_int64, _fp, _fp, C.c_void_p]),
    "gnx_pad_features": (C.c_int32, [C.c_void_p, C.c_int32, _fp, C.c_int32, C.c_int64, _fp, C.c_void_p]),
    "gnx_unpad_features": (C.c_int32, [C.c_void_p, C.c_int32, _fp, C.c_int32, C.c_int64, _fp, C.c_void_p]),
    "gnx_model_create": (C.c_int32, [C.c_void_p, C.POINTER(Layer), C.c_int32, C.c_int64, _pp]),
    "gnx_model_destroy": (C.c_int32, [C.c_void_p]),
    "gnx_model_out_dims": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gnx_model_forward": (C.c_int32, [C.c_void_p] + [_fp] * 6 + [C.c_uint32, C.c_void_p]),
    "gnx_model_refresh_weights": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "gnx_block_prepare": (C.c_int32, [C.POINTER(BlockParams), C.c_void_p, _pp]),
    "gnx_core_prepare": (C.c_int32, [C.POINTER(CoreParams), C.c_void_p, _pp]),
    "gnx_prepared_refresh": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "gnx_prepared_destroy": (C.c_int32, [C.c_void_p]),
    "gnx_prepared_bytes": (C.c_int64, [C.c_void_p]),
    "gnx_dist_partition": (C.c_int32, [_i64p, C.c_int64, C.c_int32, _i64p, _i64p]),
    "gnx_dist_create": (C.c_int32, [C.POINTER(C.c_int32), C.c_int32, _i64p, _i64p, C.c_int64, C.c_int32, _pp]),
    "gnx_dist_destroy": (C.c_int32, [C.c_void_p]),
    "gnx_dist_gather_plan": (C.c_int32, [_i64p, _i64p, C.c_int32, C.c_int64, C.POINTER(C.c_int32), _i64p]),
    "gnx_dist_permute_rows": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "gnx_dist_allgather_gf": (C.c_int32, [C.c_void_p, _pp, _pp, _pp]),
    "gnx_dist_block_forward": (C.c_int32, [C.c_void_p] + [_pp] * 10 + [C.POINTER(C.c_size_t), C.c_uint32, _pp]),
    "gnx_dist_block_forward_steps": (C.c_int32, [C.c_void_p, C.c_int32] + [_pp] * 9 + [C.POINTER(C.c_size_t), C.c_uint32, _pp]),
    "gnx_jit_precompile": (C.c_int32, [C.POINTER(BlockParams), C.c_int32, C.POINTER(C.c_size_t)]),
    "gnx_jit_precompile_core_post": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "gnx_jit_stats": (C.c_int32, [_i64p]),
    "gnx_profile_enable": (C.c_int32, [C.c_int32]),
    "gnx_profile_reset": (C.c_int32, []),
    "gnx_profile_calibrate": (C.c_int32, [C.c_int32, C.c_void_p]),
    "gnx_profile_read": (C.c_int32, [C.POINTER(ProfileEntry), C.c_int32, C.POINTER(C.c_int32)]),
}

_lib = None


def load():
    """Loads libgnx.so.  torch is imported first so that the library's libamdhip64.so.7 dependency resolves to
    the HIP runtime torch already loaded (one runtime per process: streams and pointers are shared)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python graphnets.jl_amd/build.py` "
                          "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    import torch  # noqa: F401  (side effect: loads torch's libamdhip64 with RTLD_GLOBAL-equivalent soname match)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != GNX_OK:
        raise GnxError(rc, load().gnx_last_error().decode("utf-8", "replace"))


def profile_enable(on=True):
    check(load().gnx_profile_enable(1 if on else 0))


def profile_reset():
    check(load().gnx_profile_reset())


def profile_calibrate(n, stream):
    check(load().gnx_profile_calibrate(int(n), stream))


def profile_read():
    lib = load()
    n = C.c_int32(0)
    buf = (ProfileEntry * 64)()
    check(lib.gnx_profile_read(buf, 64, C.byref(n)))
    return {buf[i].name.decode(): dict(launches=buf[i].launches, total_ms=buf[i].total_ms, kernels=buf[i].kernels) for i in range(min(n.value, 64))}

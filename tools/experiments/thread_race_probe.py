#!/usr/bin/env python3
"""Stress form of tests/test_gpu_core.py::test_two_host_threads_run_core_forwards_on_one_handle_concurrently: T threads x ITER forwards on ONE handle,
every result compared with the serial one-stream run; prints what differs (entity, rows, magnitude).   python tools/experiments/thread_race_probe.py [T] [ITER] [ROUNDS]"""
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import graphnets_jl_amd as gn
from oracle import gn_oracle as O
from tests import util as U

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ITER = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
extra = int(os.environ.get("PROBE_FLAGS", "0"))
rng = np.random.default_rng(4900)
dims = (128, 64, 32)
SIZES = ((5000, 60000), (4500, 40000)) if os.environ.get("PROBE_BIG") else ((900, 12000), (300, 2500))  # PROBE_BIG: >= 4096 nodes (projections and node FeedForward six-term too)
graphs = [U.er_csc(rng, n, e) for n, e in SIZES]
g = gn.GNGraphBatch.from_csc([c for c, _ in graphs], [r for _, r in graphs], [n for n, _ in SIZES])
p = O.make_core_params(rng, dims)
core = U.core_from_params(gn, p)
xs = [U.to_nt(gn, g, *U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims)) for _ in range(4)]
tflags = [int(v) for v in os.environ.get("PROBE_THREAD_FLAGS", "").split(",") if v]  # per-thread forms (thread i: tflags[i]); refs per form
refs = {f: [tuple(t.clone() for t in (y.ef, y.nf, y.gf)) for y in (core(x, flags=gn._lib.FLAG_NO_FORK | f) for x in xs)] for f in set(tflags + [extra])}
ref = refs[extra]
torch.cuda.synchronize()
GUARD = int(os.environ.get("PROBE_GUARD_MB", "0")) << 20
guards = []
if GUARD:
    # every workspace of the handle inside its own allocation, between two guard bands holding a pattern: an out-of-bounds write of a forward
    # shows as a changed guard (and no longer lands in another thread's buffers)
    def guarded_workspace(self, nbytes, layout):
        nbytes = max(int(nbytes), 256)
        key = (torch.cuda.current_stream(self.device).cuda_stream, layout, nbytes)
        ws = self._ws.get(key)
        if ws is None:
            big = torch.full((nbytes + 2 * GUARD,), 0x5A, dtype=torch.uint8, device=self.device)
            ws = big[GUARD:GUARD + nbytes]
            guards.append((key, big, nbytes))
            self._ws[key] = ws
        return ws
    gn.GNGraphBatch.workspace = guarded_workspace
OUT_GUARD = int(os.environ.get("PROBE_OUT_GUARD_KB", "0")) << 8  # floats on each side of every output tensor
out_guard_hits = []
if OUT_GUARD:
    _orig_empty_like = torch.empty_like

    def guarded_empty_like(t, *a_, **k_):
        if not (t.is_cuda and t.dtype == torch.float32 and not a_ and not k_ and t.is_contiguous()):
            return _orig_empty_like(t, *a_, **k_)
        big = torch.full((t.numel() + 2 * OUT_GUARD,), 12345.0, dtype=torch.float32, device=t.device)
        return big[OUT_GUARD:OUT_GUARD + t.numel()].view(t.shape)
    torch.empty_like = guarded_empty_like

    def check_out_guards(tid, it, y):
        for name, t in (("ef", y.ef), ("nf", y.nf), ("gf", y.gf)):
            base = t._base if t._base is not None else t
            while base._base is not None:
                base = base._base
            flat = base.flatten()
            if flat.numel() < 2 * OUT_GUARD:
                continue
            lo, hi = flat[:OUT_GUARD], flat[-OUT_GUARD:]
            nlo, nhi = int((lo != 12345.0).sum()), int((hi != 12345.0).sum())
            if nlo or nhi:
                out_guard_hits.append((tid, it, name, nlo, nhi))
bad = []
sigs = {}
LOCK = int(os.environ.get("PROBE_LOCK", "0"))
lock = threading.Lock()
notes = []


OTHER = os.environ.get("PROBE_OTHER", "")  # "copy" / "matmul": every thread but 0 runs unrelated torch work on its stream instead of forwards
stop = threading.Event()


AGG_GRID = int(os.environ.get("PROBE_AGG_GRID", "2048"))   # workgroups of 4 waves per launch (256: one per CU)
AGG_ITERS = int(os.environ.get("PROBE_AGG_ITERS", "24"))  # x 64 matrix instructions per wave and launch


def agg_worker(tid):
    """PROBE_OTHER=agg_bf16 | agg_f32 | agg_bf16lds: the stand-alone matrix-instruction load of tools/experiments/mfma_agg.hip (built to /tmp/libmfma_agg.so)
    as short kernels in a loop on this thread's stream"""
    import ctypes
    lib = ctypes.CDLL(os.environ.get("PROBE_AGG_LIB", "/tmp/libmfma_agg.so"))
    lib.agg_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    kind = {"agg_bf16": 0, "agg_f32": 1, "agg_bf16lds": 2, "agg_bf16hi": 3, "agg_bf16lds73": 4, "agg_cut_edge": 5, "agg_acc_low": 6, "agg_acc_high": 7}[OTHER]
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        while not stop.is_set():
            for _ in range(4):
                assert lib.agg_launch(kind, AGG_ITERS if kind != 1 else AGG_ITERS // 2, AGG_GRID, st.cuda_stream) == 0
            st.synchronize()


def other_worker(tid):
    if OTHER.startswith("agg_"):
        return agg_worker(tid)
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        a = torch.randn(2048, 2048, device="cuda")
        if OTHER == "matmul_bf16":
            a = a.bfloat16()
        b = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        c = torch.empty_like(b)
        while not stop.is_set():
            if OTHER.startswith("matmul"):
                a = (a @ a).clamp_(-1, 1)
            else:
                c.copy_(b)
            st.synchronize()


blk = None
if OTHER.startswith("block"):  # every thread but 0 runs a plain GNBlock at core dims (k_edge_x6 without a LayerNorm; "block_fp32": its fp32 form)
    blk = U.block_from_params(gn, p["block"])
    if OTHER == "block_prepared":
        blk.prepare()


def block_worker(tid):
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    fl = gn._lib.FLAG_FP32_MFMA if OTHER == "block_fp32" else 0
    with torch.cuda.stream(st):
        while not stop.is_set():
            y = blk(xs[1], flags=fl)
            st.synchronize()


def worker(tid):
    if OTHER.startswith("block") and tid > 0:
        return block_worker(tid)
    if OTHER and tid > 0:
        return other_worker(tid)
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    extra = tflags[tid] if tid < len(tflags) else globals()["extra"]
    ref = refs[extra]
    with torch.cuda.stream(st):
        for it in range(ITER):
            k = (tid + 2 * it) % len(xs)
            if LOCK == 1:    # host-side serialisation of the CALLS only: the two streams' kernels still overlap on the GPU
                with lock:
                    y = core(xs[k], flags=extra)
                st.synchronize()
            elif LOCK == 2:  # call + completion under the lock: nothing overlaps (control)
                with lock:
                    y = core(xs[k], flags=extra)
                    st.synchronize()
            else:
                y = core(xs[k], flags=extra)
                st.synchronize()
            if OUT_GUARD:
                check_out_guards(tid, it, y)
            sig = "".join(n_[0] if not torch.equal(a_, b_) else "-" for n_, a_, b_ in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref[k]))
            sigs[sig] = sigs.get(sig, 0) + 1
            for name, a, b in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref[k]):
                if not torch.equal(a, b):
                    d = (a - b).abs()
                    rows = torch.nonzero(d.amax(dim=0).flatten() > 0).flatten()
                    idx = {"ef": 0, "nf": 1, "gf": 2}[name]
                    match = [kk for kk in range(len(xs)) if kk != k and torch.equal(a[:, rows, :], ref[kk][idx][:, rows, :])]
                    bad.append((tid, it, k, name, int(rows.numel()), rows[:8].tolist(), float(d.max()), float(b.abs().max())))
                    cols = torch.nonzero(d[:, rows[0], 0] > 0).flatten().tolist()
                    notes.append(f"    thread {tid} it {it} input {k} {name}: row {int(rows[0])}: {len(cols)} of {d.shape[0]} columns differ: {cols[:40]}; diffs {[round(float(v), 4) for v in (a - b)[:, rows[0], 0][cols[:6]]]}")
                    notes.append(f"    thread {tid} it {it} input {k} {name}: the differing rows equal the reference rows of input(s) {match}" if match else
                                 f"    thread {tid} it {it} input {k} {name}: the differing rows equal no other input's reference")


for r in range(ROUNDS):
    stop.clear()
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    [t.start() for t in ts]
    ts[0].join()
    stop.set()
    [t.join() for t in ts]
print(f"other={OTHER!r} {T} threads x {ITER} forwards x {ROUNDS} rounds, flags {extra:#x}, lock {LOCK}: {len(bad)} mismatching tensors; workspaces {sorted((k[0], v.data_ptr(), v.numel()) for k, v in g._ws.items())[:6]}")
for b in bad[:int(os.environ.get("PROBE_SHOW", "6"))]:
    print("  thread %d it %d input %d %s: %d rows differ (first %s), max |diff| %.3e of %.3e" % b)
for n in notes[:int(os.environ.get("PROBE_SHOW", "6"))]:
    print(n)

for key, big, nbytes in guards:
    lo, hi = big[:GUARD], big[GUARD + nbytes:]
    nlo, nhi = int((lo != 0x5A).sum()), int((hi != 0x5A).sum())
    if nlo or nhi:
        il = torch.nonzero(lo != 0x5A).flatten(); ih = torch.nonzero(hi != 0x5A).flatten()
        print(f"  GUARD of workspace {key[1]} ({nbytes} B): {nlo} bytes changed BELOW (last at -{GUARD - int(il[-1]) if nlo else 0}), {nhi} bytes changed ABOVE (first at +{int(ih[0]) if nhi else 0}, last at +{int(ih[-1]) if nhi else 0})")
print(f"guards checked: {len(guards)}")

if OUT_GUARD:
    print(f"output guards ({OUT_GUARD * 4} B each side): {len(out_guard_hits)} hits {out_guard_hits[:10]}")

print("forwards by which outputs differ (e = ef, n = nf, g = gf):", dict(sorted(sigs.items())))

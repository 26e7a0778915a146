#!/usr/bin/env python3
"""Race hunt: two threads run GNCore forwards on one handle; at the first mismatch the thread clones its WORKSPACE (the forward's intermediates are
still in it), everything stops, the same forward is re-run serially on the same stream (same workspace) and the two workspace images are diffed:
the byte ranges that differ name the first corrupted intermediate.   python tools/experiments/thread_race_dump.py"""
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import graphnets_jl_amd as gn
from oracle import gn_oracle as O
from tests import util as U

extra = int(os.environ.get("PROBE_FLAGS", "4096"))  # NO_FORK by default: one stream per forward
rng = np.random.default_rng(4900)
dims = (128, 64, 32)
graphs = [U.er_csc(rng, n, e) for n, e in ((900, 12000), (300, 2500))]
g = gn.GNGraphBatch.from_csc([c for c, _ in graphs], [r for _, r in graphs], [900, 300])
p = O.make_core_params(rng, dims)
core = U.core_from_params(gn, p)
xs = [U.to_nt(gn, g, *U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims)) for _ in range(4)]
ref = [tuple(t.clone() for t in (y.ef, y.nf, y.gf)) for y in (core(x, flags=gn._lib.FLAG_NO_FORK | extra) for x in xs)]
torch.cuda.synchronize()
stop = threading.Event()
found = []


def worker(tid):
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for it in range(400):
            if stop.is_set():
                return
            k = (tid + 2 * it) % len(xs)
            y = core(xs[k], flags=extra)
            st.synchronize()
            if stop.is_set():
                return
            names = [n for n, a, b in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref[k]) if not torch.equal(a, b)]
            if names:
                stop.set()
                key = [kk for kk in g._ws if kk[0] == st.cuda_stream][0]
                found.append((tid, it, k, names, st, key, g._ws[key].clone(), [t.clone() for t in (y.ef, y.nf, y.gf)]))
                return


for rnd in range(30):
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if found:
        break
if not found:
    print("no mismatch caught")
    sys.exit(0)
torch.cuda.synchronize()
tid, it, k, names, st, key, ws_bad, ybad = found[0]
print(f"caught: thread {tid} it {it} input {k}: {names} differ; workspace {ws_bad.numel()} bytes, E {g.n_edges} N {g.n_nodes} G {g.n_graphs}")
with torch.cuda.stream(st):
    y = core(xs[k], flags=extra)
    st.synchronize()
    assert all(torch.equal(a, b) for a, b in zip((y.ef, y.nf, y.gf), ref[k])), "the serial re-run does not reproduce the reference"
    ws_good = g._ws[key].clone()
w_bad, w_good = ws_bad.view(torch.int32), ws_good.view(torch.int32)
diff = torch.nonzero(w_bad != w_good).flatten().cpu().numpy()
print(f"{diff.size} of {w_bad.numel()} workspace words differ")
if diff.size:
    # ranges of differing words, merged when closer than 64 words
    starts = [diff[0]]; ends = []
    for a, b in zip(diff[:-1], diff[1:]):
        if b - a > 64:
            ends.append(a); starts.append(b)
    ends.append(diff[-1])
    for s_, e_ in list(zip(starts, ends))[:60]:
        print(f"   bytes [{4 * s_:>10d}, {4 * e_ + 4:>10d})  {4 * (e_ - s_ + 1):>8d} B   first words bad {w_bad[s_:s_ + 2].view(torch.float32).tolist()} good {w_good[s_:s_ + 2].view(torch.float32).tolist()}")
    print(f"   ... {len(starts)} ranges in all")
for n, a, b in zip(("ef", "nf", "gf"), ybad, ref[k]):
    rows = torch.nonzero((a - b).abs().amax(dim=0).flatten() > 0).flatten()
    print(f"   {n}: {rows.numel()} rows differ {rows[:12].tolist()}")

#!/usr/bin/env python3
"""Are the library's DEFAULT forms independent of what else runs on the GPU?  (profiles/r05_mfma_mix_hazard.log.)

Round 5: k_rows_gemm / k_ffn_fused came out wrong now and then while a dense bf16 matrix kernel ran on ANOTHER stream.  Round 6: no default path
issues the fp32 matrix instruction (csrc/gnx_x6_mma.h), the site of the damage is found and guarded (csrc/gnx_wide.hip: GNX_LN_GUARD) and the
library's calls overlap by default (GNX_TAKE_TURNS=1 in the environment: round 5's turn-taking).  The probe checks, bit for bit against serial
one-stream results:

  (i)  a GNCore(128,64,32) forward + backward loop (small batch: the general kernels; big batch: >= 4096 nodes, the six-term kernels proper) on one
       stream while a torch bf16 GEMM loop (2048^3, hipBLASLt) runs on another stream of the same device;
  (ii) two captured hipGraphs of default-form forwards (two handles of the same batch) replayed concurrently on two streams.

python tests/overlap_probe.py [forms=default|fp32|nofork] [iters=40] [big=0|1]      -> one JSON line with the mismatch counts
(nofork: default forms with GNX_FLAG_NO_FORK — every kernel of a forward on the caller's stream)"""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

forms = sys.argv[1] if len(sys.argv) > 1 else "default"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
big = len(sys.argv) > 3 and sys.argv[3] == "1"
poison = len(sys.argv) > 4 and sys.argv[4] == "poison"

import torch  # noqa: E402

import graphnets_jl_amd as gn  # noqa: E402
from oracle import gn_oracle as O  # noqa: E402  (parameter shapes only)
from tests import util as U  # noqa: E402

F = gn._lib
flags = F.FLAG_FP32_MFMA if forms == "fp32" else (F.FLAG_NO_FORK if forms == "nofork" else (int(forms.split(":")[1], 0) if forms.startswith("flags:") else 0))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rng = np.random.default_rng(6100)
dims = (128, 64, 32)
sizes = ((3000, 40000), (1500, 16000)) if big else ((900, 12000), (300, 2500))
graphs = [U.er_csc(rng, n, e) for n, e in sizes]
mk_batch = lambda: gn.GNGraphBatch.from_csc([c for c, _ in graphs], [r for _, r in graphs], [n for n, _ in sizes])
g = mk_batch()
p = O.make_core_params(rng, dims)
core = U.core_from_params(gn, p)
core.flags = flags
packed = [U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims) for _ in range(3)]


def fwd_bwd(x_np):
    """forward + backward of sum(y * w): returns (ef, nf, gf, and every gradient) as one tuple of tensors"""
    x = U.to_nt(gn, g, *x_np)
    leaves = [t.detach().clone().requires_grad_(True) for t in (x.ef, x.nf, x.gf)]
    for t in core.parameters():
        t.requires_grad_(True)
        t.grad = None
    y = core(gn.NT(g, *leaves))
    loss = (y.ef * 0.5).sum() + (y.nf * 0.25).sum() + y.gf.sum()
    loss.backward()
    return tuple(t.detach().clone() for t in (y.ef, y.nf, y.gf)) + tuple(t.grad.clone() for t in leaves) + tuple(t.grad.clone() for t in core.parameters())


ref = [fwd_bwd(x) for x in packed]
torch.cuda.synchronize()
stop = threading.Event()
bad_fb, n_fb, signatures = [], [0], []


def victim():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for it in range(iters):
            k = it % len(packed)
            if poison:  # every workspace of the handle filled with 0xFF bytes (NaN floats) on this stream before the forward: a kernel that reads
                for w in g._ws.values():  # what an EARLIER kernel of the same forward wrote — and sees older bytes instead — turns its output into NaN
                    w.fill_(255)
            out = fwd_bwd(packed[k])
            st.synchronize()
            n_fb[0] += 1
            wrong = [i for i, (a, b) in enumerate(zip(out, ref[k])) if not torch.equal(a, b)]
            if wrong:
                bad_fb.append((it, wrong[:6]))
                if len(signatures) < 4:  # what the damage looks like: tensor, wrong rows, wrong columns per wrong row, size of the error
                    for i in wrong[:3]:
                        pk = lambda t: (t.permute(2, 1, 0)[0] if t.dim() == 3 else t.reshape(-1, t.shape[-1])).double()  # Julia-shaped (D, T, 1) -> [T][D]
                        a, b = pk(out[i]), pk(ref[k][i])
                        d = torch.nan_to_num((a - b).abs(), nan=1e30)
                        rows = torch.nonzero(d.amax(dim=1) > 0).flatten()
                        signatures.append({"it": it, "tensor": i, "shape": list(a.shape), "wrong_rows": int(rows.numel()), "first_rows": rows[:8].tolist(),
                                           "wrong_cols_in_first_row": int((d[rows[0]] > 0).sum()) if rows.numel() else 0, "first_cols": torch.nonzero(d[rows[0]] > 0).flatten()[:4].tolist() if rows.numel() else [],
                                           "max_rel_err": float((d / b.abs().clamp_min(1e-30)).max()), "max_abs_err": float(d.max()), "max_ref": float(b.abs().max()), "nan_elements": int(torch.isnan(a).sum())})
    stop.set()


def aggressor():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        X = torch.randn(2048, 2048, device=dev).to(torch.bfloat16)
        while not stop.is_set():
            X = (X @ X).clamp_(-1, 1)
            st.synchronize()


ts = [threading.Thread(target=victim), threading.Thread(target=aggressor)]
[t.start() for t in ts]
[t.join() for t in ts]

# (ii) two captured graphs of the forward, one handle each (a handle's workspace is per handle), replayed on two streams at once
for t in core.parameters():
    t.requires_grad_(False)
g2 = [mk_batch(), mk_batch()]
caps, refs2 = [], []
for i, gi in enumerate(g2):
    x = U.to_nt(gn, gi, *packed[i])
    yi = core(x)
    refs2.append(tuple(t.clone() for t in (yi.ef, yi.nf, yi.gf)))
    caps.append(gn.Graphed(core, x))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad_graphs = 0
for it in range(iters):
    for st, c in zip(streams, caps):
        with torch.cuda.stream(st):
            c.graph.replay()
    torch.cuda.synchronize()
    for c, r in zip(caps, refs2):
        if not all(torch.equal(a, b) for a, b in zip((c._out.ef, c._out.nf, c._out.gf), r)):
            bad_graphs += 1
print(json.dumps({"forms": forms, "big": big, "turn_taking": "on" if os.environ.get("GNX_TAKE_TURNS", "0") not in ("", "0") else "off", "forward_backward_runs": n_fb[0],
                  "forward_backward_wrong": len(bad_fb), "first_wrong": bad_fb[:3], "signatures": signatures, "graph_replay_pairs": iters, "graph_replays_wrong": bad_graphs}))

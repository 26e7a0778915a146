#!/usr/bin/env python3
"""Training through the HIP forward/backward — a small analogue of the reference's only end-to-end workload
(/root/reference/examples/sort/sort.jl:31-81,116-134): every graph is a fully connected set of N random numbers, the
model (GNBlock -> GNCoreList -> GNBlock) learns, per node, whether the node holds the minimum, and per edge i->j whether
x_j is the successor of x_i in sorted order; loss = logitcrossentropy on flatunpaddednf + flatunpaddedef.
    python examples/train_sort.py [--iters 300] [--graphs 64] [--n 8] [--width 16]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphnets_jl_amd as gn  # noqa: E402


def make_batch(rng, n_graphs, n, device, vocab=0):
    """vocab = 0: n random reals per graph as a 1-wide node feature; vocab > 0: the reference's setup — 2..n random integers in
    1..vocab per graph, one-hot node features of width vocab (sort.jl:13-26)."""
    adjs, nfs, tn, te = [], [], [], []
    n_max = n
    for _ in range(n_graphs):
        if vocab:
            n = int(rng.integers(2, n_max + 1))
            ints = rng.integers(1, vocab + 1, n)
            x = ints.astype(np.float32) + 1e-3 * rng.random(n).astype(np.float32)  # ties broken at random for the targets
        else:
            x = rng.random(n).astype(np.float32)
        order = np.argsort(x)
        rank = np.empty(n, dtype=np.int64); rank[order] = np.arange(n)
        adjs.append(np.ones((n, n), dtype=np.int64))
        nfs.append(np.eye(vocab, dtype=np.float32)[ints - 1].T.copy() if vocab else x[None, :])
        tn.append(np.stack([rank == 0, rank != 0]).astype(np.float32))                     # (2, n): is-minimum one-hot
        succ = (rank[None, :] == rank[:, None] + 1)                                        # succ[i, j]: x_j follows x_i
        flat = succ.flatten(order="F")                                                     # edge order = column-major ones
        te.append(np.stack([flat, ~flat]).astype(np.float32))                              # (2, n*n)
    x = gn.batch(dict(graphs=adjs, ef=None, nf=nfs, gf=None), device=device)
    return x, torch.from_numpy(np.concatenate(tn, axis=1)).to(device), torch.from_numpy(np.concatenate(te, axis=1)).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--graphs", type=int, default=64)
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--width", type=int, default=16)
    ap.add_argument("--lr", type=float, default=3e-3)
    ap.add_argument("--profile", action="store_true", help="per-kernel GPU time of the last iteration (gnx_profile_*)")
    ap.add_argument("--dropout", type=float, default=0.0, help="GNCore(dims; dropout): the FeedForwards' Dropout, active in the training calls (sort.jl:119 trains with 0)")
    ap.add_argument("--vocab", type=int, default=0, help="one-hot integer inputs like the reference (sort.jl uses 100)")
    ap.add_argument("--reference-config", action="store_true",
                    help="the reference's sizes: vocab 100, 2..10 nodes, core width 384, batch 4, 2 cores (sort.jl:11-16,86-89,116)")
    args = ap.parse_args()
    if args.reference_config:
        args.vocab, args.n, args.width, args.graphs = 100, 10, 384, 4
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    w = args.width
    enc = gn.GNBlock((0, args.vocab or 1, 0), (w, w, w), device=dev, act=("relu", "relu", "relu"))
    cores = gn.GNCoreList([gn.GNCore((w, w, w), dropout=args.dropout, device=dev) for _ in range(2)])
    dec = gn.GNBlock((w, w, w), (2, 2, 0), device=dev)
    params = []
    for blk in (enc, dec):
        for l in (blk.edgefn, blk.nodefn, blk.graphfn):
            params += [l.weight, l.bias]
    for c in cores.list:
        params += c.parameters()
    params = [q for q in params if q.numel() > 0]
    for q in params:
        q.requires_grad_(True)
    opt = torch.optim.AdamW(params, lr=args.lr)
    rng = np.random.default_rng(0)
    hist = []
    import time
    t_start = None
    for it in range(args.iters):
        if it == min(10, args.iters - 1):
            torch.cuda.synchronize(); t_start = (time.perf_counter(), it)
        x, tn, te = make_batch(rng, args.graphs, args.n, dev, args.vocab)
        if args.profile and it == args.iters - 1:
            torch.cuda.synchronize(); gn._lib.profile_enable(True); gn._lib.profile_reset()
        y = dec(cores(enc(x)))
        loss = gn.logitcrossentropy(gn.flatunpaddednf(y), tn) + gn.logitcrossentropy(gn.flatunpaddedef(y), te)
        opt.zero_grad()
        loss.backward()
        opt.step()
        hist.append(float(loss.detach()))
        if it % 50 == 0 or it == args.iters - 1:
            acc = float((gn.flatunpaddednf(y).argmax(0) == tn.argmax(0)).float().mean())
            print(f"iter {it:4d}  loss {hist[-1]:.4f}  node accuracy {acc:.3f}")
    torch.cuda.synchronize()
    if args.profile:
        prof = sorted(((v["total_ms"] * 1e3, k, v["launches"]) for k, v in gn._lib.profile_read().items() if not k.startswith("__")), reverse=True)
        gn._lib.profile_enable(False)
        print("last iteration, gnx kernels (us, launches):", [(round(t, 1), k, n) for t, k, n in prof], "total %.1f us" % sum(t for t, _, _ in prof))
    if t_start and args.iters - t_start[1] > 0:
        print(f"{(time.perf_counter() - t_start[0]) / (args.iters - t_start[1]) * 1e3:.2f} ms / training iteration (forward + backward + AdamW, incl. host batch construction)")
    return hist


if __name__ == "__main__":
    main()

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


def make_batch(rng, n_graphs, n, device):
    adjs, nfs, tn, te = [], [], [], []
    for _ in range(n_graphs):
        x = rng.random(n).astype(np.float32)
        order = np.argsort(x)
        rank = np.empty(n, dtype=np.int64); rank[order] = np.arange(n)
        adjs.append(np.ones((n, n), dtype=np.int64))
        nfs.append(x[None, :])
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
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    w = args.width
    enc = gn.GNBlock((0, 1, 0), (w, w, w), device=dev, act=("relu", "relu", "relu"))
    cores = gn.GNCoreList([gn.GNCore((w, w, w), device=dev) for _ in range(2)])
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
    for it in range(args.iters):
        x, tn, te = make_batch(rng, args.graphs, args.n, dev)
        y = dec(cores(enc(x)))
        loss = gn.logitcrossentropy(gn.flatunpaddednf(y), tn) + gn.logitcrossentropy(gn.flatunpaddedef(y), te)
        opt.zero_grad()
        loss.backward()
        opt.step()
        hist.append(float(loss.detach()))
        if it % 50 == 0 or it == args.iters - 1:
            acc = float((gn.flatunpaddednf(y).argmax(0) == tn.argmax(0)).float().mean())
            print(f"iter {it:4d}  loss {hist[-1]:.4f}  node accuracy {acc:.3f}")
    return hist


if __name__ == "__main__":
    main()

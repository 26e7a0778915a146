#!/usr/bin/env python3
"""GNGraphBatch construction time for a dense 4096-graph batch: `python tools/time_build.py [G]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import graphnets_jl_amd as gn  # noqa: E402
import torch  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n, e = bench.hetero_spec(5, G, 1_000_000 * G // 4096)
rng = np.random.default_rng(0)
adjs = []
for ni, ei in zip(n, e):
    a = np.zeros(int(ni) * int(ni), dtype=np.uint8)
    a[rng.choice(a.size, int(ei), replace=False)] = 1
    adjs.append(a.reshape(int(ni), int(ni)))
entries = sum(a.size for a in adjs)
for name, conv in (("bool", lambda a: a.astype(bool)), ("int64", lambda a: a.astype(np.int64)), ("float32", lambda a: a.astype(np.float32))):
    mats = [conv(a) for a in adjs]
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = gn.GNGraphBatch(mats)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{G} graphs, {entries / 1e6:.0f}M adjacency entries as {name}: GNGraphBatch in {best * 1e3:.1f} ms ({g.n_edges} edges, {g.n_nodes} nodes)")

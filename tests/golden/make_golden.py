#!/usr/bin/env python3
"""Generates tests/golden/*.npz — golden input/output vectors for the GNBlock / GNCore forward.

The reference (Julia) cannot run in this image, so these vectors are produced by the float64 DENSE-form oracle
(oracle/gn_oracle.py: a line-by-line restatement of the reference's padded one-hot formulation) — except `er1k`,
which is too large for the dense form and uses the sparse form (proved equal to the dense form in tests/test_oracle.py).
Inputs follow the reference's tests: rand(Float32) features (test/runtests.jl:141-142), glorot-uniform weights; biases
are drawn non-zero so the bias path is exercised.  Everything is stored packed ([R][T][D], the C-ABI layout) together
with the global CSC; outputs are float64.      Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gn_oracle as O  # noqa: E402

README_ADJ = np.array([[1, 0, 1], [1, 1, 0], [0, 0, 1]])
README_ADJ2 = np.array([[1, 0, 1, 0], [1, 1, 0, 1], [0, 0, 1, 0], [1, 1, 0, 1]])


def flat_params(p, prefix=""):
    out = {}
    for k, v in p.items():
        if isinstance(v, dict):
            out.update(flat_params(v, prefix + k + "."))
        else:
            out[prefix + k] = np.asarray(v)
    return out


def unflat_params(z, prefix=""):
    p = {}
    for k in z.files:
        if not k.startswith("p."):
            continue
        cur = p
        parts = k[2:].split(".")
        for q in parts[:-1]:
            cur = cur.setdefault(q, {})
        v = z[k]
        cur[parts[-1]] = tuple(int(x) for x in v) if parts[-1] in ("in_dims", "out_dims", "dims") else (v.item() if v.ndim == 0 else v)
    return p


def save(name, adjs, csc, p, ins, outs, kind):
    d = {"kind": np.array(kind), "n_graphs": np.array(len(csc[2]) - 1)}
    for k, a in zip(("colptr", "rowval", "node_off", "edge_off"), csc):
        d[k] = a
    if adjs is not None:
        for i, a in enumerate(adjs):
            d[f"adj{i}"] = np.asarray(a, dtype=np.int8)
    for k, v in flat_params(p, "p.").items():
        d[k] = v
    for k, v in zip(("ef", "nf", "gf"), ins):
        if v is not None:
            d["in_" + k] = v
    for k, v in zip(("ef", "nf", "gf"), outs):
        if v is not None:
            d["out_" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items() if k.startswith(("in_", "out_"))})


def dense_block_packed(p, adjs, shared, ef, nf, gf):
    """Runs the DENSE form on Julia-shaped inputs and returns packed outputs."""
    if shared:
        y = O.unbatch_dense(O.block_forward_dense(p, O.batch_dense(adjs[0], ef, nf, gf)))
        pk = O.packed_from_julia_shared
        return pk(y["ef"]), pk(y["nf"]), None if y["gf"] is None else pk(y["gf"][:, None, :])
    y = O.unbatch_dense(O.block_forward_dense(p, O.batch_dense(adjs, ef, nf, gf)))
    pk = O.packed_from_julia_vector
    return pk(y["ef"]), pk(y["nf"]), pk(y["gf"])


def main():
    # 1. README example 1 = BASELINE configs[0]: shared 3-node/5-edge graph, batch_size 2, (10,5,0)=>(3,4,5)
    rng = np.random.default_rng(1)
    p = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    ef, nf = rng.random((10, 5, 2), dtype=np.float32), rng.random((5, 3, 2), dtype=np.float32)
    outs = dense_block_packed(p, [README_ADJ], True, ef, nf, None)
    pk = O.packed_from_julia_shared
    save("readme_ex1", [README_ADJ], O.csc_from_adj([README_ADJ]), p, (pk(ef), pk(nf), None), outs, "block")

    # 2. README example 2: vector of two graphs
    rng = np.random.default_rng(2)
    adjs = [README_ADJ, README_ADJ2]
    p = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    ef = [rng.random((10, int(a.sum())), dtype=np.float32) for a in adjs]
    nf = [rng.random((5, a.shape[0]), dtype=np.float32) for a in adjs]
    outs = dense_block_packed(p, adjs, False, ef, nf, None)
    pv = O.packed_from_julia_vector
    save("readme_ex2", adjs, O.csc_from_adj(adjs), p, (pv(ef), pv(nf), None), outs, "block")

    # 3. batch-invariance pair (test/runtests.jl:62-116): encoder (0,2,0)=>(2,2,2) on [A, B]
    rng = np.random.default_rng(3)
    adjs = [np.ones((2, 2), dtype=int), np.ones((3, 3), dtype=int)]
    p = O.make_block_params(rng, (0, 2, 0), (2, 2, 2))
    nf = [rng.random((2, 2), dtype=np.float32), rng.random((2, 3), dtype=np.float32)]
    outs = dense_block_packed(p, adjs, False, None, nf, None)
    save("batch_invariance_AB", adjs, O.csc_from_adj(adjs), p, (None, pv(nf), None), outs, "block")

    # 4. all three inputs present, non-identity activations, heterogeneous batch with an edgeless graph
    rng = np.random.default_rng(4)
    adjs = [(rng.random((n, n)) < 0.4).astype(int) for n in (5, 1, 9, 3)]
    adjs[3][:] = 0
    p = O.make_block_params(rng, (3, 2, 4), (3, 4, 5), act=(O.ACT_RELU, O.ACT_TANH, O.ACT_SIGMOID))
    ef = [rng.random((3, int(a.sum())), dtype=np.float32) for a in adjs]
    nf = [rng.random((2, a.shape[0]), dtype=np.float32) for a in adjs]
    gf = [rng.random((4,), dtype=np.float32) for a in adjs]
    outs = dense_block_packed(p, adjs, False, ef, nf, gf)
    save("hetero_all_inputs", adjs, O.csc_from_adj(adjs), p, (pv(ef), pv(nf), pv(gf)), outs, "block")

    # 5. GNCore(3,4,5) on the README graph, batch_size 2 (test/runtests.jl:685-709), Flux-0.14 LayerNorm convention
    rng = np.random.default_rng(5)
    pc = O.make_core_params(rng, (3, 4, 5), eps_mode=0)
    ef, nf, gf = rng.random((3, 5, 2), dtype=np.float32), rng.random((4, 3, 2), dtype=np.float32), rng.random((5, 2), dtype=np.float32)
    y = O.unbatch_dense(O.core_forward_dense(pc, O.batch_dense(README_ADJ, ef, nf, gf)))
    outs = (pk(y["ef"]), pk(y["nf"]), pk(y["gf"][:, None, :]))
    save("core_readme", [README_ADJ], O.csc_from_adj([README_ADJ]), pc, (pk(ef), pk(nf), pk(gf[:, None, :])), outs, "core")

    # 6. 1000-node / 8000-edge Erdos-Renyi graph (C2's recipe, scaled), README dims — sparse form
    rng = np.random.default_rng(6)
    N, E = 1000, 8000
    k = np.sort(rng.permutation(np.unique(rng.integers(0, N * N, int(E * 1.2))))[:E])
    colptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(colptr, k // N + 1, 1)
    csc = (np.cumsum(colptr), (k % N).astype(np.int64), np.array([0, N]), np.array([0, E]))
    p = O.make_block_params(rng, (10, 5, 0), (3, 4, 5))
    ef, nf = rng.random((1, E, 10), dtype=np.float32), rng.random((1, N, 5), dtype=np.float32)
    outs = O.block_forward_sparse(p, csc, ef, nf, None)
    save("er1k", None, csc, p, (ef, nf, None), outs, "block")


if __name__ == "__main__":
    main()

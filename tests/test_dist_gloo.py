"""The N > 1 path on CPU: 2 ranks over gloo.  Graph partitioning, per-rank forward on the shard, all-gather of gf' and
the permutation back to original graph order.  There is no GPU here, so the per-rank forward is the ORACLE standing in
for the HIP path (tests may use it as the checker); the multi-rank result must equal the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _free_port():
    """A TCP port nobody is bound to right now (a fixed rendezvous port can be taken on a shared host)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

def _batch(seed=7, G=11):
    rng = np.random.default_rng(seed)
    adjs = [(rng.random((n, n)) < 0.4).astype(np.int64) for n in rng.integers(2, 9, G)]
    nf = [rng.random((a.shape[0], 4), dtype=np.float32) for a in adjs]
    ef = [rng.random((int(a.sum()), 3), dtype=np.float32) for a in adjs]
    return adjs, ef, nf


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import graphnets_jl_amd  # noqa: F401  (registers the package)
    from graphnets_jl_amd.dist import GfGather, partition_graphs, sharded_block_forward
    from oracle import gn_oracle as O
    adjs, ef, nf = _batch()
    p = O.make_block_params(np.random.default_rng(1), (3, 4, 0), (2, 3, 5))
    shards = partition_graphs([int(a.sum()) for a in adjs], world)
    mine = shards[rank]
    csc = O.csc_from_adj([adjs[i] for i in mine])

    def forward(x):  # oracle as the per-rank forward (CPU stand-in for the HIP path)
        e, n, g = O.block_forward_sparse(p, csc, x["ef"], x["nf"], None)
        return dict(ef=e, nf=n, gf=torch.from_numpy(g[0].astype(np.float32)))

    x = dict(ef=np.concatenate([ef[i] for i in mine])[None], nf=np.concatenate([nf[i] for i in mine])[None])
    gather = GfGather(shards, rank, world, dg=5, device="cpu")
    _, gf_all = sharded_block_forward(forward, x, gather)
    if rank == 0:
        np.save(out, gf_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    sys.path.insert(0, ROOT)
    import graphnets_jl_amd  # noqa: F401
    from graphnets_jl_amd.dist import partition_graphs
    rng = np.random.default_rng(0)
    counts = rng.integers(10, 5000, 4096)
    shards = partition_graphs(counts, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(4096))
    assert all(len(s) == 512 for s in shards)
    loads = np.array([counts[s].sum() for s in shards], dtype=np.float64)
    assert loads.max() / loads.mean() < 1.01


@pytest.mark.timeout(120)
def test_two_rank_gloo_matches_single_process(tmp_path):
    from oracle import gn_oracle as O
    out = str(tmp_path / "gf_all.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    adjs, ef, nf = _batch()
    p = O.make_block_params(np.random.default_rng(1), (3, 4, 0), (2, 3, 5))
    _, _, gf = O.block_forward_sparse(p, O.csc_from_adj(adjs), np.concatenate(ef)[None], np.concatenate(nf)[None], None)
    np.testing.assert_allclose(np.load(out), gf[0], rtol=1e-6, atol=1e-6)


def _stack_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import graphnets_jl_amd  # noqa: F401
    from graphnets_jl_amd.dist import GfGather, partition_graphs
    counts = np.random.default_rng(3).integers(1, 50, 7)  # 7 graphs over 2 ranks: unequal shards (4 + 3)
    shards = partition_graphs(counts, world)
    ga = GfGather(shards, rank, world, dg=2, device="cpu", stack=3)
    # row (m, g) of the full table holds [100*m + g, rank that owns g]
    for m in range(3):
        for j, gid in enumerate(shards[rank]):
            ga.send[m, j, 0], ga.send[m, j, 1] = 100 * m + int(gid), rank
    ga.start_inplace()
    res = ga.finish()
    if rank == 0:
        np.save(out, res.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_stacked_gather_restores_original_graph_order(tmp_path):
    """M stacked gf' tables in one collective, unequal shards: every (step, graph) row lands at its original graph id."""
    from graphnets_jl_amd.dist import partition_graphs
    out = str(tmp_path / "stack.npy")
    mp.spawn(_stack_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = np.load(out)
    assert res.shape == (3, 7, 2)
    shards = partition_graphs(np.random.default_rng(3).integers(1, 50, 7), 2)
    owner = np.zeros(7)
    owner[shards[1]] = 1
    for m in range(3):
        np.testing.assert_array_equal(res[m, :, 0], 100 * m + np.arange(7))
        np.testing.assert_array_equal(res[m, :, 1], owner)


def test_bench_hetero_generators_and_self_launch(monkeypatch):
    """bench.py: the heterogeneous batch has exactly the requested totals, a rank can generate just its shard, and
    `--gpus N` outside torchrun starts N ranks as a child job (never an exec of this process)."""
    sys.path.insert(0, ROOT)
    import bench
    n, e = bench.hetero_spec(5, 4096, 1_000_000)
    assert len(n) == 4096 and int(e.sum()) == 1_000_000 and n.min() >= 32 and n.max() <= 256 and (e <= n * n).all()
    cps, rvs, nn = bench.make_hetero(5, 4096, 1_000_000, only=[7, 4000])
    cp2, rv2 = bench.hetero_graph(5, 4000, n[4000], e[4000])
    assert nn == [int(n[7]), int(n[4000])] and np.array_equal(cps[1], cp2) and np.array_equal(rvs[1], rv2) and cps[1][-1] == e[4000]
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert "127.0.0.1" in cmd


def _strong_worker(rank, world, port, out):
    """What a rank of `bench.py --gpus N` does before it touches a GPU: pick the sharded workload, partition it, generate ITS graphs."""
    import argparse
    import zlib
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import graphnets_jl_amd  # noqa: F401
    from graphnets_jl_amd.dist import partition_graphs
    Gtot, Etot, seed = bench.sharded_workload(argparse.Namespace(scaling="strong", hetero_graphs=None, hetero_edges=None), world)
    _, e_all = bench.hetero_spec(seed, Gtot, Etot)
    shards = partition_graphs(e_all, world)
    cps, rvs, nn = bench.make_hetero(seed, Gtot, Etot, only=shards[rank])
    # per graph of this rank: (original id, nodes, edges, crc of its CSC arrays)
    rows = np.array([[int(g), n, len(rv), zlib.crc32(cp.tobytes() + rv.tobytes())] for g, cp, rv, n in zip(shards[rank], cps, rvs, nn)], dtype=np.int64)
    pad = np.zeros((Gtot, 4), dtype=np.int64)
    pad[:len(rows)] = rows
    parts = [torch.zeros((Gtot, 4), dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(pad))
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(rows)]))
    if rank == 0:
        np.save(out, np.concatenate([p.numpy()[:int(c)] for p, c in zip(parts, counts)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bench_strong_scaling_shards_one_fixed_batch(tmp_path):
    """`bench.py --gpus N` (default --scaling strong) measures BASELINE configs[4]: the SAME 4096 graphs / 1M edges (seed 5) at every N.  Two gloo ranks
    pick, partition and generate the workload exactly as bench.py's ranks do; together they hold every graph of the one-rank batch once, bit for bit."""
    import argparse
    import zlib
    sys.path.insert(0, ROOT)
    import bench
    out = str(tmp_path / "strong.npy")
    mp.spawn(_strong_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    rows = np.load(out)
    assert rows.shape == (4096, 4) and sorted(rows[:, 0].tolist()) == list(range(4096)) and int(rows[:, 2].sum()) == 1_000_000
    # world = 1 picks the same batch: configs[4] (4096 graphs, 1M edges, seed 5), and so does world = 8
    for world in (1, 8):
        assert bench.sharded_workload(argparse.Namespace(scaling="strong", hetero_graphs=None, hetero_edges=None), world) == (4096, 1_000_000, 5)
    cps, rvs, nn = bench.make_hetero(5, 4096, 1_000_000)
    whole = {g: (n, len(rv), zlib.crc32(cp.tobytes() + rv.tobytes())) for g, (cp, rv, n) in enumerate(zip(cps, rvs, nn))}
    for g, n, e, crc in rows.tolist():
        assert whole[g] == (n, e, crc)
    # the two shards are balanced: 2048 graphs each, edge counts within 1 %
    half = rows[:2048, 2].sum(), rows[2048:, 2].sum()
    assert abs(int(half[0]) - int(half[1])) < 10_000
    # weak scaling (opt-in) keeps the per-GPU shard fixed instead
    assert bench.sharded_workload(argparse.Namespace(scaling="weak", hetero_graphs=None, hetero_edges=None), 8) == (4096, 8_000_000, 5)
    assert bench.sharded_workload(argparse.Namespace(scaling="weak", hetero_graphs=None, hetero_edges=None), 2)[:2] == (1024, 2_000_000)


def test_c_partition_equals_the_specified_rule():
    """gnx_dist_partition against an independent restatement of its rule: sort by edge count descending (stable), deal in snake
    order, every rank keeps ascending original ids; also odd sizes and more ranks than graphs."""
    sys.path.insert(0, ROOT)
    import graphnets_jl_amd  # noqa: F401
    from graphnets_jl_amd.dist import partition_graphs

    def rule(counts, world):
        order = np.argsort(-np.asarray(counts, dtype=np.int64), kind="stable")
        shards = [[] for _ in range(world)]
        for i, gidx in enumerate(order):
            rnd, pos = divmod(i, world)
            shards[pos if rnd % 2 == 0 else world - 1 - pos].append(int(gidx))
        return [np.asarray(sorted(s), dtype=np.int64) for s in shards]

    rng = np.random.default_rng(11)
    for G, world in ((4096, 8), (7, 2), (3, 8), (1, 1), (1000, 3)):
        counts = rng.integers(0, 5000, G)
        got, want = partition_graphs(counts, world), rule(counts, world)
        assert len(got) == world
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)


def test_c_gather_plan_equals_the_rule_for_eight_unequal_shards():
    """`gnx_dist_gather_plan` — the index table of gnx_dist_create and of GfGather — against its rule written out in numpy, for 8 UNEQUAL
    shards (one of them empty) of a shuffled permutation; broken partitions are rejected.  The first real 8-GPU run must not be the first
    execution of this arithmetic."""
    from graphnets_jl_amd.dist import gather_plan, GfGather
    from graphnets_jl_amd._lib import GnxError
    rng = np.random.default_rng(11)
    sizes = [1, 700, 0, 2000, 33, 512, 849, 1]
    G = sum(sizes)
    perm = rng.permutation(G)
    shards, o = [], 0
    for n in sizes:
        shards.append(perm[o:o + n].astype(np.int64)); o += n
    src, mc = gather_plan(shards)
    assert mc == 2000 and src.dtype == np.int32
    ref = np.empty(G, dtype=np.int64)
    for r, s in enumerate(shards):
        ref[s] = r * mc + np.arange(len(s))
    assert np.array_equal(src, ref)
    # GfGather's stacked table is that plan applied per step: rank r contributes M consecutive [max_count] tables
    M = 3
    ga = GfGather(shards, 0, 8, 5, "cpu", overlap=False, stack=M)
    want = np.concatenate([(ref // mc * M + m) * mc + ref % mc for m in range(M)])
    assert np.array_equal(ga.src_index.numpy(), want)
    wire = torch.arange(8 * M * mc * 5, dtype=torch.float32).reshape(8 * M * mc, 5)
    ga.recv.copy_(wire)
    out = ga.result().numpy()
    for r, s in enumerate(shards):
        for m in range(M):
            for k in (0, len(s) - 1):
                if len(s):
                    assert np.array_equal(out[m, s[k]], wire[(r * M + m) * mc + k].numpy())
    bad = [s.copy() for s in shards]
    bad[1][0] = bad[3][0]  # a graph owned twice
    with pytest.raises(GnxError):
        gather_plan(bad)


def _replica_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import graphnets_jl_amd  # noqa: F401
    from graphnets_jl_amd.dist import GfGather, partition_replicas, sharded_replica_forward
    from oracle import gn_oracle as O
    adj, ef, nf, p = _shared_batch()
    shards = partition_replicas(ef.shape[0], world)
    mine = shards[rank]
    csc = O.csc_from_adj([adj])

    def forward(x):  # oracle as the per-rank forward (CPU stand-in for the HIP path); gf' Julia-shaped (DG, 1, R_local)
        _, _, g = O.block_forward_sparse(p, csc, x["ef"], x["nf"], None)
        return dict(gf=torch.from_numpy(g.astype(np.float32)).permute(2, 1, 0))

    gather = GfGather(shards, rank, world, dg=5, device="cpu")
    _, gf_all = sharded_replica_forward(forward, dict(ef=ef[mine], nf=nf[mine]), gather)
    if rank == 0:
        np.save(out, gf_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _shared_batch(R=5):
    from oracle import gn_oracle as O
    rng = np.random.default_rng(17)
    adj = (rng.random((9, 9)) < 0.4).astype(np.int64)
    ef = rng.random((R, int(adj.sum()), 3), dtype=np.float32)   # packed [R][E][DE]
    nf = rng.random((R, 9, 4), dtype=np.float32)
    return adj, ef, nf, O.make_block_params(np.random.default_rng(2), (3, 4, 0), (2, 3, 5))


@pytest.mark.timeout(120)
def test_shared_graph_batch_shards_over_its_data_batch(tmp_path):
    """"shared-graph batches shard over the data batch B the same way" — 5 replicas of one graph over 2 gloo ranks (3 + 2),
    gf' gathered in replica order, equal to the one-process result."""
    sys.path.insert(0, ROOT)
    import graphnets_jl_amd  # noqa: F401
    from graphnets_jl_amd.dist import partition_replicas
    from oracle import gn_oracle as O
    parts = partition_replicas(5, 2)
    assert [p.tolist() for p in parts] == [[0, 1], [2, 3, 4]]
    assert [len(p) for p in partition_replicas(4096, 8)] == [512] * 8 and sum(len(p) for p in partition_replicas(3, 8)) == 3
    out = str(tmp_path / "gf_rep.npy")
    mp.spawn(_replica_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    adj, ef, nf, p = _shared_batch()
    _, _, gf = O.block_forward_sparse(p, O.csc_from_adj([adj]), ef, nf, None)
    np.testing.assert_allclose(np.load(out), gf[:, 0, :], rtol=1e-6, atol=1e-6)

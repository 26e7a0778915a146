"""BASELINE configs[4] on the GPU: the 4096-graph heterogeneous batch on one MI355X against the oracle, and the by-graph
sharded path — partition, per-rank forward, all-gather of gf', permutation back to original graph order — through real RCCL
(world 1) and through 8 virtual ranks on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bench
from oracle import gn_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gn():
    import graphnets_jl_amd as gn
    return gn


@pytest.mark.parametrize("dims", [((10, 5, 0), (3, 4, 5)), ((10, 5, 3), (10, 5, 3))], ids=["readme", "with-gf"])
def test_config5_4096_graphs_1m_edges_one_gpu(gn, dims):
    """BASELINE configs[4]'s batch (C5 law: 4096 graphs, 32..256 nodes, exactly 1M edges, seed 5) on ONE GPU vs the oracle."""
    colptrs, rowvals, nn = bench.make_hetero(5, 4096, 1_000_000)
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn)
    assert g.n_graphs == 4096 and g.n_edges == 1_000_000
    rng = np.random.default_rng(500)
    p = O.make_block_params(rng, *dims)
    ef, nf, gf = U.packed_inputs(rng, 1, g.n_edges, g.n_nodes, g.n_graphs, dims[0])
    y = U.block_from_params(gn, p)(U.to_nt(gn, g, ef, nf, gf))
    ref, scale = O.block_forward_sparse(p, (*g.csc(), g.node_off, g.edge_off), ef, nf, gf, return_scale=True)
    for name, got, r, s in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref, scale):
        U.assert_close(U.from_jl(got), r, s, name, log=f"configs[4] C5 4096 graphs 1M edges {dims[0]}=>{dims[1]} (one GPU)")


def _sharded_forward(gn, world, rank, shards, colptrs, rowvals, nn, ef, nf, node_off, edge_off, blk, gather):
    """What one rank does: handle of ITS graphs, forward on ITS feature rows, gf' rows into the gather's send buffer."""
    import torch
    mine = shards[rank]
    g = gn.GNGraphBatch.from_csc([colptrs[i] for i in mine], [rowvals[i] for i in mine], [nn[i] for i in mine])
    efl = np.concatenate([ef[0, edge_off[i]:edge_off[i + 1]] for i in mine])[None]
    nfl = np.concatenate([nf[0, node_off[i]:node_off[i + 1]] for i in mine])[None]
    plan = gn.BlockPlan(blk, g)
    eo, no, _ = plan.outputs()
    plan(torch.from_numpy(efl).to(g.device), torch.from_numpy(nfl).to(g.device), None, eo, no, gather.send[:, :len(mine)])
    return eo, no


def test_by_graph_sharding_eight_virtual_ranks_equals_the_oracle(gn):
    """The whole N > 1 data path except the wire: 8 shards of the 4096-graph batch run one after the other on this GPU, their
    send buffers are concatenated exactly as all_gather_into_tensor would, the C boundary's gather plan (gnx_dist_gather_plan: GfGather's
    index table AND gnx_dist_permute_rows, the kernel behind gnx_dist_allgather_gf) restores original graph order — gf' of the WHOLE batch (and every shard's ef', nf') equals the oracle's single-process result."""
    import torch
    world, G, E = 8, 4096, 1_000_000
    from graphnets_jl_amd.dist import GfGather, partition_graphs
    colptrs, rowvals, nn = bench.make_hetero(5, G, E)
    n_all, e_all = bench.hetero_spec(5, G, E)
    shards = partition_graphs(e_all, world)
    assert all(len(s) == G // world for s in shards)
    loads = np.array([e_all[s].sum() for s in shards], dtype=np.float64)
    assert loads.max() / loads.mean() < 1.01
    node_off, edge_off = np.concatenate([[0], np.cumsum(nn)]), np.concatenate([[0], np.cumsum(e_all)])
    dims = ((10, 5, 0), (3, 4, 5))
    rng = np.random.default_rng(501)
    p = O.make_block_params(rng, *dims)
    ef, nf, _ = U.packed_inputs(rng, 1, E, int(node_off[-1]), G, dims[0])
    blk = U.block_from_params(gn, p)
    gathers = [GfGather(shards, r, world, 5, "cuda", overlap=False) for r in range(world)]
    outs = [_sharded_forward(gn, world, r, shards, colptrs, rowvals, nn, ef, nf, node_off, edge_off, blk, gathers[r]) for r in range(world)]
    torch.cuda.synchronize()
    wire = torch.cat([gt.send.view(-1, 5) for gt in gathers], dim=0).contiguous()  # = all_gather_into_tensor of the 8 send buffers
    gathers[0].recv.copy_(wire)
    gf_all = gathers[0].result().cpu().numpy()
    # the same wire through the C boundary's own plan and kernel (what gnx_dist_allgather_gf runs behind ncclAllGather): bit-equal
    import ctypes as C
    from graphnets_jl_amd.dist import gather_plan
    src, mc = gather_plan(shards)
    assert mc == G // world
    src_dev = torch.from_numpy(src).cuda()
    out_c = torch.empty((G, 5), dtype=torch.float32, device="cuda")
    gn._lib.check(gn._lib.load().gnx_dist_permute_rows(wire.data_ptr(), src_dev.data_ptr(), G, 5, out_c.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(out_c.cpu().numpy(), gf_all)
    cp = np.concatenate([[0]] + [c[1:] + edge_off[i] for i, c in enumerate(colptrs)])
    rv = np.concatenate([r + node_off[i] for i, r in enumerate(rowvals)])
    ref, scale = O.block_forward_sparse(p, (cp, rv, node_off, edge_off), ef, nf, None, return_scale=True)
    U.assert_close(gf_all[None], ref[2], scale[2], "gathered gf' in original graph order")
    for r in range(world):
        eidx = np.concatenate([np.arange(edge_off[i], edge_off[i + 1]) for i in shards[r]])
        nidx = np.concatenate([np.arange(node_off[i], node_off[i + 1]) for i in shards[r]])
        U.assert_close(outs[r][0].cpu().numpy(), ref[0][:, eidx], scale[0][:, eidx], f"ef' of shard {r}")
        U.assert_close(outs[r][1].cpu().numpy(), ref[1][:, nidx], scale[1][:, nidx], f"nf' of shard {r}")


def test_gfgather_through_rccl_world_1(gn):
    """The collective itself: torch.distributed "nccl" (= RCCL) with one rank, stacked tables, side stream."""
    import torch
    import torch.distributed as dist
    from graphnets_jl_amd.dist import GfGather
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a port nobody holds right now (the GPU host is shared)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        shards = [np.array([3, 0, 2, 1])[np.argsort([3, 0, 2, 1])]]  # one rank owns every graph
        ga = GfGather(shards, 0, 1, 5, "cuda", stack=3)
        x = torch.rand((3, 4, 5), device="cuda")
        ga.start(x)
        out = ga.finish()
        torch.cuda.synchronize()
        assert torch.equal(out, x)
    finally:
        dist.destroy_process_group()


def _check_strong_line(line, backend):
    """the N > 1 line of bench.py: configs[4]'s FIXED batch, strong scaling, the same batch on one GPU beside it, with / without the all-gather"""
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["value"] > 1e9
    wl = line["config"]["workload"]
    assert "configs[4]" in wl and "4096 random graphs" in wl and "1M edges" in wl and "seed 5" in wl
    assert line["config"]["edges_whole_job"] == 1_000_000 and line["config"]["graphs_whole_job"] == 4096
    assert line["config"]["per_rank_edges_nodes_graphs"][0][0] == 1_000_000 and line["config"]["per_rank_edges_nodes_graphs"][0][2] == 4096
    single = line["single_gpu_same_workload"]
    assert single["graphs"] == 4096 and single["edges"] == 1_000_000 and single["value"] > 1e9
    w, wo = line["with_allgather"], line["without_allgather"]
    assert w["ms_per_step"] == line["ms_per_step"] and 0 < wo["ms_per_step"] <= w["ms_per_step"] * 1.15  # (the gather never makes the region faster, up to noise)
    assert line["roofline"]["frac"] <= 1.0
    c5w = line["secondary"]["c5w"]
    assert c5w["edges_whole_job"] == 8_000_000 and c5w["graphs_whole_job"] == 4096 and c5w["single_gpu_same_workload"]["edges"] == 8_000_000
    assert c5w["value"] > line["value"]  # eight times the edges per launch: the larger batch runs closer to the roof


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("backend", ["torch", "gnx"])
def test_bench_sharded_prints_the_compact_line_last(backend):
    """What the driver's 8-GPU SCALE run will parse: `bench.py --gpus 1 --force-dist` (the N > 1 branch: configs[4]'s fixed 4096-graph batch sharded
    by the product's partitioner, hipGraphs of stacked steps, RCCL all-gather on a side stream, one rank) and `--dist-backend gnx` (one process
    driving the devices through gnx_dist_block_forward_steps; its without_allgather leg is GNX_FLAG_DIST_NO_GATHER), started as the driver starts
    them (no --full-line).  The LAST stdout line is the compact headline — the contract's keys, roofline, the with / without-gather pair, the same
    batch on one GPU and C5w as numbers, below 3 KB; the secondary line is prefixed; the whole line is in gpurun_out/bench_detail_*.json and names
    configs[4]'s workload, the same batch on one GPU, the region with and without the collective, and C5w beside it."""
    from tests.test_bench_line import check_compact
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "16", "--warmup", "2"]
    cmd += ["--force-dist", "--no-cpu-baseline"] if backend == "torch" else ["--dist-backend", "gnx"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1100)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    c = json.loads(lines[-1])
    check_compact(c, 1)
    assert c["steps"] == 16 and c["warmup"] == 2 and c["scaling"] == "strong"
    assert c["config"]["edges_whole_job"] == 1_000_000 and c["config"]["graphs_whole_job"] == 4096 and "configs[4]" in c["config"]["workload"]
    assert c["with_allgather_ms"] == c["ms_per_step"] and c["without_allgather_ms"] > 0 and c["single_gpu_same_workload_ms"] > 0
    assert c["c5w"]["value"] > c["value"]
    assert any(l.startswith("bench-secondary (not the result line): ") for l in lines[:-1])
    with open(os.path.join(ROOT, c["detail"])) as f:
        line = json.load(f)
    _check_strong_line(line, backend)
    if backend == "torch":
        assert line["config"]["graphs_per_gpu"] == 4096 and line["roofline"]["kernel_us"] > 0 and c["roofline"]["kernel_us"] == line["roofline"]["kernel_us"]


@pytest.mark.timeout(900)
def test_bench_weak_scaling_stays_available():
    """`--scaling weak`: N x 512 graphs (the per-GPU shard fixed); `--full-line` prints the whole line as rounds 1-5 did"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--scaling", "weak", "--steps", "16", "--warmup", "2", "--no-single-gpu-leg", "--full-line"],
                         capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "weak" and line["config"]["graphs_per_gpu"] == 512 and line["single_gpu_same_workload"] is None

def test_gnx_dist_c_entry_points_world_1(gn):
    """The sharded path through the C boundary (what the Julia shim binds): gnx_dist_partition, gnx_dist_create (ncclCommInitAll
    over one device), gnx_dist_block_forward = per-rank gnx_block_forward + ncclAllGather of gf' + restoration of the ORIGINAL
    graph order.  With one device the single rank holds every graph — in a shuffled order, so the index table is exercised."""
    import ctypes as C
    import torch
    lib = gn._lib.load()
    rng = np.random.default_rng(77)
    G = 24
    adjs = [(rng.random((n, n)) < 0.3).astype(np.int64) for n in rng.integers(3, 30, G)]
    e_counts = np.array([int(a.sum()) for a in adjs], dtype=np.int64)
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    off, ids = np.zeros(2, dtype=np.int64), np.zeros(G, dtype=np.int64)
    gn._lib.check(lib.gnx_dist_partition(p64(e_counts), G, 1, p64(off), p64(ids)))
    assert off.tolist() == [0, G] and ids.tolist() == list(range(G))
    order = rng.permutation(G).astype(np.int64)  # the rank's local order of its graphs
    dims = ((4, 3, 2), (3, 4, 5))
    p = O.make_block_params(rng, *dims)
    ef = [rng.random((int(a.sum()), 4), dtype=np.float32) for a in adjs]
    nf = [rng.random((a.shape[0], 3), dtype=np.float32) for a in adjs]
    gf = rng.random((G, 2), dtype=np.float32)
    # local handle and local rows in the SHUFFLED order
    g = gn.GNGraphBatch([adjs[i] for i in order])
    dev = g.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    efl, nfl, gfl = t(np.concatenate([ef[i] for i in order])), t(np.concatenate([nf[i] for i in order])), t(gf[order])
    blk = U.block_from_params(gn, p)
    keep = []
    bp = blk._c(keep)
    eo = torch.empty((g.n_edges, 3), device=dev); no = torch.empty((g.n_nodes, 4), device=dev)
    gl = torch.empty((G, 5), device=dev); gall = torch.full((G, 5), float("nan"), device=dev)
    ws = torch.empty(int(lib.gnx_block_workspace_bytes(g._h, C.byref(bp), 1)), dtype=torch.uint8, device=dev)
    d = C.c_void_p()
    devs = (C.c_int32 * 1)(dev.index or 0)
    gn._lib.check(lib.gnx_dist_create(devs, 1, p64(off), p64(order), G, 5, C.byref(d)))
    try:
        arr = lambda v: (C.c_void_p * 1)(v)
        nbytes = (C.c_size_t * 1)(ws.numel())
        stream = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(3):  # repeated calls reuse the communicator and its buffers
            gall.fill_(float("nan"))
            gn._lib.check(lib.gnx_dist_block_forward(d, arr(g._h.value), arr(C.addressof(bp)), arr(efl.data_ptr()), arr(nfl.data_ptr()), arr(gfl.data_ptr()),
                                                     arr(eo.data_ptr()), arr(no.data_ptr()), arr(gl.data_ptr()), arr(gall.data_ptr()), arr(ws.data_ptr()),
                                                     nbytes, 0, arr(stream)))
            torch.cuda.synchronize()
            # oracle on the batch in ORIGINAL order
            ref, scale = O.block_forward_sparse(p, O.csc_from_adj(adjs), np.concatenate(ef)[None], np.concatenate(nf)[None], gf[None], return_scale=True)
            U.assert_close(gall.cpu().numpy()[None], ref[2], scale[2], "gf' gathered into original graph order")
            U.assert_close(gl.cpu().numpy()[None], ref[2][:, order], scale[2][:, order], "local gf' rows (shard order)")
        # argument checks: a handle that does not match the shard, a partition that is not a permutation
        g2 = gn.GNGraphBatch(adjs[:5])
        rc = lib.gnx_dist_block_forward(d, arr(g2._h.value), arr(C.addressof(bp)), arr(efl.data_ptr()), arr(nfl.data_ptr()), arr(gfl.data_ptr()),
                                        arr(eo.data_ptr()), arr(no.data_ptr()), arr(gl.data_ptr()), arr(gall.data_ptr()), arr(ws.data_ptr()), nbytes, 0, arr(stream))
        assert rc == gn._lib.ERR_COUNT_MISMATCH
    finally:
        gn._lib.check(lib.gnx_dist_destroy(d))
    bad = order.copy(); bad[0] = bad[1]
    d2 = C.c_void_p()
    assert lib.gnx_dist_create(devs, 1, p64(off), p64(bad), G, 5, C.byref(d2)) == gn._lib.ERR_INVALID_ARG


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs in one process (the driver's GPU test box has one)")
def test_gnx_dist_c_entry_points_two_devices(gn):
    """The sharded path through the C boundary with TWO ranks in one process: gnx_dist_partition over 2 ranks with unequal shards,
    one handle / parameter set / feature set per device, ncclCommInitAll over both, the grouped all-gather of gf' and the restoration of the
    original graph order on BOTH devices.  Also the batch constructor's staging path on a device that is not the process's first one."""
    import ctypes as C
    import torch
    lib = gn._lib.load()
    rng = np.random.default_rng(78)
    G, n = 30, 2
    adjs = [(rng.random((k, k)) < 0.3).astype(np.int64) for k in rng.integers(3, 40, G)]
    e_counts = np.array([int(a.sum()) for a in adjs], dtype=np.int64)
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    off, ids = np.zeros(n + 1, dtype=np.int64), np.zeros(G, dtype=np.int64)
    gn._lib.check(lib.gnx_dist_partition(p64(e_counts), G, n, p64(off), p64(ids)))
    assert off[0] == 0 and off[n] == G and sorted(ids.tolist()) == list(range(G))
    dims = ((4, 3, 2), (3, 4, 5))
    p = O.make_block_params(rng, *dims)
    ef = [rng.random((int(a.sum()), 4), dtype=np.float32) for a in adjs]
    nf = [rng.random((a.shape[0], 3), dtype=np.float32) for a in adjs]
    gf = rng.random((G, 2), dtype=np.float32)
    keep, hs, bps, bufs = [], [], [], []
    for r in range(n):
        dev = torch.device("cuda", r)
        mine = ids[off[r]:off[r + 1]]
        g = gn.GNGraphBatch([adjs[i] for i in mine], device=dev)   # dense adjacency: the device-side constructor on device r
        assert g.device == dev
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        blk = U.block_from_params(gn, p, device=dev)
        bp = blk._c(keep)
        with torch.cuda.device(dev):
            ws = torch.empty(int(lib.gnx_block_workspace_bytes(g._h, C.byref(bp), 1)), dtype=torch.uint8, device=dev)
        bufs.append(dict(g=g, ef=t(np.concatenate([ef[i] for i in mine])), nf=t(np.concatenate([nf[i] for i in mine])), gf=t(gf[mine]),
                         eo=torch.empty((g.n_edges, 3), device=dev), no=torch.empty((g.n_nodes, 4), device=dev),
                         gl=torch.empty((len(mine), 5), device=dev), gall=torch.full((G, 5), float("nan"), device=dev), ws=ws,
                         stream=torch.cuda.current_stream(dev).cuda_stream))
        hs.append(g); bps.append(bp); keep.append(blk)
    d = C.c_void_p()
    devs = (C.c_int32 * n)(*range(n))
    gn._lib.check(lib.gnx_dist_create(devs, n, p64(off), p64(ids), G, 5, C.byref(d)))
    try:
        arr = lambda f: (C.c_void_p * n)(*[f(b) for b in bufs])
        nbytes = (C.c_size_t * n)(*[b["ws"].numel() for b in bufs])
        bparr = (C.c_void_p * n)(*[C.addressof(bp) for bp in bps])
        for _ in range(2):
            gn._lib.check(lib.gnx_dist_block_forward(d, arr(lambda b: b["g"]._h.value), bparr, arr(lambda b: b["ef"].data_ptr()), arr(lambda b: b["nf"].data_ptr()),
                                                     arr(lambda b: b["gf"].data_ptr()), arr(lambda b: b["eo"].data_ptr()), arr(lambda b: b["no"].data_ptr()),
                                                     arr(lambda b: b["gl"].data_ptr()), arr(lambda b: b["gall"].data_ptr()), arr(lambda b: b["ws"].data_ptr()),
                                                     nbytes, 0, arr(lambda b: b["stream"])))
            for r in range(n):
                torch.cuda.synchronize(r)
            ref, scale = O.block_forward_sparse(p, O.csc_from_adj(adjs), np.concatenate(ef)[None], np.concatenate(nf)[None], gf[None], return_scale=True)
            for r in range(n):
                U.assert_close(bufs[r]["gall"].cpu().numpy()[None], ref[2], scale[2], f"gf' in original graph order on device {r}")
    finally:
        gn._lib.check(lib.gnx_dist_destroy(d))


def test_gnx_dist_replay_form_world_1(gn):
    """gnx_dist_block_forward_steps (the replay form of the sharded forward): n_steps forwards per device captured into ONE hipGraph per
    device, gf' rows written straight into the stacked send buffer, ONE all-gather (real RCCL, world 1), stacked tables restored to ORIGINAL
    graph order.  Eager (GNX_FLAG_NO_GRAPH), first call (eager + capture) and replays give the same bits; every step equals the oracle;
    a second argument set gets its own graphs; growing n_steps re-sizes the stacked buffers."""
    import torch
    from graphnets_jl_amd.dist import DistBlockRunner
    rng = np.random.default_rng(91)
    G, E = 48, 6000
    colptrs, rowvals, nn = bench.make_hetero(1234, G, E)
    order = rng.permutation(G).astype(np.int64)  # the single rank's local order of its graphs: the index table is exercised
    dims = ((10, 5, 3), (3, 4, 5))
    p = O.make_block_params(rng, *dims)
    graphs_of = lambda r: ([colptrs[i] for i in order], [rowvals[i] for i in order], [nn[i] for i in order])
    run = DistBlockRunner([torch.cuda.current_device()], [order], graphs_of, lambda dev: U.block_from_params(gn, p, device=dev), dims, n_sets=3, max_steps=5)
    try:
        g = run.handles[0]
        csc = (*g.csc(), g.node_off, g.edge_off)

        def oracle_tables(first, n):
            out = []
            for s in range(n):
                b = run.sets[0][(first + s) % 3]
                ref, scale = O.block_forward_sparse(p, csc, b["ef"].cpu().numpy()[None], b["nf"].cpu().numpy()[None], b["gf"].cpu().numpy()[None], return_scale=True)
                out.append((ref, scale))
            return out

        def check(first, n, what):
            run.synchronize()
            got = run.gall[0][:n].cpu().numpy()
            for s, (ref, scale) in enumerate(oracle_tables(first, n)):
                back = np.empty_like(ref[2][0]); sc = np.empty_like(scale[2][0])
                back[order] = ref[2][0]; sc[order] = scale[2][0]  # local row k is original graph order[k]
                U.assert_close(got[s][None], back[None], sc[None], f"{what}: step {s} gf' in original graph order")
                b = run.sets[0][(first + s) % 3]
                U.assert_close(b["eo"].cpu().numpy()[None], ref[0], scale[0], f"{what}: step {s} ef'")
            return got.copy()

        run.run(0, 4, flags=gn._lib.FLAG_NO_GRAPH)
        eager = check(0, 4, "eager")
        run.gall[0].fill_(float("nan"))
        run.run(0, 4)                       # first sight: eager pass + capture
        first = check(0, 4, "first call")
        for _ in range(3):                  # replays: one hipGraphLaunch per device
            run.gall[0].fill_(float("nan"))
            run.run(0, 4)
            again = check(0, 4, "replay")
            assert np.array_equal(again, eager) and np.array_equal(first, eager)
        run.run(1, 2); check(1, 2, "second argument set")
        run.run(1, 2); check(1, 2, "second argument set, replay")
        run.run(0, 4); assert np.array_equal(check(0, 4, "first set again"), eager)
        run.run(2, 5); check(2, 5, "more steps than the buffers held: re-sized")
        run.run(2, 5); check(2, 5, "re-sized, replay")
        # argument checks
        import ctypes as C
        assert gn._lib.load().gnx_dist_block_forward_steps(run._d, 0, None, None, None, None, None, None, None, None, None, None, 0, None) == gn._lib.ERR_INVALID_ARG
    finally:
        run.close()


def test_shared_graph_batch_sharded_by_replica_three_virtual_ranks(gn):
    """a shared-adjacency batch shards over its data batch.  7 replicas of one 300-node graph over 3 virtual ranks (2 + 2 + 3) on this GPU:
    every rank runs the HIP block on ITS replicas with the one shared handle, the padded send tables are concatenated as all_gather_into_tensor
    would and the C boundary's gather plan restores replica order — ef', nf', gf' bit-identical to the one-call result."""
    import torch
    from graphnets_jl_amd.dist import gather_plan, partition_replicas
    rng = np.random.default_rng(61)
    dims = ((10, 5, 3), (3, 4, 5))
    cp, rv = U.er_csc(rng, 300, 3000)
    g = gn.GNGraphBatch.from_csc([cp], [rv], [300])
    R, world = 7, 3
    p = O.make_block_params(rng, *dims)
    blk = U.block_from_params(gn, p)
    ef, nf, gf = U.packed_inputs(rng, R, g.n_edges, g.n_nodes, 1, dims[0])
    whole = blk(U.to_nt(gn, g, ef, nf, gf))
    shards = partition_replicas(R, world)
    assert [len(s) for s in shards] == [2, 2, 3]
    src, max_count = gather_plan(shards)
    table = torch.zeros((world * max_count, 5), device=g.device)
    for rank, mine in enumerate(shards):
        y = blk(U.to_nt(gn, g, ef[mine], nf[mine], gf[mine]))
        assert torch.equal(y.ef, whole.ef[:, :, mine[0]:mine[-1] + 1]) and torch.equal(y.nf, whole.nf[:, :, mine[0]:mine[-1] + 1])
        table[rank * max_count:rank * max_count + len(mine)] = y.gf.permute(2, 1, 0)[:, 0, :]
    gathered = table[torch.from_numpy(src.astype(np.int64)).to(g.device)]
    assert torch.equal(gathered, whole.gf.permute(2, 1, 0)[:, 0, :])


@pytest.mark.timeout(300)
def test_graphed_captures_beside_a_live_process_group():
    """api.Graphed (and bench.py) capture with capture_error_mode="thread_local": torch's NCCL watchdog thread polls the events of collectives in
    flight, which inside a "global" capture is an error thrown in THAT thread — the process aborts (profiles/r06_capture_vs_watchdog.log: the
    default mode aborts every time).  A child process: one-rank RCCL group, collectives in flight, forty Graphed captures + replays."""
    child = r"""
import os, socket, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import graphnets_jl_amd as gn
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
rng = np.random.default_rng(0)
adjs = [(rng.random((n, n)) < 0.3).astype(np.int64) for n in (9, 14, 30)]
x = gn.batch(dict(graphs=adjs, ef=[rng.random((10, int(a.sum())), dtype=np.float32) for a in adjs], nf=[rng.random((5, a.shape[0]), dtype=np.float32) for a in adjs], gf=None))
blk = gn.GNBlock((10, 5, 0), (3, 4, 5))
ref = blk(x)
t = torch.ones(1 << 20, device="cuda")
for i in range(40):
    works = [dist.all_reduce(t, async_op=True) for _ in range(4)]
    gr = gn.Graphed(blk, x, warmup=1)
    y = gr(x)
    for w in works:
        w.wait()
    t.fill_(1.0)
    assert torch.equal(y.ef, ref.ef) and torch.equal(y.gf, ref.gf)
torch.cuda.synchronize(); dist.destroy_process_group(); print("ok")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.returncode, out.stderr[-1500:])

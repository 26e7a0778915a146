#!/usr/bin/env python3
"""bench.py — edges updated/sec of the GNBlock forward on a 1M-edge batch (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dims readme|core] [--workload c2|hetero]

A "step" is one GNBlock forward (edge + node + graph update) over one resident batch.
  N = 1 : BASELINE configs[1] — one shared Erdős–Rényi graph, 100k nodes / 1M edges, batch_size 1.  The K timed steps
          are ONE gnx_block_forward_steps call (the library's loop over batches: K forwards in order, step i's graph update at the
          front of step i + 1's launch, the last one flushed inside the call) captured into ONE hipGraph (the step is ~20 µs
          of GPU work; eager launches from Python would time the host) and rotate over NSETS disjoint buffer sets so the
          footprint (>256 MiB) defeats the Infinity Cache: `value` is a cache-cold, HBM-resident number.  `two_launch_form`:
          the same K steps as K separate gnx_block_forward calls (rounds 1-5's headline form; `--separate-calls` makes it the
          timed form).  `warm_ms_per_step` (two buffer sets, cache-resident) is extra.
  N > 1 : BASELINE configs[4], STRONG scaling of ONE FIXED batch: the 4096-graph heterogeneous batch (32..256 nodes, 1M edges, seed 5 — the
          metric's "1M-edge batch") is sharded BY GRAPH over the N ranks with the product's partitioner (equal graph counts, snake order by
          edge count: graphnets.jl_amd/dist.py); every rank builds the handle of its own 4096/N graphs; graphs never cross ranks; the only
          collective is the RCCL all-gather of gf' into original graph order (stacked over the steps of one hipGraph replay, on a side
          stream).  The same graphs at every N, so the 1/2/4/8 values are one scaling curve; every line also carries
          `single_gpu_same_workload` (rank 0 runs the WHOLE batch alone first), `with_allgather` / `without_allgather`, and
          `secondary.c5w`: the same 4096 graphs at 8M edges through the same ranks.  `--scaling weak` keeps the N x 512-graph /
          N x 1M-edge form (per-GPU shard fixed).  `--dist-backend gnx`: the same measurement driven by ONE process through gnx_dist_*.
          `python bench.py --gpus N` starts its own N ranks (python -m torch.distributed.run, before anything touches a
          GPU); under an external torchrun (RANK / WORLD_SIZE set) it is one of the ranks.
Timing: W warm-up steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize(), MAX over ranks; the K-step
region is run three times and the MEDIAN is reported (`timing` says so).  Rank 0 prints ONE JSON line.
`roofline` comes from a second pass over the same K steps with per-kernel HIP events (gnx_profile_*); `cpu_baseline` from
the oracle's C restatement (test infrastructure) on the host cores.
The default N = 1 invocation also carries `"secondary"`: the other BASELINE configs measured by the same script in the same run —
core dims on C2, C3 (configs[2]), C5 on one GPU (configs[4]'s batch), C4 (configs[3]) and README ex.3 at its own widths — each with
ms_per_step, roofline (traffic + provenance), cpu_baseline and, for the hetero batches, the cost of `batch()` itself.  They run as
child processes BEFORE this process touches the GPU, one at a time (`--no-secondary` skips them).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFS = 157.3  # exact-f32 MFMA (v_mfma_f32_32x32x2_f32); no xf32 on gfx950
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md); an fp32 product carried by six bf16 terms (k_ffn_x6): / 6
NSETS = 8                  # rotating buffer sets: 8 x ~60 MB > 256 MiB Infinity Cache
DIMS = {"readme": ((10, 5, 0), (3, 4, 5)), "core": ((128, 64, 32), (128, 64, 32)),
        "odd": ((7, 3, 2), (5, 6, 1)), "mid": ((20, 10, 4), (12, 9, 3))}  # odd: fused kernel specialised at run time (GNX_JIT=0: generic kernels); mid: generic/MFMA path
KERNEL_SOURCES = {"narrow": ("gnx_wave_kernel.h", "gnx_device.h", "gnx_narrow.hip", "gnx_forward.hip", "gnx_graphs.cpp"),
                  "wide": ("gnx_wide.hip", "gnx_edge_x6.hip", "gnx_x6_stats.h", "gnx_device.h", "gnx_forward.hip", "gnx_graphs.cpp")}


def kernel_source_sha(din, dout):
    """sha256 over the sources of the kernels that run this width set (fused narrow path or matrix-core path): a committed
    traffic profile is only quoted while it describes THIS code."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES["wide" if max(tuple(din) + tuple(dout)) > 32 else "narrow"]:
        with open(os.path.join(ROOT, "graphnets.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def make_c2(seed=2, N=100_000, E=1_000_000):
    """E distinct directed pairs (self-loops possible), reference edge order (dst, then src)."""
    rng = np.random.default_rng(seed)
    k = np.unique(rng.integers(0, N * N, int(E * 1.1)))
    k = np.sort(rng.permutation(k)[:E])
    dst, src = k // N, k % N
    colptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(colptr, dst + 1, 1)
    return [np.cumsum(colptr)], [src.astype(np.int64)], [N]


def hetero_spec(seed, G=512, E=1_000_000):
    """SURVEY §8d C3/C5 law: G graphs, n_g ~ U{32..256}, constant density, exactly E edges in total.
    Returns (n_g[G], e_g[G]) — cheap, every rank computes it for the whole batch."""
    rng = np.random.default_rng(seed)
    n = rng.integers(32, 257, G)
    dens = E / float((n.astype(np.int64) ** 2).sum())
    eg = np.floor(dens * n.astype(np.float64) ** 2).astype(np.int64)
    short = E - int(eg.sum())
    eg[np.argsort(-n, kind="stable")[:short]] += 1
    return n.astype(np.int64), eg


def hetero_graph(seed, i, n, e):
    """Graph i of the batch `seed`: e distinct directed pairs of an n-node graph in reference edge order, drawn from a
    generator that depends on (seed, i) only — a rank generates just the graphs of its shard."""
    rng = np.random.default_rng([int(seed), int(i)])
    n, e = int(n), int(e)
    k = np.sort(rng.choice(n * n, e, replace=False))
    cp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(cp, k // n + 1, 1)
    return np.cumsum(cp), (k % n).astype(np.int64)


def make_hetero(seed, G=512, E=1_000_000, only=None):
    """The whole batch (or the graphs `only`, ascending original ids) as per-graph CSC lists."""
    n, eg = hetero_spec(seed, G, E)
    ids = range(G) if only is None else [int(i) for i in only]
    colptrs, rowvals = [], []
    for i in ids:
        cp, rv = hetero_graph(seed, i, n[i], eg[i])
        colptrs.append(cp); rowvals.append(rv)
    return colptrs, rowvals, [int(n[i]) for i in ids]


def algorithmic_bytes(E, N, G, din, dout):
    """every tensor once, gathers counted per node, no intermediates."""
    (de, dn, dg), (oe, on, og) = din, dout
    ke, kn, kg = de + 2 * dn + dg, oe + dn + dg, oe + on + dg
    b = 4 * (E * (de + oe) + N * (dn + on) + G * (dg + og)) + 4 * E + 4 * (N + 1)
    if G > 1:
        b += 8 * (G + 1)
    return b + 4 * (ke * oe + oe + kn * on + on + kg * og + og)


def algorithmic_flops(E, N, G, din, dout):
    (de, dn, dg), (oe, on, og) = din, dout
    return 2 * (E * (de + 2 * dn + dg) * oe + N * (oe + dn + dg) * on + G * (oe + on + dg) * og)


def executed_flops(E, N, G, din, dout):
    """FLOPs the MFMA path actually executes: gf is folded into a per-graph bias and, when dn >= 16, the nf columns of the
    edge function are projected once per NODE (W*[ef;nf_s;nf_d] = We*ef + (Ws*nf)[src] + (Wd*nf)[dst])."""
    (de, dn, dg), (oe, on, og) = din, dout
    edge = E * de * oe + (2 * N * dn * oe if dn >= 16 else 2 * E * dn * oe)
    return 2 * (edge + N * (oe + dn) * on + G * (oe + on + dg) * og + G * dg * (oe + on))


FLAG_FFN_FP32, FLAG_EDGE_FP32 = 0x20, 0x40  # include/gnx.h: the call's arithmetic (GNX_FLAG_FP32_MFMA = both)


def fp32_forms(flags):
    """(FeedForwards on the fp32 matrix instruction?, edge update / projections?) for a run with these call flags: the flags themselves, or the
    process-wide defaults the library reads from the environment"""
    return (bool(flags & FLAG_FFN_FP32) or os.environ.get("GNX_FFN_FP32", "0") not in ("", "0"),
            bool(flags & FLAG_EDGE_FP32) or os.environ.get("GNX_EDGE_FP32", "0") not in ("", "0"))


def glorot(rng, out_d, in_d):
    s = np.sqrt(6.0 / max(in_d + out_d, 1))
    return rng.uniform(-s, s, size=(out_d, in_d)).astype(np.float32)


def _dense_np(gn, W, b, act, dev):
    return gn.Dense.from_numpy(W, b, act, device=dev)


def c4_model(gn, torch, core, dev, seed=0):
    """README ex.3 (README.md:129-145): encoder (10,5,0) => core, two GNCore(core), decoder core => (3,4,5).  Parameters are drawn in numpy
    (glorot weights, zero biases, LayerNorm 1 / 0 — Flux's initialisation) so that the CPU port can be timed on the very same model."""
    from oracle import gn_oracle as O  # parameter shapes only; the layers below are the HIP path
    rng = np.random.default_rng(seed)
    ps = [("block", O.make_block_params(rng, (10, 5, 0), core, random_bias=False)),
          ("core", O.make_core_params(rng, core, random_bias=False)), ("core", O.make_core_params(rng, core, random_bias=False)),
          ("block", O.make_block_params(rng, core, (3, 4, 5), random_bias=False))]

    def block(p):
        b = gn.GNBlock(p["in_dims"], p["out_dims"], device=dev)
        b.edgefn, b.nodefn, b.graphfn = (_dense_np(gn, p["W" + t], p["b" + t], "identity", dev) for t in "eng")
        return b
    layers = []
    for kind, p in ps:
        if kind == "block":
            layers.append(block(p))
            continue
        c = gn.GNCore(p["dims"], device=dev, eps=p["eps"], eps_mode=p["eps_mode"])
        c.block = block(p["block"])
        for t, l1, l2 in zip("eng", (c.gn1.edgeln, c.gn1.nodeln, c.gn1.graphln), (c.gn2.edgeln, c.gn2.nodeln, c.gn2.graphln)):
            for ln, name in ((l1, "ln1"), (l2, "ln2")):
                p[f"{name}_{t}_gamma"][:] = 1.0; p[f"{name}_{t}_beta"][:] = 0.0
                ln.gamma = torch.from_numpy(p[f"{name}_{t}_gamma"]).to(dev); ln.beta = torch.from_numpy(p[f"{name}_{t}_beta"]).to(dev)
        mk = lambda t: (_dense_np(gn, p[f"ff_{t}_W1"], p[f"ff_{t}_b1"], "relu", dev), _dense_np(gn, p[f"ff_{t}_W2"], p[f"ff_{t}_b2"], "identity", dev))
        c.ffwd.eff, c.ffwd.nff, c.ffwd.gff = mk("e"), mk("n"), mk("g")
        layers.append(c)
    return layers, ps


CLOCK_WARMUP_MS = 150.0  # untimed load before every timed region (see spin_up)


def spin_up(torch, dev, run, ms=None):
    """UNTIMED: keeps the GPU under the very load that is about to be timed for `ms` milliseconds.  After an idle period (graph capture,
    host-side setup) the MI355X's power management needs tens of milliseconds of continuous load to settle: in a rocprofv3 trace of this
    script the same k_rows_gemm launch takes 416 us right after the idle gap, 480-520 us a few milliseconds later and 372-380 us from
    ~50 ms of load on (profiles/r04_clock_transient_core.txt); a 0.5-6 ms timed region started cold measures that transient, not the
    kernels.  Replays `run` (batches of 4 between synchronisations, so the queue stays short) until the time is up."""
    ms = CLOCK_WARMUP_MS if ms is None else ms
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            run()
        torch.cuda.synchronize(dev)
        n += 4
    return n


def calibrated_kernel_us(prof, per=None):
    """Per-scope kernel time in us from gnx_profile_read: dispatch timestamps (hipExtLaunchKernel start / stop events).  Those read a CONSTANT
    ~3.9 us above rocprofv3's kernel trace on the same launches, whatever the kernel's length (an empty kernel: 7.8 vs 3.98 us; k_block_wave:
    23.94 vs 20.08; the edge GEMM: 446.6 vs 442.6 — profiles/r04_selfcheck_*.txt), so (empty kernel timed live the same way) - (rocprofv3's
    figure for it, profiles/calibration.json) is subtracted per KERNEL of a scope.  per = forwards the pass ran (else: per scope entry)."""
    with open(os.path.join(ROOT, "profiles", "calibration.json")) as f:
        calib = json.load(f)
    null = prof.get("__empty_bracket__")
    null_us = null["total_ms"] / max(null["launches"], 1) * 1e3 if null else calib["null_kernel_rocprof_us"]
    off = max(null_us - calib["null_kernel_rocprof_us"], 0.0)
    out = {}
    for k, v in prof.items():
        if k == "__empty_bracket__":
            continue
        n_kernels = v.get("kernels", v["launches"])
        out[k] = round(max(v["total_ms"] * 1e3 - off * n_kernels, 0.0) / max(per or v["launches"], 1), 3)
    return out


def c4_cpu_baseline(ps, budget_s=32.0):
    """The oracle's C restatement of the 4-layer model on the host cores.  The sample grows from an Erdos-Renyi graph of the C2 law at 1/16 of C2's
    size by x4 up to the FULL workload (C2 itself: 100k nodes / 1M edges, ~8 s per forward of the wide model on 128 threads) while the time
    already spent plus the next size's projected forwards stays inside `budget_s` seconds of CPU work; edges/s through the model at the largest
    size reached (the line names it)."""
    from oracle import c_port
    cores = min(os.cpu_count() or 1, c_port.max_threads())
    rng = np.random.default_rng(9)
    best = None
    scale = 1.0 / 16
    t_all = time.perf_counter()
    while True:
        colptrs, rowvals, nn = make_c2(seed=2, N=int(100_000 * scale), E=int(1_000_000 * scale))
        N, E = nn[0], len(rowvals[0])
        csc = (colptrs[0], rowvals[0], np.array([0, N]), np.array([0, E]))
        x = (rng.random((1, E, 10), dtype=np.float32), rng.random((1, N, 5), dtype=np.float32), None)

        def fwd():
            y = x
            for kind, p in ps:
                y = (c_port.block_forward if kind == "block" else c_port.core_forward)(p, csc, *y, nthreads=cores)
            return y
        t0 = time.perf_counter(); fwd(); t_first = time.perf_counter() - t0
        times = [] if scale < 1.0 else [t_first]  # (the full-size forward is seconds: its first run counts)
        while len(times) < 2 or (sum(times) < 1.5 and len(times) < 8):
            t0 = time.perf_counter(); fwd(); times.append(time.perf_counter() - t0)
        t = float(np.median(times))
        best = dict(value=round(E / t, 1), unit="edges/s", cores=cores, kind="port",
                    sample=(f"{len(times)} forwards of the 4-layer model on " + ("the FULL workload (the C2 graph's law at full size: " if scale >= 1.0 else "an Erdos-Renyi graph of the C2 law with ") +
                            f"{N} nodes / {E} edges" + (")" if scale >= 1.0 else f", {scale:g} of C2") + f"; median {t * 1e3:.0f} ms per forward; oracle/gn_oracle_c.c with OpenMP on {cores} threads"))
        spent = time.perf_counter() - t_all
        if scale >= 1.0 or spent + (2 if scale * 4 >= 1.0 else 3) * (4.4 * t) > budget_s:  # next size: ~4.4x the time per forward; two forwards at full size, else three
            return best
        scale *= 4


def bench_c4(args, gn, torch, dev, c_abi=None):
    """BASELINE configs[3]: Encoder -> 2 x GNCore -> Decoder (README Example 3) at core_dims (128,64,32) on the C2 graph.
    One "step" = the whole 4-layer model forward; reported as edges/s through the model."""
    colptrs, rowvals, nn = make_c2()
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
    core = tuple(int(v) for v in args.core_dims.split(","))
    model, ps = c4_model(gn, torch, core, dev)
    for layer in model:
        layer.flags = args.flags  # the forms of every call of the run (--flags 96 = GNX_FLAG_FP32_MFMA: every product on the fp32 matrix instruction)
        if not args.no_prepare:
            layer.prepare()  # gnx_block_prepare / gnx_core_prepare: the weight blocks in the kernels' forms, ONCE — as `model |> device` happens once (examples/sort/sort.jl:29,89)
    ffn_fp32, edge_fp32 = fp32_forms(args.flags)
    tg = torch.Generator(device=dev); tg.manual_seed(1)
    x = gn.NT(g, torch.rand((1, g.n_edges, 10), generator=tg, device=dev).permute(2, 1, 0),
              torch.rand((1, g.n_nodes, 5), generator=tg, device=dev).permute(2, 1, 0), None)

    n_fw = [0]  # forwards executed by this process (tools/summarize_prof.py --model-traffic divides a profiled run's bytes by it)

    def fwd():
        n_fw[0] += 1
        y = x
        for layer in model:
            y = layer(y)
        return y
    for _ in range(max(args.warmup, 1)):
        fwd()
    torch.cuda.synchronize(dev)
    spin_up(torch, dev, fwd)  # (the per-kernel figures below are steady-state figures too)
    gn.profile_reset(); gn.profile_enable(True)
    for _ in range(3):
        fwd()
    gn.profile_calibrate(20, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    gn.profile_enable(False)
    praw = gn.profile_read(); gn.profile_reset()
    kern = calibrated_kernel_us(praw, per=3)
    K = args.steps
    t0 = time.perf_counter()
    for _ in range(K):
        fwd()
    torch.cuda.synchronize(dev)
    dt_eager = (time.perf_counter() - t0) / K
    # the timed region: ONE forward captured into a hipGraph, replayed K times (at narrow widths an eager forward is ten ~20-us launches
    # and costs the host about as much as the GPU: the eager figure then measures the Python mirror, not the kernels); median of 3 regions
    def model_fn(t):
        y = t
        for layer in model:
            y = layer(y)
        return y
    graphed = gn.Graphed(model_fn, x)
    n_fw[0] += 2  # (Graphed: one eager call, one captured)

    def replay():
        n_fw[0] += 1
        graphed.graph.replay()
    for _ in range(3):
        replay()
    torch.cuda.synchronize(dev)
    spin_up(torch, dev, replay)
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            replay()
        torch.cuda.synchronize(dev)
        reps.append((time.perf_counter() - t0) / K)
    dt = float(np.median(reps))
    E, N = g.n_edges, g.n_nodes
    if core == (128, 64, 32):
        aflops = 699.2e9  # whole-model algorithmic FLOPs at 1M edges
    else:
        aflops = float("nan")
    # executed: enc + dec blocks, two cores = block + FeedForward 16*(E*de^2 + N*dn^2 + G*dg^2) each
    ce, cn, cg = core
    ex = (executed_flops(E, N, 1, (10, 5, 0), core) + executed_flops(E, N, 1, core, (3, 4, 5)) +
          2 * (executed_flops(E, N, 1, core, core) + 16 * (E * ce * ce + N * cn * cn + cg * cg)))
    tkey = "c4" if core == (128, 64, 32) else "c4_" + "-".join(map(str, core))
    sha = model_source_sha(core)
    traffic, tsrc = load_traffic(tkey, "__model__", sha)
    if (ffn_fp32 or edge_fp32) and ce == 128:  # (the profiled traffic is the six-term form's)
        traffic, tsrc = None, {"note": "profiles/traffic_c4.json was measured with k_ffn_x6; round 3 measured this form at 15.8 GB"}
    # The roof of the model: its executed flops at the rate of the instruction that carries them.  The two edge FeedForwards at width 128 run as
    # k_ffn_x6 — every fp32 product as six bf16 matrix-core terms with fp32 accumulation (csrc/gnx_ffn_x6.hip; as accurate as the fp32 MFMA:
    # tests/test_gpu_core.py) — unless the calls carry GNX_FLAG_FFN_FP32; everything else on the fp32 MFMA.
    x6 = ce == 128 and not ffn_fp32
    x6_flops = 2 * 16 * E * ce * ce if x6 else 0
    if ce == 128 and not edge_fp32:
        x6_flops += 2 * 2 * E * ce * ce  # the cores' projected edge updates (k_edge_x6: K = 128 -> 128 per edge)
    t_roof = x6_flops / (MFMA_BF16_PEAK_TFS / 6 * 1e12) + (ex - x6_flops) / (MFMA_F32_PEAK_TFS * 1e12)
    line = {"metric": "edges/sec through Encoder->2xGNCore(%s)->Decoder, 1M-edge graph (BASELINE configs[3])" % ",".join(map(str, core)),
            "value": round(E / dt, 1), "unit": "edges/s", "ms_per_step": round(dt * 1e3, 4), "steps": K, "dtype": "f32",
            "roofline": {"bound": "mfma", "achieved": round(ex / dt / 1e12, 2), "peak": round(ex / t_roof / 1e12, 1), "unit": "TFLOP/s",
                         "frac": round(t_roof / dt, 4),
                         "counts": "EXECUTED flops of the whole model / whole-step time; peak = the same flops at the rate of the instruction that carries them "
                                   "(fp32 MFMA %.1f TFLOP/s; the edge FeedForwards as six bf16 terms per fp32 product: %.0f / 6 = %.1f)" % (MFMA_F32_PEAK_TFS, MFMA_BF16_PEAK_TFS, MFMA_BF16_PEAK_TFS / 6),
                         "frac_of_fp32_mfma_roof": round(ex / dt / 1e12 / MFMA_F32_PEAK_TFS, 4),
                         "flops_on_bf16_six_terms": x6_flops,
                         "arithmetic": ("fp32 in, fp32 out, fp32 accumulation; the edge FeedForward's products as hi/mid/lo bf16 parts (24 mantissa bits, exact split), six matrix-core terms" if x6
                                        else "fp32 MFMA throughout"),
                         "executed_flops": ex, "algorithmic_flops": aflops, "algorithmic_tflops": round(aflops / dt / 1e12, 2),
                         "traffic": traffic, "traffic_source": tsrc},
            "kernel_us_one_forward": kern,
            "config": {"workload": "C4: Encoder -> 2 x GNCore(%s) -> Decoder on the C2 graph (100k nodes / 1M edges); one step = the whole model forward" % ",".join(map(str, core)),
                       "launch": "one forward captured into a hipGraph, replayed %d times" % K,
                       "timing": "median of 3 regions (%s ms/step)" % [round(r * 1e3, 4) for r in reps],
                       "eager_ms_per_step": round(dt_eager * 1e3, 4),
                       # (for tools/summarize_prof.py --model-traffic: bytes of a profiled run / forwards = bytes per forward)
                       "forwards_executed": n_fw[0]}}
    if max(core) < 32:  # narrow widths run on the vector units, not the matrix cores: the model is priced against HBM like a narrow block
        ab = (algorithmic_bytes(E, N, 1, (10, 5, 0), core) + 2 * algorithmic_bytes(E, N, 1, core, core) + algorithmic_bytes(E, N, 1, core, (3, 4, 5)) +
              2 * 4 * (3 * 8 * (ce * ce + cn * cn + cg * cg) + 4 * (ce + cn + cg)))  # + the cores' FeedForward and LayerNorm parameters
        line["roofline"] = {"bound": "hbm", "achieved": round(ab / dt / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ab / dt / 1e9 / HBM_PEAK_GBS, 4),
                            "counts": "algorithmic bytes of the four layers (every layer's inputs and outputs once, SURVEY 8d; a core = its block's bytes) / whole-step time",
                            "algorithmic_bytes": ab, "executed_flops": ex, "traffic": traffic, "traffic_source": tsrc}
    assert line["roofline"]["frac"] <= 1.0
    if c_abi is not None:  # the same model as ONE gnx_model driven from plain C (library-owned intermediates and hipGraph)
        line["c_abi_ms_per_step"] = round(c_abi["model_us_per_step"] * 1e-3, 4) if "model_us_per_step" in c_abi else None
        line["c_abi"] = ({"vs_torch_captured": round(c_abi["model_us_per_step"] * 1e-3 / (dt * 1e3), 4), "what": c_abi.get("model_what"), "reps_us": c_abi.get("model_reps_us"),
                          "event_us_per_step": c_abi.get("model_event_us_per_step"), "batch_ms": c_abi.get("batch_ms"), "program": "tests/c/abi_bench.c --mode c4"}
                         if "model_us_per_step" in c_abi else c_abi)
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = c4_cpu_baseline(ps)
    line.update({"n_gpus": 1, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
    emit(line, args, "c4" if core == (128, 64, 32) else "c4_" + "-".join(map(str, core)))


def measure_dist_gnx(args, gn, torch, n, din, dout, Gtot, Etot, seed, K, W):
    """One sharded measurement through gnx_dist_block_forward_steps: the batch partitioned over n devices driven by THIS process; the whole batch on
    device 0 first (single_gpu_same_workload); then the K-step region with and without the grouped all-gather, median of three each."""
    from graphnets_jl_amd.dist import DistBlockRunner, partition_graphs
    single = None
    if not args.no_single_gpu_leg:
        torch.cuda.set_device(0)
        single = single_gpu_same_workload(gn, torch, torch.device("cuda", 0), seed, Gtot, Etot, din, dout, K, W, args.flags)
    n_all, e_all = hetero_spec(seed, Gtot, Etot)
    shards = partition_graphs(e_all, n)
    M = max(m for m in range(1, 65) if K % m == 0)
    per_shard = algorithmic_bytes(Etot // n, int(n_all.sum()) // n, Gtot // n, din, dout)
    nsets = rotating_sets(per_shard)
    run = DistBlockRunner(list(range(n)), shards, lambda r: make_hetero(seed, Gtot, Etot, only=shards[r]), lambda dev: bench_weights(gn, din, dout, dev), (din, dout),
                          n_sets=nsets, max_steps=M)
    assert run.edges == Etot and sum(g.n_graphs for g in run.handles) == Gtot, "the shards do not add up to the batch"

    def region(flags=0):
        for j in range(K // M):
            run.run((j * M) % nsets, M, flags=flags)

    def timed(flags):
        reps = []
        for _ in range(3):
            run.synchronize()
            t0 = time.perf_counter()
            region(flags)
            run.synchronize()
            reps.append(time.perf_counter() - t0)
        return sorted(reps)
    for _ in range(max(2, -(-W // M))):  # warm-up: every (first set, M) argument set the region uses gets captured here
        region()
    run.synchronize()
    t_w = time.perf_counter()
    while (time.perf_counter() - t_w) * 1e3 < CLOCK_WARMUP_MS:  # the same untimed load as spin_up
        region(); run.synchronize()
    reps = timed(0)
    reps_wo = timed(gn._lib.FLAG_DIST_NO_GATHER)
    # the eager form of the same entry point (no per-device hipGraph): what the replay form saves
    t0 = time.perf_counter()
    region(gn._lib.FLAG_NO_GRAPH)
    run.synchronize()
    dt_eager = time.perf_counter() - t0
    res = {"E_job": run.edges, "G_job": Gtot, "seed": seed, "M": M, "nsets": nsets, "dt": reps[1], "dt_without": reps_wo[1], "reps": reps, "reps_without": reps_wo,
           "per_rank": [[g.n_edges, g.n_nodes, g.n_graphs] for g in run.handles], "single": single, "eager_ms_per_step": round(dt_eager / K * 1e3, 6)}
    run.close()
    del run
    torch.cuda.empty_cache()
    return res


def bench_dist_gnx(args):
    """BASELINE configs[4] through the C boundary's OWN multi-GPU entry (gnx_dist_*): ONE process, --gpus devices.  Default `--scaling strong`: the
    FIXED 4096-graph / 1M-edge batch (and C5w beside it) partitioned by the product's partitioner over the devices; `--scaling weak`: N x 512 graphs.
    K steps = K / M calls of gnx_dist_block_forward_steps with M steps each (one hipGraph launch per device + ONE grouped all-gather of the M
    stacked gf' tables per call).  Timed region: all devices synchronised on both sides; median of three."""
    import torch
    import graphnets_jl_amd as gn
    n = args.gpus
    assert torch.cuda.device_count() >= n, f"--dist-backend gnx drives {n} devices from this process; {torch.cuda.device_count()} visible"
    din, dout = DIMS[args.dims] if args.dims in DIMS else tuple(tuple(int(v) for v in part.split(",")) for part in args.dims.split(":"))
    Gtot, Etot, seed = sharded_workload(args, n)
    K, W = args.steps, args.warmup
    res = measure_dist_gnx(args, gn, torch, n, din, dout, Gtot, Etot, seed, K, W)
    e = sharded_entry(res, K, n)
    M = res["M"]
    strong = args.scaling == "strong"
    abytes = sum(algorithmic_bytes(c[0], c[1], c[2], din, dout) for c in res["per_rank"])
    dt = res["dt"]
    line = {"metric": "edges updated/sec, GNBlock fwd, 1M-edge batch", "value": e["value"], "unit": "edges/s", "n_gpus": n, "steps": K, "warmup": W,
            "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: ONE FIXED heterogeneous batch of {Gtot} random graphs (32-256 nodes, {Etot / 1e6:g}M edges, seed {seed}) sharded by graph over {n} GPU(s) "
                                    f"— the same graphs at every N (strong scaling) — " if strong else
                                    f"ONE heterogeneous batch of {Gtot} random graphs (32-256 nodes, {Etot / 1e6:g}M edges, seed {seed}) sharded by graph over {n} GPU(s) (weak scaling) ") +
                                   "driven by ONE process through gnx_dist_block_forward_steps",
                       "dims": f"{din}=>{dout}", "edges_whole_job": e["edges_whole_job"], "graphs_whole_job": e["graphs_whole_job"], "per_rank_edges_nodes_graphs": e["per_rank_edges_nodes_graphs"],
                       "graphs_per_gpu": [c[2] for c in res["per_rank"]], "dist_backend": "gnx",
                       "launch": f"{K // M} calls of gnx_dist_block_forward_steps per region, {M} steps each over {res['nsets']} rotating buffer sets: one hipGraphLaunch per device + ONE grouped ncclAllGather of the {M} stacked gf' tables per call",
                       "timing": f"median of 3 regions ({e['with_allgather']['reps_us_per_step']} us/step)", "eager_ms_per_step": res["eager_ms_per_step"]},
            "with_allgather": e["with_allgather"], "without_allgather": e["without_allgather"], "single_gpu_same_workload": e["single_gpu_same_workload"],
            "roofline": {"bound": "hbm", "achieved": round(abytes / (dt / K) / 1e9, 2), "peak": HBM_PEAK_GBS * n, "unit": "GB/s", "frac": round(abytes / (dt / K) / 1e9 / (HBM_PEAK_GBS * n), 4),
                         "counts": "algorithmic bytes of every device's shard / whole-step time / (n_gpus x 8 TB/s)", "algorithmic_bytes": abytes}}
    if "speedup_vs_single_gpu_same_workload" in e:
        line["speedup_vs_single_gpu_same_workload"] = e["speedup_vs_single_gpu_same_workload"]
    if strong and (Gtot, Etot) == (4096, 1_000_000) and not args.no_secondary:
        c5w = measure_dist_gnx(args, gn, torch, n, din, dout, 4096, 8_000_000, 5, K, W)
        ew = sharded_entry(c5w, K, n)
        ew["what"] = "C5w (SURVEY 8e): the same 4096 graphs at configs[2]'s density — 8M edges — sharded over the same devices"
        line["secondary"] = {"c5w": ew}
    line["cpu_baseline"] = None
    emit(line, args, f"dist_gnx_n{n}")


def model_source_sha(core):
    """sha256 over the kernel sources a C4 model runs (both paths' files at wide widths, the narrow files otherwise)."""
    files = sorted(set(KERNEL_SOURCES["wide"] + KERNEL_SOURCES["narrow"] + (("gnx_ffn_fused.hip", "gnx_ffn_x6.hip", "gnx_generic.hip") if max(core) >= 32 else ("gnx_core_narrow.hip", "gnx_core_post_kernel.h"))))
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "graphnets.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


COMPACT_MAX_BYTES = 3072  # the driver parses the LAST stdout line; round 5's 20.6-KB line was not parsed: the headline stays under 3 KB


def _short(v, n):
    """prose cut to n characters (the full text is in the detail file)"""
    if not isinstance(v, str) or len(v) <= n:
        return v
    return v[:n - 1].rstrip() + "…"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(line, detail_path=None):
    """The line the driver parses: the contract's keys + `roofline` + `cpu_baseline`, numbers only, every prose field cut short — always below
    COMPACT_MAX_BYTES whatever the run measured.  Everything else of `line` (secondary configs, batch construction times, the C program's leg, the
    `what` / `counts` / `timing` texts) goes to the detail file and to an earlier, prefixed stdout line (emit)."""
    cfg, roof, cpu = line.get("config") or {}, line.get("roofline"), line.get("cpu_baseline")
    out = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["metric"] = _short(out["metric"], 120)
    c = {"workload": _short(cfg.get("workload"), 200)}
    c.update(_pick(cfg, ("dims", "edges_whole_job", "graphs_whole_job", "edges_per_gpu", "nodes_per_gpu", "graphs_per_gpu", "parallelism")))
    if isinstance(c.get("graphs_per_gpu"), list):
        c["graphs_per_gpu"] = c["graphs_per_gpu"][:8]
    c["launch"] = _short(cfg.get("launch"), 160)
    if "dist_backend" in cfg:
        c["dist_backend"] = _short(cfg["dist_backend"], 24)
    out["config"] = c
    if isinstance(roof, dict):
        r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "frac_whole_step", "frac_whole_step_two_launch", "algorithmic_bytes", "kernel", "kernel_us", "kernel_us_rocprofv3", "executed_flops", "matrix_roof_frac"))
        r["traffic"] = roof.get("traffic")  # (null stays null: the contract names the key)
        ts = roof.get("traffic_source")
        if isinstance(ts, dict):  # where the figure is from: counters collected in THIS run, or the committed profile of the same kernel sources
            r["traffic_measured"] = "this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over a child of the workload)" if ts.get("live") else _short("profiles/ (" + str(ts.get("file")) + ")", 60)
        wj = roof.get("whole_job")
        if isinstance(wj, dict):
            r["whole_job"] = _pick(wj, ("algorithmic_bytes", "achieved", "peak", "frac"))
        out["roofline"] = r
    else:
        out["roofline"] = None
    out["cpu_baseline"] = dict(_pick(cpu, ("value", "unit", "cores", "kind")), sample=_short(cpu.get("sample"), 140)) if isinstance(cpu, dict) else None
    # the other forms of the same step, as bare numbers (ms per step)
    for key, src in (("with_allgather_ms", line.get("with_allgather")), ("without_allgather_ms", line.get("without_allgather")),
                     ("single_gpu_same_workload_ms", line.get("single_gpu_same_workload")), ("two_launch_ms", line.get("two_launch_form")),
                     ("chained_ms", line.get("chained_graph_update")), ("pipelined_two_streams_ms", line.get("pipelined_two_streams"))):
        if isinstance(src, dict) and src.get("ms_per_step") is not None:
            out[key] = src["ms_per_step"]
    for k in ("c_abi_ms_per_step", "speedup_vs_single_gpu_same_workload"):
        if line.get(k) is not None:
            out[k] = line[k]
    c5w = (line.get("secondary") or {}).get("c5w")
    if isinstance(c5w, dict) and "value" in c5w:  # N > 1: the 8M-edge batch through the same ranks
        out["c5w"] = dict(_pick(c5w, ("value", "ms_per_step")), **{k: v["ms_per_step"] for k, v in (("without_allgather_ms", c5w.get("without_allgather")),
                          ("single_gpu_same_workload_ms", c5w.get("single_gpu_same_workload"))) if isinstance(v, dict) and v.get("ms_per_step") is not None})
    if detail_path:
        out["detail"] = detail_path
    # never above the cap: prose first, then the optional numbers
    for drop in (("config", "launch"), ("cpu_baseline", "sample"), ("c5w",), ("config", "workload")):
        if len(json.dumps(out)) <= COMPACT_MAX_BYTES:
            break
        if len(drop) == 1:
            out.pop(drop[0], None)
        elif isinstance(out.get(drop[0]), dict):
            out[drop[0]][drop[1]] = _short(out[drop[0]].get(drop[1]), 40)
    assert len(json.dumps(out)) <= COMPACT_MAX_BYTES, "bench.py: the headline line outgrew its cap"
    return out


def secondary_summary(sec):
    """`secondary` as numbers only (per config: value, ms_per_step, roofline bound / frac / kernel_us / traffic, CPU figure): the earlier stdout line"""
    out = {}
    for k, e in (sec or {}).items():
        if not isinstance(e, dict):
            continue
        if "error" in e:
            out[k] = {"error": _short(str(e["error"]), 80)}
            continue
        r = e.get("roofline") or {}
        s = _pick(e, ("value", "ms_per_step", "steps", "c_abi_ms_per_step"))
        s.update({"roof_" + kk: r[kk] for kk in ("bound", "frac", "frac_whole_step", "kernel", "kernel_us", "traffic", "algorithmic_bytes", "executed_flops") if r.get(kk) is not None})
        if isinstance(e.get("cpu_baseline"), dict):
            s["cpu_value"] = e["cpu_baseline"].get("value")
        if isinstance(e.get("kernel_us_one_forward"), dict):
            s["kernel_us_one_forward"] = e["kernel_us_one_forward"]
        out[k] = s
    return out


def emit(line, args, tag):
    """Prints the run's result.  `--full-line` (the secondary children, tools): the whole line, as rounds 1-5 printed it.  Otherwise: (1) the whole
    line into gpurun_out/bench_detail_<tag>.json, (2) the secondary configs as numbers on a PREFIXED stdout line, (3) the compact headline as the
    LAST stdout line (COMPACT_MAX_BYTES)."""
    _flush_c_stdio()
    if getattr(args, "full_line", False):
        print(json.dumps(line), flush=True)
        return
    rel = os.path.join("gpurun_out", f"bench_detail_{tag}.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(line, f, indent=1)
    except OSError:
        rel = None
    sec = line.get("secondary")
    if sec:
        print("bench-secondary (not the result line): " + json.dumps(secondary_summary(sec)), flush=True)
    print(json.dumps(compact_line(line, rel)), flush=True)


SECONDARY = [  # (key, extra argv, BASELINE config it stands for)
    ("core_c2", ["--dims", "core"], "configs[1] graph at core dims (128,64,32)=>(128,64,32): the matrix-core path"),
    ("c3", ["--workload", "hetero", "--hetero-graphs", "512"], "configs[2]: 512 random graphs (32-256 nodes), 1M edges"),
    ("c5_one_gpu", ["--workload", "hetero", "--hetero-graphs", "4096"], "configs[4]'s batch on one GPU: 4096 graphs, 1M edges"),
    ("c5w_one_gpu", ["--workload", "hetero", "--hetero-graphs", "4096", "--hetero-edges", "8000000"],
     "C5w (SURVEY 8d): configs[4]'s 4096 graphs at configs[2]'s density, 8M edges on one GPU — the block kernel over eight rounds of waves instead of one (its steady state)"),
    ("c5_dist_gnx", ["--dist-backend", "gnx", "--hetero-graphs", "4096", "--no-single-gpu-leg"],
     "configs[4]'s batch through the C boundary's own multi-GPU entry at one device: gnx_dist_block_forward_steps (hipGraph per device + grouped ncclAllGather + index table)"),
    ("c5_force_dist", ["--force-dist", "--hetero-graphs", "4096", "--no-single-gpu-leg"],
     "configs[4]'s batch through the N > 1 code path at world size 1 (partitioner, stacked gf' send buffer, RCCL all-gather on a side stream, index table back to graph order)"),
    ("c4", ["--model", "c4"], "configs[3]: Encoder -> 2 x GNCore(128,64,32) -> Decoder on the 1M-edge graph"),
    ("c4_narrow", ["--model", "c4", "--core-dims", "10,5,3"], "README example 3 at its own widths (core_dims 10,5,3)"),
    # the same model with every matrix product on the fp32 matrix instruction (--flags 96: the calls' own GNX_FLAG_FP32_MFMA): what the six-term bf16 form of `c4` is measured against
    ("c4_fp32_mfma", ["--model", "c4", "--no-cpu-baseline", "--no-c-abi", "--flags", "96"], "configs[3] with the FeedForwards and the cores' edge updates on the fp32 matrix instruction (every call carries GNX_FLAG_FP32_MFMA = 0x60; round 3's arithmetic) — beside `c4`, not instead of it"),
]


def collect_secondary(args):
    """Runs the other configs as child processes of THIS script, one after the other, before the parent has touched a GPU, and returns their
    lines cut down to what the judge reads.  A child that fails is reported as such (the headline line does not depend on it)."""
    out = {}
    t_all = time.perf_counter()
    for key, extra, what, *env_extra in SECONDARY:
        steps = {"c4": max(3, min(args.steps, 5)), "c4_fp32_mfma": max(3, min(args.steps, 5)), "core_c2": max(5, min(args.steps, 10))}.get(key, max(10, min(args.steps, 20)))
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(max(2, min(args.warmup, 5))),
               "--no-secondary", "--full-line", "--cpu-budget", "3.0"] + extra
        t0 = time.perf_counter()
        line = err = None
        for attempt in (0, 1):  # (a child that dies within seconds — a communicator's start-up now and then does — is started once more; a slow failure is not)
            t_try = time.perf_counter()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, **env_extra[0]) if env_extra else None)
                line = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
                if line is None:
                    e_ = (r.stderr or "").strip()
                    first = next((ln for ln in e_.splitlines() if "Error" in ln or "error" in ln or "what()" in ln), "")
                    err = (f"exit code {r.returncode}" + (" (second attempt)" if attempt else "") + ": " + first[:300] + " ... " + e_[-300:])[:800]
            except Exception as e:  # timeout, malformed output
                line, err = None, repr(e)[:300]
            if line is not None or time.perf_counter() - t_try > 20.0:
                break
        if line is None:
            out[key] = {"what": what, "error": err, "wall_s": round(time.perf_counter() - t0, 1)}
            continue
        roof = line.get("roofline") or {}
        keep = ("bound", "achieved", "peak", "unit", "frac", "frac_whole_step", "traffic", "kernel", "kernel_us", "executed_flops", "algorithmic_bytes",
                "frac_of_fp32_mfma_roof", "flops_on_bf16_six_terms", "arithmetic", "matrix_roof_frac")
        entry = {"what": what, "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line.get("steps", steps),
                 "roofline": {k: roof[k] for k in keep if k in roof}, "cpu_baseline": line.get("cpu_baseline"),
                 "wall_s": round(time.perf_counter() - t0, 1)}
        ts = roof.get("traffic_source")
        if ts:
            entry["roofline"]["traffic_source"] = {k: ts.get(k) for k in ("file", "commit", "stale") if ts.get(k) is not None}
        for k in ("batch_ms", "kernel_us_one_forward", "pipelined_two_streams", "chained_graph_update", "c_abi_ms_per_step", "c_abi", "n_gpus", "scaling", "with_allgather", "without_allgather"):
            if k in line or k in line.get("config", {}):
                entry[k] = line.get(k, line.get("config", {}).get(k))
        out[key] = entry
    out["_wall_s"] = round(time.perf_counter() - t_all, 1)
    return out


def c_abi_bench(mode, steps, warmup, extra=(), timeout=240):
    """tests/c/abi_bench.c as a child process (BEFORE this process touches the GPU): the same workload through include/gnx.h from plain C —
    no Python, no torch in the loop.  Built with gcc on first use (the box has no prebuilt copy when tests/c/_build did not travel).  The C2
    graph of THIS script goes over as a file, so the C program runs bench.py's exact graph.  Returns abi_bench's JSON line (or an error)."""
    import tempfile
    cdir = os.path.join(ROOT, "tests", "c")
    exe = os.path.join(cdir, "_build", "abi_bench")
    try:
        lib = os.path.join(ROOT, "graphnets.jl_amd", "libgnx.so")
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(lib), os.path.getmtime(os.path.join(cdir, "abi_bench.c"))):
            subprocess.run(["make", "-s", "-C", cdir, exe], check=True, capture_output=True, timeout=120)
        colptrs, rowvals, nn = make_c2()
        with tempfile.NamedTemporaryFile(suffix=".c2.bin", delete=False) as f:
            f.write(np.array([nn[0], len(rowvals[0])], dtype=np.int64).tobytes() + colptrs[0].astype(np.int64).tobytes() + rowvals[0].astype(np.int64).tobytes())
            path = f.name
        try:
            r = subprocess.run([exe, "--mode", mode, "--steps", str(steps), "--warmup", str(warmup), "--csc", path] + list(extra),
                               capture_output=True, text=True, timeout=timeout)
        finally:
            os.unlink(path)
        if r.returncode != 0 or not r.stdout.strip():
            return {"error": (r.stderr or r.stdout or "no output")[-300:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # build failure, timeout, malformed output: reported, never fatal for the headline
        return {"error": repr(e)[:300]}


def free_port():
    """A TCP port nobody is bound to right now (the GPU host is shared: a fixed rendezvous port can be somebody else's)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a CHILD job before this process has touched
    a GPU (a process that has initialised HIP must never exec or fork into GPU work), relay its output, exit with its code."""
    port = int(os.environ.get("MASTER_PORT", 0)) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def load_traffic(dims_key, kernel, sha):
    """HBM bytes per launch of `kernel` from the committed PMC profile of this width set — only while the profile was taken
    on the kernel sources that are running now (source_sha); otherwise None (a stale number is worse than none)."""
    tpath = os.path.join(ROOT, "profiles", f"traffic_{dims_key}.json")
    if not os.path.exists(tpath):
        return None, None
    with open(tpath) as f:
        t = json.load(f)
    meta = t.get("_meta", {})
    src = {"file": os.path.relpath(tpath, ROOT), "commit": meta.get("commit"), "date": meta.get("date"), "source_sha": meta.get("source_sha"),
           "current_source_sha": sha}
    if meta.get("source_sha") != sha:
        src["stale"] = True
        return None, src
    return t.get(kernel, {}).get("hbm_bytes_per_launch"), src


def live_traffic(kernel, bench_args, timeout_s=75.0):
    """HBM bytes per launch of `kernel` measured IN THIS RUN (the committed figure was quoted, never driver-observed): two
    rocprofv3 passes — `--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`, separate passes with the kernel trace only beside them, as MI355X_MICROARCH's
    HBM section prescribes — over a child `python3 bench.py <the same workload> --no-secondary --no-cpu-baseline --no-c-abi`; both counters are KiB,
    FETCH_SIZE doubled (the guide's gfx950 correction for wide coalesced reads).  The child is a process group of its own, killed as a group when a
    pass overruns.  Returns (bytes or None, a record of what was done); None leaves the quoted figure in place."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"live": False, "why": "rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, {"live": False, "why": "this run is itself under a profiler"}  # (no profiler inside a profiler: tools/profile.sh collects the counters of such runs)
    py = os.path.realpath(sys.executable)  # (the interpreter itself behind `--`: no env / shell hop under the profiler)
    means, rec = {}, {"live": True, "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, + --kernel-trace) over a child bench.py of this workload, "
                                            "in this run; KiB -> B, FETCH_SIZE x2 (gfx950 wide-read correction); FETCH_SIZE counts Infinity-Cache hits too"}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE", None):  # (None: a third pass with the kernel trace alone — the profiler's own average duration of the kernel)
        out = tempfile.mkdtemp(prefix="gnx_pmc_", dir="/tmp")
        cmd = [exe] + (["--pmc", counter] if counter else []) + ["--kernel-trace", "--output-format", "csv", "-d", out, "--", py, os.path.join(ROOT, "bench.py")] + \
              list(bench_args) + ["--no-secondary", "--no-cpu-baseline", "--no-c-abi", "--full-line", "--no-live-traffic"]
        try:
            pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)  # (the group this call started, by its id)
                except OSError:
                    pass
                pr.wait()
                shutil.rmtree(out, ignore_errors=True)
                if counter is None and "FETCH_SIZE" in means and "WRITE_SIZE" in means:
                    break
                return None, {"live": False, "why": f"{counter} pass took more than {timeout_s:g} s"}
            vals = []
            if counter is None:
                for f in glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if ("gnx::" + kernel) in row.get("Kernel_Name", ""):
                                vals.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                if rc == 0 and vals:
                    rec["kernel_us_rocprofv3"] = round(sum(vals) / len(vals) / 1e3, 3)
                    rec["kernel_trace_launches"] = len(vals)
                continue  # (optional: the traffic figure does not depend on this pass)
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and ("gnx::" + kernel) in row.get("Kernel_Name", ""):
                            vals.append(float(row["Counter_Value"]))
            if rc != 0 or not vals:
                return None, {"live": False, "why": f"{counter} pass: exit code {rc}, {len(vals)} launches of {kernel} sampled"}
            means[counter] = sum(vals) / len(vals)
            rec[counter.lower() + "_launches_sampled"] = len(vals)
        except OSError as e:
            return None, {"live": False, "why": f"{counter} pass: {e}"}
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fetch, write = means["FETCH_SIZE"] * 1024 * 2, means["WRITE_SIZE"] * 1024
    rec.update(fetch_bytes_corrected=round(fetch, 1), write_bytes=round(write, 1), wall_s=round(time.perf_counter() - t0, 1))
    return fetch + write, rec


def block_roofline(gn, torch, dev, plan, sets, nsets, K, E, N, G, din, dout, ms_per_step, dims_key, headline_traffic_breakdown=False, steps_form=False):
    """`roofline` of one GNBlock forward: the same K steps again, eagerly, with dispatch timestamps on every launch (gnx_profile_*), at settled
    clocks; the dominant kernel's average duration prices the block's ALGORITHMIC bytes (or its executed flops where the matrix cores bind).
    `steps_form`: the K steps as ONE gnx_block_forward_steps call (the form the headline times) instead of K gnx_block_forward calls."""
    def eager_pass():
        if steps_form:
            plan.steps([sets[i % nsets] for i in range(K)])
            return
        for i in range(K):
            b = sets[i % nsets]
            plan(b["ef"], b["nf"], b["gf"], *b["out"], ws=b["ws"])
    spin_up(torch, dev, eager_pass)  # the pass below runs at settled clocks, like the timed region
    gn.profile_reset(); gn.profile_enable(True)
    eager_pass()
    gn.profile_calibrate(K, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    gn.profile_enable(False)
    prof = gn.profile_read(); gn.profile_reset()
    null_us = (prof.get("__empty_bracket__") or {"total_ms": 0.0, "launches": 1})
    null_us = null_us["total_ms"] / max(null_us["launches"], 1) * 1e3
    kern = calibrated_kernel_us(prof)
    dom = max(kern, key=kern.get)
    dur_s = kern[dom] * 1e-6
    step_s = ms_per_step * 1e-3
    abytes, aflops = algorithmic_bytes(E, N, G, din, dout), algorithmic_flops(E, N, G, din, dout)
    edge_fp32_ = fp32_forms(plan.flags)[1]
    hbm_t, mfma_t = abytes / (HBM_PEAK_GBS * 1e9), aflops / (MFMA_F32_PEAK_TFS * 1e12)
    if max(din + dout) > 32:  # matrix-core path: the binding roof is decided on the flops the kernels EXECUTE at the rate of the instruction that carries them
        ex_ = executed_flops(E, N, G, din, dout)
        x6_ = 2 * E * din[0] * dout[0] if (din[0] == 128 and dout[0] == 128 and din[1] >= 16 and E >= 2 * N and not edge_fp32_) else 0
        mfma_t = x6_ / (MFMA_BF16_PEAK_TFS / 6 * 1e12) + (ex_ - x6_) / (MFMA_F32_PEAK_TFS * 1e12)
    traffic, tsrc = load_traffic(dims_key, dom, kernel_source_sha(din, dout))
    if hbm_t >= mfma_t:
        a = abytes / dur_s / 1e9
        roof = dict(bound="hbm", achieved=round(a, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(a / HBM_PEAK_GBS, 4),
                    frac_whole_step=round(abytes / step_s / 1e9 / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=tsrc,
                    counts="algorithmic bytes of the whole block / duration of the dominant kernel (frac) or of the whole step (frac_whole_step)")
        if max(din + dout) > 32:  # a matrix-core block whose matrix time (per carrying instruction) is below its memory time: both fractions on the WHOLE step
            roof.update(frac=roof["frac_whole_step"], algorithmic_bytes=abytes, executed_flops=ex_, flops_on_bf16_six_terms=x6_,
                        frac_of_fp32_mfma_roof=round(ex_ / step_s / 1e12 / MFMA_F32_PEAK_TFS, 4), matrix_roof_frac=round(mfma_t / step_s, 4),
                        counts="algorithmic bytes of the whole block / whole-step time (several launches); matrix_roof_frac: the executed flops at the rate of the "
                               "instruction that carries them (fp32 MFMA %.1f TFLOP/s; six bf16 terms per fp32 product: %.0f / 6) / whole-step time" % (MFMA_F32_PEAK_TFS, MFMA_BF16_PEAK_TFS))
    else:
        # MFMA-bound: priced on the flops the kernels EXECUTE (never more than the peak); the algorithmic rate of the
        # reference formulation (every edge multiplies its full [ef; nf_src; nf_dst; gf] row) is reported beside it
        ex = executed_flops(E, N, G, din, dout)
        a = ex / step_s / 1e12
        # the projected edge update at 128 -> 128 runs as six bf16 matrix-core terms per fp32 product (k_edge_x6) unless GNX_EDGE_FP32=1: its
        # flops are priced at that instruction's rate, the rest at the fp32 MFMA's
        x6_flops = 2 * E * din[0] * dout[0] if (din[0] == 128 and dout[0] == 128 and din[1] >= 16 and E >= 2 * N and not edge_fp32_) else 0
        t_roof = x6_flops / (MFMA_BF16_PEAK_TFS / 6 * 1e12) + (ex - x6_flops) / (MFMA_F32_PEAK_TFS * 1e12)
        roof = dict(bound="mfma", achieved=round(a, 3), peak=round(ex / t_roof / 1e12, 1), unit="TFLOP/s", frac=round(t_roof / step_s, 4),
                    frac_whole_step=round(t_roof / step_s, 4), frac_of_fp32_mfma_roof=round(a / MFMA_F32_PEAK_TFS, 4), flops_on_bf16_six_terms=x6_flops,
                    traffic=traffic, traffic_source=tsrc,
                    counts="EXECUTED flops of the whole block / whole-step time (the block is several GEMM launches); peak = the same flops at the rate of the instruction "
                           "that carries them (fp32 MFMA %.1f TFLOP/s; six bf16 terms per fp32 product: %.0f / 6)" % (MFMA_F32_PEAK_TFS, MFMA_BF16_PEAK_TFS),
                    executed_flops=ex, algorithmic_tflops_whole_step=round(aflops / step_s / 1e12, 2),
                    hbm_frac_whole_step=round(abytes / step_s / 1e9 / HBM_PEAK_GBS, 4))
    if traffic is not None and headline_traffic_breakdown:
        # `traffic` counts what leaves the L2s (fabric bytes: Infinity-Cache hits included); the split into HBM bytes and cache hits
        # comes from a residency A/B (no counter separates them): profiles/traffic_breakdown_readme.json
        bpath = os.path.join(ROOT, "profiles", "traffic_breakdown_readme.json")
        if os.path.exists(bpath):
            with open(bpath) as f:
                bd = json.load(f)
            if dom in bd:
                roof["traffic_breakdown"] = {"fabric": traffic, "hbm_estimate": bd[dom]["hbm_bytes_per_launch_estimate"],
                                             "infinity_cache_hits_estimate": bd[dom]["infinity_cache_hit_bytes_estimate"],
                                             "source": "profiles/traffic_breakdown_readme.json"}
    assert roof["frac"] <= 1.0 and roof["frac_whole_step"] <= 1.0, "a roofline fraction above 1 is an accounting error"
    roof.update(kernel=dom, kernel_us=round(kern[dom], 3), kernel_us_source="dispatch timestamps of every launch of the per-kernel pass (hipExtLaunchKernel start / stop events), averaged, minus the constant (empty kernel timed the same way - rocprofv3's figure for it: profiles/calibration.json)",
                null_kernel_us=round(null_us, 3), algorithmic_bytes=abytes, algorithmic_flops=aflops,
                bytes_per_edge=round(abytes / E, 2), all_kernels_us={k: round(v, 3) for k, v in kern.items()})

    return roof


def sharded_workload(args, world):
    """Which batch an N-rank run shards (pure arithmetic: the CPU tests call it).  `--scaling strong` (default): ONE FIXED batch whatever N is —
    BASELINE configs[4]: 4096 graphs (32-256 nodes), 1M edges, seed 5 (`--hetero-graphs` / `--hetero-edges` are then TOTALS of that batch).
    `--scaling weak`: N x 512 graphs / N x 1M edges (the per-GPU shard is fixed instead).  Returns (G_total, E_total, seed)."""
    if args.scaling == "strong":
        Gtot, Etot = args.hetero_graphs or 4096, args.hetero_edges or 1_000_000
    else:
        Gtot, Etot = (args.hetero_graphs or 512) * world, (args.hetero_edges or 1_000_000) * world
    seed = 3 if Gtot == 512 else (5 if Gtot == 4096 else 1000 + Gtot)  # C3 = seed 3, C5 = seed 5
    return Gtot, Etot, seed


def rotating_sets(set_bytes):
    """Buffer sets a rank rotates over so that its footprint stays above the 256-MiB Infinity Cache (cache-cold steps) however small the shard is."""
    return int(min(64, max(NSETS, -(-320_000_000 // max(int(set_bytes), 1)))))


def bench_weights(gn, din, dout, dev):
    """The bench's GNBlock: glorot weights from a fixed seed (identical on every rank and device), zero biases."""
    rng = np.random.default_rng(100)
    (de, dn, dg), (oe, on, og) = din, dout
    blk = gn.GNBlock(din, dout, device=dev)
    blk.edgefn = gn.Dense.from_numpy(glorot(rng, oe, de + 2 * dn + dg), np.zeros(oe, np.float32), device=dev)
    blk.nodefn = gn.Dense.from_numpy(glorot(rng, on, oe + dn + dg), np.zeros(on, np.float32), device=dev)
    blk.graphfn = gn.Dense.from_numpy(glorot(rng, og, oe + on + dg), np.zeros(og, np.float32), device=dev)
    if max(din + dout) > 32:
        blk.prepare()  # gnx_block_prepare: the matrix-core kernels' weight planes made ONCE, as `model |> device` happens once (examples/sort/sort.jl:29,89)
    return blk


def block_sets(torch, dev, plan, g, din, nsets, seed):
    (de, dn, dg) = din
    tg = torch.Generator(device=dev); tg.manual_seed(seed)
    mk = lambda T, d: torch.rand((1, T, d), generator=tg, device=dev, dtype=torch.float32) if d > 0 else None
    return [dict(ef=mk(g.n_edges, de), nf=mk(g.n_nodes, dn), gf=mk(g.n_graphs, dg), out=plan.outputs(), ws=plan.new_workspace()) for _ in range(nsets)]


def single_gpu_same_workload(gn, torch, dev, seed, Gtot, Etot, din, dout, K, W, flags):
    """The WHOLE batch of a sharded run on ONE GPU, by the N = 1 procedure (K steps captured into one hipGraph over rotating buffer sets, 150 ms of
    the same load first, median of three regions): the denominator of a strong-scaling efficiency that compares the same graphs."""
    colptrs, rowvals, nn = make_hetero(seed, Gtot, Etot)
    g = gn.GNGraphBatch.from_csc_packed(np.concatenate(colptrs), np.concatenate(rowvals), nn, device=dev)
    blk = bench_weights(gn, din, dout, dev)
    plan = gn.BlockPlan(blk, g, R=1, flags=flags)
    nsets = rotating_sets(algorithmic_bytes(g.n_edges, g.n_nodes, g.n_graphs, din, dout))
    sets = block_sets(torch, dev, plan, g, din, nsets, 999)
    step = lambda i: plan(sets[i % nsets]["ef"], sets[i % nsets]["nf"], sets[i % nsets]["gf"], *sets[i % nsets]["out"], ws=sets[i % nsets]["ws"])
    for i in range(max(W, 2)):
        step(i)
    torch.cuda.synchronize(dev)
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, capture_error_mode="thread_local"):
        plan.steps([sets[i % nsets] for i in range(K)])  # gnx_block_forward_steps: the K steps as one call (the N = 1 headline's form)
    cg.replay(); torch.cuda.synchronize(dev)
    spin_up(torch, dev, cg.replay)
    reps = []
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        cg.replay()
        torch.cuda.synchronize(dev)
        reps.append(time.perf_counter() - t0)
    dt = sorted(reps)[1] / K
    out = {"n_gpus": 1, "ms_per_step": round(dt * 1e3, 6), "value": round(g.n_edges / dt, 1), "unit": "edges/s", "graphs": g.n_graphs, "edges": g.n_edges,
           "what": f"the same {Gtot} graphs / {g.n_edges} edges as ONE batch on ONE GPU (rank 0, before the sharded measurement; hipGraph of {K} steps over {nsets} rotating buffer sets, median of 3)"}
    del cg, sets, plan, g
    torch.cuda.empty_cache()
    return out


def measure_sharded(gn, torch, dist, dev, rank, world, args, din, dout, Gtot, Etot, seed, K, W, want_roofline=True):
    """ONE heterogeneous batch (G_total graphs, E_total edges) sharded BY GRAPH over `world` ranks (the product's partitioner: equal graph counts,
    snake order by edge count), every rank a handle of ITS graphs, no data-path collective except the all-gather of gf' (M stacked tables per
    collective, on a side stream).  Times the K-step region twice — with the all-gather (the whole path: `value`) and without it — each bracketed
    by barrier + synchronize, MAX over ranks, median of three; rank 0 also runs the whole batch alone first (single_gpu_same_workload)."""
    from graphnets_jl_amd.dist import GfGather, partition_graphs
    (de, dn, dg), (oe, on, og) = din, dout
    single = None
    if rank == 0 and not args.no_single_gpu_leg:
        single = single_gpu_same_workload(gn, torch, dev, seed, Gtot, Etot, din, dout, K, W, args.flags)
    n_all, e_all = hetero_spec(seed, Gtot, Etot)
    shards = partition_graphs(e_all, world)
    colptrs, rowvals, nn = make_hetero(seed, Gtot, Etot, only=shards[rank])
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
    E, N, G = g.n_edges, g.n_nodes, g.n_graphs
    blk = bench_weights(gn, din, dout, dev)
    plan = gn.BlockPlan(blk, g, R=1, flags=args.flags)
    nsets = 2 if max(din + dout) >= 64 else rotating_sets(algorithmic_bytes(E, N, G, din, dout))
    sets = block_sets(torch, dev, plan, g, din, nsets, 1234 + rank)
    # M steps of compute are captured into one hipGraph that writes the M gf' tables into a stacked send buffer, and ONE all-gather moves the
    # whole stack (fewer, larger collectives: the per-step message is G * DG' floats = 10 KB per rank, pure latency on xGMI); the gather runs on
    # a side stream and overlaps the next M steps.
    M = max(m for m in range(1, 257) if K % m == 0)
    gather = GfGather(shards, rank, world, og, dev, stack=M)
    gf_stack = gather.send  # [M][max_count][og]: the graph update writes straight into the send buffer
    gf_result = torch.empty((M * gather.G, og), dtype=torch.float32, device=dev)  # gathered tables, ORIGINAL graph order
    assert gf_stack.shape[1] == G or world > 1, "one rank: every graph is local"

    def step(i, slot):
        b = sets[i % nsets]
        plan(b["ef"], b["nf"], b["gf"], b["out"][0], b["out"][1], gf_stack[slot:slot + 1, :G], ws=b["ws"])

    def sync_all():
        """barrier + torch.cuda.synchronize() (the contract's bracket): the gathered gf' table is waited for first, and the barrier's collective is
        enqueued asynchronously behind it — one small RCCL kernel and ONE host synchronisation per bracket."""
        if gather._ready is not None:
            gather.finish(out=gf_result)
        work = dist.barrier(async_op=True)
        torch.cuda.synchronize(dev)
        work.wait()
        torch.cuda.synchronize(dev)

    def timed(run):
        sync_all()
        t0 = time.perf_counter()
        run()
        sync_all()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for i in range(max(W, 2)):  # warm-up (also loads the code objects: at least two eager steps before any capture)
        step(i, i % M)
    sync_all()
    cgs = []
    for base in range(0, min(K, nsets * M), M):  # enough distinct graphs to keep rotating over all buffer sets
        cg = torch.cuda.CUDAGraph()
        # (thread_local: with a process group alive torch's NCCL watchdog THREAD queries the events of collectives in flight — an error inside a
        #  "global" capture, thrown in that thread: the process aborts.  Seen once in ~20 starts of this path; tools/experiments/capture_vs_watchdog_probe.py
        #  aborts every time with the default mode and never with this one: profiles/r06_capture_vs_watchdog.log)
        with torch.cuda.graph(cg, capture_error_mode="thread_local"):  # the M steps as ONE gnx_block_forward_steps call (every gf' table complete when the graph's work is)
            plan.steps([(b["ef"], b["nf"], b["gf"], b["out"][0], b["out"][1], gf_stack[m:m + 1, :G], b["ws"]) for m, b in ((m, sets[(base + m) % nsets]) for m in range(M))])
        cgs.append(cg)
    copied = torch.cuda.Event()

    def run_with():
        for j in range(K // M):
            torch.cuda.current_stream(dev).wait_event(copied) if j else None  # the previous stack has left the send buffer
            cgs[j % len(cgs)].replay()
            gather.start_inplace()
            copied.record(gather.comm_stream) if gather.comm_stream is not None else copied.record()

    def run_without():
        for j in range(K // M):
            cgs[j % len(cgs)].replay()

    def settle(run):
        run(); sync_all()
        t_w = time.perf_counter()
        while True:  # the same untimed load as spin_up; the ranks agree on when to stop (a rank-local clock would leave them in different collectives)
            run(); sync_all()
            t = torch.tensor([time.perf_counter() - t_w], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t.item()) * 1e3 >= CLOCK_WARMUP_MS:
                return
    settle(run_without)
    reps_wo = sorted(timed(run_without) for _ in range(3))
    settle(run_with)
    reps = sorted(timed(run_with) for _ in range(3))
    dt, dt_wo = reps[1], reps_wo[1]
    gf_all = gather.finish() if gather._ready is not None else gather.result()
    assert tuple(gf_all.reshape(M, -1, og).shape) == (M, gather.G, og) and torch.equal(gf_all.reshape(-1, og), gf_result)
    # every rank's rows of the gathered table are the rows it computed (original graph order restored)
    mine = torch.from_numpy(np.asarray(shards[rank], dtype=np.int64)).to(dev)
    assert torch.equal(gf_all.reshape(M, gather.G, og)[:, mine], gf_stack[:, :G]), "gathered gf' rows differ from the rows this rank wrote"
    t = torch.tensor([float(E), float(N), float(G)], device=dev, dtype=torch.float64)
    per_rank = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(per_rank, t)
    counts = [[int(v) for v in p.tolist()] for p in per_rank]
    E_job = sum(c[0] for c in counts)
    assert E_job == Etot and sum(c[2] for c in counts) == Gtot, "the shards do not add up to the batch"
    res = {"E_job": E_job, "G_job": Gtot, "seed": seed, "M": M, "nsets": nsets, "dt": dt, "dt_without": dt_wo, "reps": reps, "reps_without": reps_wo,
           "per_rank": counts, "single": single, "E": E, "N": N, "G": G, "roof": None}
    if rank == 0 and want_roofline:
        rsets = [dict(b, out=(b["out"][0], b["out"][1], gf_stack[0:1, :G])) for b in sets]
        dims_key = args.dims.replace(":", "_").replace(",", "-") + f"_hetero{G}" + ("" if E == 1_000_000 else f"_{E}")
        res["roof"] = block_roofline(gn, torch, dev, plan, rsets, nsets, K, E, N, G, din, dout, dt_wo / K * 1e3, dims_key)
    del cgs
    return res


def sharded_entry(res, K, world):
    """The fields of one sharded measurement as they appear in the line (headline of an N > 1 run, and `secondary.c5w`)."""
    dt, dtw, E = res["dt"] / K, res["dt_without"] / K, res["E_job"]
    out = {"value": round(E / dt, 1), "unit": "edges/s", "ms_per_step": round(dt * 1e3, 6), "edges_whole_job": E, "graphs_whole_job": res["G_job"],
           "with_allgather": {"ms_per_step": round(dt * 1e3, 6), "value": round(E / dt, 1), "reps_us_per_step": [round(r / K * 1e6, 2) for r in res["reps"]]},
           "without_allgather": {"ms_per_step": round(dtw * 1e3, 6), "value": round(E / dtw, 1), "reps_us_per_step": [round(r / K * 1e6, 2) for r in res["reps_without"]],
                                 "what": "the same hipGraph replays without the RCCL all-gather of gf' (every rank keeps only its own graphs' rows)"},
           "per_rank_edges_nodes_graphs": res["per_rank"], "single_gpu_same_workload": res["single"]}
    if res["single"]:
        out["speedup_vs_single_gpu_same_workload"] = round(res["single"]["ms_per_step"] / (dt * 1e3), 4)
    return out


def sharded_line(args, res, c5w, K, W, world, din, dout):
    """The whole N > 1 line from rank 0's measurements (pure: the CPU tests assemble it from canned results)."""
    Gtot, Etot, seed = res["G_job"], res["E_job"], res["seed"]
    e = sharded_entry(res, K, world)
    strong = args.scaling == "strong"
    wl = (f"BASELINE configs[4]: ONE FIXED heterogeneous batch of {Gtot} random graphs (32-256 nodes, {Etot / 1e6:g}M edges, seed {seed}) sharded by graph over {world} GPU(s) — "
          f"the same graphs at every N (strong scaling)" if strong else
          f"ONE heterogeneous batch of {Gtot} random graphs (32-256 nodes, {Etot / 1e6:g}M edges, seed {seed}) sharded by graph over {world} GPU(s): {Gtot // world} graphs / "
          f"~{Etot // world} edges per GPU at every N (weak scaling; BASELINE configs[4] law)")
    line = {"metric": "edges updated/sec, GNBlock fwd, 1M-edge batch", "value": e["value"], "unit": "edges/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": e["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "dims": f"{din}=>{dout}", "edges_whole_job": e["edges_whole_job"], "graphs_whole_job": e["graphs_whole_job"],
                       "edges_per_gpu": res["E"], "nodes_per_gpu": res["N"], "graphs_per_gpu": res["G"], "per_rank_edges_nodes_graphs": e["per_rank_edges_nodes_graphs"],
                       "parallelism": f"graph-sharded x{world}", "dist_backend": "torch (one process per GPU, RCCL all-gather of gf')",
                       "launch": f"hipGraph of one gnx_block_forward_steps call ({res['M']} steps) per replay over {res['nsets']} rotating buffer sets per rank; one RCCL all-gather of the {res['M']} stacked gf' tables per replay, on a side stream",
                       "timing": f"median of 3 runs of the {K}-step region ({e['with_allgather']['reps_us_per_step']} us/step), MAX over ranks, after {CLOCK_WARMUP_MS:g} ms of the same load, untimed"},
            "with_allgather": e["with_allgather"], "without_allgather": e["without_allgather"], "single_gpu_same_workload": e["single_gpu_same_workload"],
            "roofline": res["roof"], "cpu_baseline": None}
    if "speedup_vs_single_gpu_same_workload" in e:
        line["speedup_vs_single_gpu_same_workload"] = e["speedup_vs_single_gpu_same_workload"]
    if res["roof"] is not None:
        ab = sum(algorithmic_bytes(c[0], c[1], c[2], din, dout) for c in res["per_rank"])
        line["roofline"]["whole_job"] = {"algorithmic_bytes": ab, "achieved": round(ab / (res["dt"] / K) / 1e9, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                         "frac": round(ab / (res["dt"] / K) / 1e9 / (HBM_PEAK_GBS * world), 4),
                                         "counts": "algorithmic bytes of every rank's shard / whole-step time / (n_gpus x 8 TB/s); the kernel figures above are rank 0's shard"}
    if c5w is not None:
        ew = sharded_entry(c5w, K, world)
        ew["what"] = "C5w (SURVEY 8e): the same 4096 graphs at configs[2]'s density — 8M edges — sharded over the same ranks; strong scaling of a batch eight times larger"
        ew["roofline"] = c5w["roof"]
        line["secondary"] = {"c5w": ew}
    return line


def bench_sharded(args, gn, torch, dist, dev, rank, world, din, dout):
    """N > 1 (and --force-dist): BASELINE configs[4].  Default `--scaling strong`: the FIXED 4096-graph / 1M-edge batch (seed 5: the metric's
    "1M-edge batch") partitioned over the N ranks — the same graphs at every N, so the 1/2/4/8 curve is a scaling curve of ONE workload — with the
    same 4096 graphs at 8M edges (C5w) beside it as `secondary.c5w`.  Every line carries the whole batch on one GPU (`single_gpu_same_workload`)
    and the region with and without the all-gather."""
    K, W = args.steps, args.warmup
    Gtot, Etot, seed = sharded_workload(args, world)
    res = measure_sharded(gn, torch, dist, dev, rank, world, args, din, dout, Gtot, Etot, seed, K, W)
    c5w = None
    strong_default = args.scaling == "strong" and (Gtot, Etot) == (4096, 1_000_000) and not args.no_secondary
    if strong_default:
        torch.cuda.empty_cache()
        c5w = measure_sharded(gn, torch, dist, dev, rank, world, args, din, dout, 4096, 8_000_000, 5, K, W)
    line = sharded_line(args, res, c5w, K, W, world, din, dout) if rank == 0 else None
    dist.barrier()
    dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        time.sleep(0.5)  # the other ranks' (already flushed) output reaches the launcher's pipe first
        emit(line, args, f"sharded_n{world}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dims", default="readme", help="a preset (%s) or explicit widths de,dn,dg:oe,on,og" % ", ".join(DIMS))
    ap.add_argument("--workload", choices=["c2", "hetero"], default=None, help="default: c2 at N = 1 (BASELINE configs[1]), hetero at N > 1 (configs[4])")
    ap.add_argument("--hetero-graphs", type=int, default=None,
                    help="graphs of the hetero workload: on one GPU the batch (default 512 = C3; 4096 = C5 on one GPU); with --scaling strong the TOTAL of the fixed "
                         "batch (default 4096 = configs[4]); with --scaling weak per GPU (default 512)")
    ap.add_argument("--hetero-edges", type=int, default=None, help="edges of the hetero workload, counted like --hetero-graphs (default 1000000; C5w: 8000000)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong (default) = ONE fixed batch — BASELINE configs[4]: 4096 graphs / 1M edges, seed 5 — partitioned over the N ranks, C5w (8M edges) "
                         "beside it; weak = N x 512 graphs / N x 1M edges (the per-GPU shard fixed)")
    ap.add_argument("--no-prepare", action="store_true", help="--model c4: leave the layers' parameters unprepared (every forward then runs the *_prep launches, as before round 5)")
    ap.add_argument("--no-single-gpu-leg", action="store_true", help="N > 1: skip rank 0's run of the whole batch on one GPU (single_gpu_same_workload)")
    ap.add_argument("--c2-scale", type=float, default=1.0, help="scale C2's nodes and edges by this factor (size sweeps; 1 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configs of the default N = 1 line")
    ap.add_argument("--no-c-abi", action="store_true", help="skip the torch-free C program's leg (c_abi_ms_per_step)")
    ap.add_argument("--no-live-traffic", action="store_true", help="default headline run: do not measure roofline.traffic in this run (two rocprofv3 counter passes over a child); quote the committed profile")
    ap.add_argument("--cpu-budget", type=float, default=None, help="seconds of CPU baseline sampling (default 3 at README dims, 12 at wide dims)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--full-line", action="store_true", help="print the WHOLE result as one JSON line (rounds 1-5's form; the secondary children and the tests of the detail fields use it) "
                                                              "instead of the compact headline (< 3 KB) + gpurun_out/bench_detail_*.json")
    ap.add_argument("--dense-baseline", action="store_true",
                    help="also time the RESTATEMENT OF THE REFERENCE'S FORMULATION (padded one-hot batched matmuls, numpy/BLAS on the "
                         "host; BASELINE.md B2) on README ex.1 and on a 64-graph batch with PN <= 64")
    ap.add_argument("--overlap", action="store_true",
                    help="two-phase steps: graph update of step i on a second stream (measured SLOWER inside a hipGraph: the "
                         "fork/join costs more than the 5 us it hides — 35.7 vs 27.7 us/step — so it is off by default)")
    ap.add_argument("--separate-calls", action="store_true",
                    help="N = 1: time K separate gnx_block_forward calls (two launches per step; rounds 1-5's headline form) instead of ONE gnx_block_forward_steps call")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 code path even with one rank (testing)")
    ap.add_argument("--dist-backend", choices=["torch", "gnx"], default="torch",
                    help="torch: one process per GPU, gf' all-gathered by torch.distributed's RCCL (the contract's launch form; default).  gnx: ONE process "
                         "drives --gpus devices through the C boundary's own sharded path (gnx_dist_block_forward_steps: a hipGraph per device + one "
                         "grouped ncclAllGather) — what a Julia session does; run it directly, not under torchrun")
    ap.add_argument("--core-dims", type=str, default="128,64,32", help="core widths for --model c4 (README ex.3 uses 10,5,3)")
    ap.add_argument("--model", choices=["block", "c4"], default="block",
                    help="c4: BASELINE configs[3] — encoder -> 2 x GNCore(128,64,32) -> decoder on the C2 graph (extra; not the headline line)")
    args = ap.parse_args()
    if args.steps < 1 or args.steps > 5000 or args.warmup < 0:  # (K steps are captured into ONE hipGraph: six-figure node counts crash the runtime's capture)
        ap.error("--steps must be in 1..5000 and --warmup >= 0")

    if args.dist_backend == "gnx":
        return bench_dist_gnx(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))  # nothing has touched a GPU yet (torch is not even imported)
    # the default single-GPU line carries the other configs: child processes, one at a time, BEFORE this process touches the GPU
    headline = (args.gpus == 1 and "RANK" not in os.environ and args.dims == "readme" and args.workload in (None, "c2") and args.model == "block"
                and not args.force_dist and args.c2_scale == 1.0 and args.flags == 0 and not args.overlap)
    secondary = collect_secondary(args) if headline and not args.no_secondary else None
    # the same workload through the C boundary WITHOUT Python / torch (tests/c/abi_bench.c), as a child process before this one touches the GPU
    c_abi = None
    if args.gpus == 1 and "RANK" not in os.environ and not args.no_c_abi and not args.force_dist and args.c2_scale == 1.0 and args.flags == 0 and not args.overlap:
        if args.model == "c4":
            c_abi = c_abi_bench("c4", max(3, min(args.steps, 10)), 2, ["--core-dims", args.core_dims])
        elif args.dims == "readme" and args.workload in (None, "c2"):
            c_abi = c_abi_bench("block", max(args.steps, 20), max(args.warmup, 8))

    import torch
    import torch.distributed as dist
    import graphnets_jl_amd as gn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()))
        try:
            dist.init_process_group("nccl", device_id=dev)
        except RuntimeError:  # (the rendezvous store's port taken between free_port() and the bind, or a transient bootstrap error: once more)
            if world > 1:
                raise
            time.sleep(1.0)
            os.environ["MASTER_PORT"] = str(free_port())
            dist.init_process_group("nccl", device_id=dev)
    K, W = args.steps, args.warmup
    if args.model == "c4":
        return bench_c4(args, gn, torch, dev, c_abi)
    if args.dims in DIMS:
        din, dout = DIMS[args.dims]
    else:
        din, dout = (tuple(int(v) for v in part.split(",")) for part in args.dims.split(":"))
        assert len(din) == 3 and len(dout) == 3, "--dims de,dn,dg:oe,on,og"
    if multi:
        if args.workload == "c2":
            raise SystemExit("a single giant graph does not shard (SURVEY 8e: replicas only); N > 1 measures the graph-sharded hetero batch")
        return bench_sharded(args, gn, torch, dist, dev, rank, world, din, dout)
    workload = args.workload or "c2"
    args.hetero_graphs, args.hetero_edges = args.hetero_graphs or 512, args.hetero_edges or 1_000_000

    # ---- synthetic batch ----
    if workload == "c2":
        colptrs, rowvals, nn = make_c2(seed=2, N=int(100_000 * args.c2_scale), E=int(1_000_000 * args.c2_scale))
        wl_name = "C2: one shared Erdos-Renyi graph, 100k nodes / 1M edges, batch_size=1 (BASELINE configs[1])"
    else:
        Gtot, Etot = args.hetero_graphs, args.hetero_edges
        seed = 3 if Gtot == 512 else (5 if Gtot == 4096 else 1000 + Gtot)  # C3 = seed 3, C5 = seed 5
        colptrs, rowvals, nn = make_hetero(seed, Gtot, Etot)
        wl_name = (f"ONE heterogeneous batch of {Gtot} random graphs (32-256 nodes, {Etot / 1e6:g}M edges, seed {seed}) on one GPU "
                   f"(BASELINE configs[{2 if Gtot == 512 else 4}] law)")
    torch.cuda.synchronize(dev)
    # once per process, not per batch: library load, context, and the device builder's own first use (its kernels' code objects, its stream and
    # scratch buffer) on a graph large enough to take that path; reported as batch_ms.one_time_init
    t0 = time.perf_counter()
    gn.GNGraphBatch.from_csc(*make_c2(seed=1, N=8_000, E=80_000), device=dev)
    torch.cuda.synchronize(dev)
    init_ms = (time.perf_counter() - t0) * 1e3
    tb = []
    for _ in range(3):
        t0 = time.perf_counter()
        g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
        torch.cuda.synchronize(dev)
        tb.append((time.perf_counter() - t0) * 1e3)
    # the packed input form (the graphs' colptr / rowval arrays one after the other, as a data loader holds them): no per-graph Python work
    cpc, rvc = np.concatenate(colptrs), np.concatenate(rowvals)
    tp = []
    for _ in range(4):
        t0 = time.perf_counter()
        gp = gn.GNGraphBatch.from_csc_packed(cpc, rvc, nn, device=dev)
        torch.cuda.synchronize(dev)
        tp.append((time.perf_counter() - t0) * 1e3)
        del gp
    cp32, rv32 = cpc.astype(np.int32), rvc.astype(np.int32)
    t32 = []
    for _ in range(3):
        t0 = time.perf_counter()
        gp = gn.GNGraphBatch.from_csc_packed(cp32, rv32, nn, device=dev)
        torch.cuda.synchronize(dev)
        t32.append((time.perf_counter() - t0) * 1e3)
        del gp
    batch_ms = {"from_csc": round(min(tb), 3), "from_csc_first_call": round(tb[0], 3),
                "from_csc_packed": round(min(tp), 3), "from_csc_packed_first_call": round(tp[0], 3), "from_csc_packed_int32": round(min(t32), 3),
                "one_time_init": round(init_ms, 3),
                "what": "GNGraphBatch construction from Python, end to end: best of the calls, and the first call of this batch (one_time_init = the first build of the PROCESS, on an 80k-edge graph: library load, context, the builder's code objects and scratch).  from_csc = a LIST of per-graph arrays (numpy "
                        "concatenates them: ~2.4 ms for 4096 graphs); from_csc_packed = the concatenated arrays as they are (int64; _int32: int32 indices).  "
                        "from_dense_uint8 = the reference's own input form, a LIST of dense 0/1 matrices (per-matrix Python work: ~1.5 us each); from_dense_packed = the same matrices as ONE uint8 buffer "
                        "(pageable numpy memory through the library's pinned staging pair; _pinned: one DMA — dense_bytes over PCIe bound it; _device: a device tensor, read where it is).  "
                        "Validation, device-format arrays and both tile tables are built by kernels (csrc/gnx_build_csc.hip); the matrix-core path's tables by its workspace query"}
    E, N, G = g.n_edges, g.n_nodes, g.n_graphs
    if workload == "hetero" and sum(int(n) * int(n) for n in nn) <= 2e8:  # the reference's own input form: dense 0/1 matrices
        adjs = []
        for cp, rv, n in zip(colptrs, rowvals, nn):
            a = np.zeros((n, n), dtype=np.uint8)
            a[rv, np.repeat(np.arange(n), np.diff(cp))] = 1  # A[i, j] = 1 <=> edge i -> j (src = row, dst = column)
            adjs.append(a)
        td = []
        for _ in range(3):
            t0 = time.perf_counter()
            gd = gn.GNGraphBatch(adjs, device=dev)
            torch.cuda.synchronize(dev)
            td.append((time.perf_counter() - t0) * 1e3)
        batch_ms["from_dense_uint8"] = round(min(td), 3)
        batch_ms["from_dense_uint8_first_call"] = round(td[0], 3)  # (includes the one-off allocation of the 64 MB pinned staging buffers)
        assert gd.n_edges == E and gd.n_nodes == N
        # the same matrices as ONE buffer (from_dense_packed): pageable numpy memory, a pinned tensor (one DMA), a device tensor (no copy)
        cat = np.concatenate([a.reshape(-1) for a in adjs])
        for form, buf in (("from_dense_packed", cat), ("from_dense_packed_pinned", torch.from_numpy(cat).pin_memory()), ("from_dense_packed_device", torch.from_numpy(cat).to(dev))):
            tpk = []
            for _ in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                gd = gn.GNGraphBatch.from_dense_packed(buf, nn, device=dev)
                torch.cuda.synchronize(dev)
                tpk.append((time.perf_counter() - t0) * 1e3)
            batch_ms[form] = round(min(tpk), 3)
            assert gd.n_edges == E and gd.n_nodes == N
        batch_ms["dense_bytes"] = int(cat.nbytes)
        del gd, adjs, cat, buf
    blk = bench_weights(gn, din, dout, dev)
    (de, dn, dg), (oe, on, og) = din, dout
    plan = gn.BlockPlan(blk, g, R=1, flags=args.flags)
    if max(din + dout) >= 32:
        # widths that take the matrix-core path: the first workspace query on a NEW handle also builds that path's tables (128-row tiles, the
        # destination of every edge, the aggregation chunks: kernels over the handle's device arrays) — what a loop that rebuilds its batch pays per batch
        tw = []
        for _ in range(3):
            g2 = gn.GNGraphBatch.from_csc_packed(cpc, rvc, nn, device=dev)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            gn.BlockPlan(blk, g2, R=1, flags=args.flags)
            torch.cuda.synchronize(dev)
            tw.append((time.perf_counter() - t0) * 1e3)
            del g2
        batch_ms["wide_tables_with_first_workspace_query"] = round(min(tw), 3)
    nsets = 2 if max(din + dout) >= 64 else NSETS
    tg = torch.Generator(device=dev); tg.manual_seed(1234 + rank)
    mk = lambda T, d: torch.rand((1, T, d), generator=tg, device=dev, dtype=torch.float32) if d > 0 else None
    sets = [dict(ef=mk(E, de), nf=mk(N, dn), gf=mk(G, dg), out=plan.outputs(), ws=plan.new_workspace()) for _ in range(nsets)]
    side = torch.cuda.Stream(device=dev)  # --overlap: the graph update (a few KB, pure latency) runs here

    def step(i, s=None, overlap=False):
        """One GNBlock forward.  `overlap`: two-phase form — edge+node update on the current stream, graph update on
        the side stream behind an event, so it leaves the critical path (joined before the timed region ends)."""
        b = sets[i % nsets]
        go = b["out"][2]
        if not overlap or og == 0:
            plan(b["ef"], b["nf"], b["gf"], b["out"][0], b["out"][1], go, stream=s, ws=b["ws"])
            return
        cur = torch.cuda.current_stream(dev)
        plan(b["ef"], b["nf"], b["gf"], b["out"][0], b["out"][1], go, ws=b["ws"], defer_graph_update=True)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            plan.graph_update(b["gf"], go, ws=b["ws"])

    def sync_all():
        """the contract's bracket at N = 1: torch.cuda.synchronize() (no other rank to wait for)"""
        torch.cuda.synchronize(dev)

    def timed(run):
        sync_all()
        t0 = time.perf_counter()
        run()
        sync_all()
        return time.perf_counter() - t0

    for i in range(max(W, 2)):  # warm-up (also loads the code objects: at least two eager steps before any capture, whatever W is)
        step(i)
    sync_all()

    extra = {}
    pipelined = None
    chained = None
    def capture(nsteps, rotate):
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg, capture_error_mode="thread_local"):
            for i in range(nsteps):
                step(i if rotate else (i & 1), overlap=args.overlap)  # warm: 2 sets (120 MB, cache-resident)
            torch.cuda.current_stream(dev).wait_stream(side)  # join: every graph update is inside the timed region
        return cg
    # The headline form (round 6): the K steps as ONE gnx_block_forward_steps call — the library's own loop over batches: K forwards in order,
    # bit-identical outputs (tests/test_gpu_block.py), and because it knows the next step exists, step i's graph update rides at the front of
    # step i + 1's launch (one launch per step + one flush INSIDE the call, i.e. inside the timed region).  K separate gnx_block_forward calls
    # (two launches per step) are timed beside it: `two_launch_form`.
    steps_form = not args.overlap and not args.separate_calls

    def capture_steps(nsteps, rotate):
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg, capture_error_mode="thread_local"):
            plan.steps([sets[(i if rotate else (i & 1)) % nsets] for i in range(nsteps)])
        return cg
    two_launch = None
    cold = capture_steps(K, True) if steps_form else capture(K, True)
    cold.replay(); torch.cuda.synchronize(dev)
    spin_up(torch, dev, cold.replay)
    reps = sorted(timed(cold.replay) for _ in range(3))
    dt = reps[1]
    if steps_form:
        sep = capture(K, True)
        sep.replay(); torch.cuda.synchronize(dev)
        spin_up(torch, dev, sep.replay, 50.0)
        reps_sep = sorted(timed(sep.replay) for _ in range(3))
        two_launch = {"ms_per_step": round(reps_sep[1] / K * 1e3, 6), "value": round(E / (reps_sep[1] / K), 1), "unit": "edges/s", "steps": K,
                      "reps_us_per_step": [round(r / K * 1e6, 2) for r in reps_sep],
                      "what": f"the same {K} steps as {K} separate gnx_block_forward calls in one hipGraph (two launches per step: block kernel + graph update); rounds 1-5's headline form"}
        del sep
    warm = capture_steps(K, False) if steps_form else capture(K, False)
    warm.replay(); torch.cuda.synchronize(dev)
    spin_up(torch, dev, warm.replay, 50.0)
    extra["warm_ms_per_step"] = round(sorted(timed(warm.replay) for _ in range(3))[1] / K * 1e3, 6)
    extra["launch"] = ((f"gnx_block_forward_steps: {K} steps in one call, one launch per step (step i's graph update at the front of step i+1's launch) + one flush, captured in one hipGraph; "
                        if steps_form else f"hipGraph of {K} gnx_block_forward calls; ") + f"{nsets} rotating buffer sets (cache-cold); " +
                       ("single stream" if not args.overlap else "graph update of step i on a 2nd stream, overlapping step i+1 (joined inside the timed region)"))
    # NOT the headline: the same K steps as TWO hipGraphs (even / odd steps) replayed on two streams.  The steps are independent batches
    # (different buffer sets), so a serving loop would pipeline them: step i's graph-update launch (~4 us of latency on 83 KB) runs under
    # step i+1's block kernel, and one kernel's ramp / drain under its neighbour.  Same results bit for bit (tools/experiments/
    # two_stream_pipeline.py); `value` above stays the single-stream figure, where every step waits for the one before it.
    if K >= 4 and not args.overlap:
        Kp = max(K, 80)  # (two short graphs would measure their own replay overhead: at least 40 steps per stream)

        def capture_half(par):
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, capture_error_mode="thread_local"):
                for i in range(par, Kp, 2):
                    step(i)
            return cg
        halves = (capture_half(0), capture_half(1))
        pstreams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))

        def run_two():
            for st, cg in zip(pstreams, halves):
                with torch.cuda.stream(st):
                    cg.replay()
        run_two(); torch.cuda.synchronize(dev)
        spin_up(torch, dev, run_two, 50.0)
        dt2 = sorted(timed(run_two) for _ in range(3))[1]
        pipelined = {"ms_per_step": round(dt2 / Kp * 1e3, 6), "value": round(E / (dt2 / Kp), 1), "unit": "edges/s", "steps": Kp,
                     "what": f"{Kp} steps of the same kind as two hipGraphs (even / odd steps) on two streams: independent batches pipelined; results bit-identical; not the headline"}
        del halves
    # NOT the headline either: the chained form (gnx_block_forward_chained) — step i's launch carries step i - 1's graph update in
    # workgroups at its front, so a loop over batches is ONE launch per step (opt-in: gf' of a step is complete one call later or after
    # the flush).  Bit-identical outputs (tests/test_gpu_block.py::test_chained_forward...).
    if not args.overlap and og > 0 and not steps_form:
        def capture_chained():
            cgc = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cgc, capture_error_mode="thread_local"):
                pend = None
                for i in range(K):
                    b = sets[i % nsets]
                    pend = plan.chained(b["ef"], b["nf"], b["gf"], *b["out"], ws=b["ws"], prev=pend)
                plan.flush(pend)
            return cgc, pend
        cgc, last = capture_chained()
        cgc.replay(); torch.cuda.synchronize(dev)
        spin_up(torch, dev, cgc.replay, 50.0)
        dtc = sorted(timed(cgc.replay) for _ in range(3))[1]
        chained = {"ms_per_step": round(dtc / K * 1e3, 6), "value": round(E / (dtc / K), 1), "unit": "edges/s", "steps": K,
                   "what": f"gnx_block_forward_chained: {K} steps + one flush in one hipGraph — every step ONE launch (the previous step's graph update rides in "
                           "workgroups at the front of the next block kernel); opt-in for loops over batches, results bit-identical; not the headline"}
        del cgc
    extra["timing"] = f"median of 3 runs of the {K}-step region ({[round(r / K * 1e6, 2) for r in reps]} us/step), after {CLOCK_WARMUP_MS:g} ms of the same load, untimed (clock settling: bench.py::spin_up)"
    ms_per_step = dt / K * 1e3
    E_job = E
    value = E_job / (dt / K)

    # ---- roofline: the same K steps again with per-kernel dispatch timestamps (block_roofline) ----
    roof = None
    if rank == 0:
        dims_key = args.dims.replace(":", "_").replace(",", "-") + ("" if workload == "c2" else f"_hetero{G}")
        if workload != "c2" and args.hetero_edges != 1_000_000:
            dims_key += f"_{args.hetero_edges}"  # the committed PMC profiles are of the 1M-edge batches: no traffic figure for another size
        roof = block_roofline(gn, torch, dev, plan, sets, nsets, K, E, N, G, din, dout, ms_per_step, dims_key,
                              headline_traffic_breakdown=(workload == "c2" and args.dims == "readme" and args.c2_scale == 1.0), steps_form=steps_form)
        if two_launch is not None and roof.get("bound") == "hbm" and "algorithmic_bytes" in roof:
            roof["frac_whole_step_two_launch"] = round(roof["algorithmic_bytes"] / (two_launch["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # the headline's traffic measured in THIS run (two counter passes over a child of the same workload); the committed figure stays beside it
        if (workload == "c2" and args.dims == "readme" and args.c2_scale == 1.0 and world == 1 and not multi and not args.full_line and not args.no_live_traffic
                and not args.no_secondary and roof.get("kernel")):
            live, lrec = live_traffic(roof["kernel"], ["--gpus", "1", "--steps", str(K), "--warmup", str(W)])
            quoted = roof.get("traffic")
            if live is not None:
                roof["traffic"] = live
                if lrec.get("kernel_us_rocprofv3") is not None:
                    roof["kernel_us_rocprofv3"] = lrec["kernel_us_rocprofv3"]  # the profiler's average of the same kernel in a child of this run, beside `kernel_us`
                lrec["committed_profile_figure"] = quoted
                if quoted:
                    lrec["vs_committed"] = round(live / quoted, 4)
                lrec["committed_profile"] = roof.get("traffic_source")
                roof["traffic_source"] = lrec
            else:
                roof["traffic_live"] = lrec  # (why not; `traffic` is then the committed profile's, as before)

    # ---- CPU baseline: the oracle's C restatement ("port") on the host cores, rank 0, N = 1 only ----
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import c_port
        p = dict(in_dims=din, out_dims=dout, We=blk.edgefn.weight.cpu().numpy(), be=np.zeros(oe, np.float32),
                 Wn=blk.nodefn.weight.cpu().numpy(), bn=np.zeros(on, np.float32), Wg=blk.graphfn.weight.cpu().numpy(),
                 bg=np.zeros(og, np.float32), act_e=0, act_n=0, act_g=0)
        b = sets[0]
        host = lambda a: None if a is None else a.cpu().numpy()
        csc = (*g.csc(), g.node_off, g.edge_off)
        cores = min(os.cpu_count() or 1, c_port.max_threads())
        runner = c_port.BlockRunner(p, csc, host(b["ef"]), host(b["nf"]), host(b["gf"]), nthreads=cores)
        runner.run()
        t_budget, times = (args.cpu_budget if args.cpu_budget is not None else (3.0 if args.dims == "readme" else 12.0)), []
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < t_budget or len(times) < 3:
            t0 = time.perf_counter(); out = runner.run(); times.append(time.perf_counter() - t0)
        cpu = dict(value=round(E / float(np.median(times)), 1), unit="edges/s", cores=cores, kind="port",
                   sample=f"{len(times)} full forwards of the same 1M-edge batch (median), oracle/gn_oracle_c.c with OpenMP on {cores} threads")
        # the checker also checks: HIP output of set 0 vs the C port on the same inputs (loose: both fp32)
        plan(b["ef"], b["nf"], b["gf"], *b["out"], ws=b["ws"]); torch.cuda.synchronize(dev)
        for got, ref in zip(b["out"], out):
            if got is not None:
                assert np.allclose(got.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(ref).max()))), "HIP vs CPU port mismatch"

    dense = None
    if rank == 0 and args.dense_baseline:
        from oracle import gn_oracle as O
        rngd = np.random.default_rng(7)
        pd_ = O.make_block_params(rngd, din, dout) if args.dims == "readme" else O.make_block_params(rngd, (10, 5, 0), (3, 4, 5))
        dense = {}
        for name, adjs in (("README ex.1 (3 nodes, 5 edges, batch 2)", [np.array([[1, 0, 1], [1, 1, 0], [0, 0, 1]])] * 2),
                           ("64 graphs, 16..64 nodes, density 0.1", [(rngd.random((n, n)) < 0.1).astype(np.int64) for n in rngd.integers(16, 65, 64)])):
            efs = [rngd.random((10, int(a_.sum())), dtype=np.float32) for a_ in adjs]
            nfs = [rngd.random((5, a_.shape[0]), dtype=np.float32) for a_ in adjs]
            t0 = time.perf_counter(); xb = O.batch_dense(adjs, efs, nfs, None); t_batch = time.perf_counter() - t0
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); O.block_forward_dense(pd_, xb); ts.append(time.perf_counter() - t0)
            ne = int(sum(a_.sum() for a_ in adjs))
            dense[name] = dict(edges=ne, batch_s=round(t_batch, 4), forward_s=round(min(ts), 4), edges_per_s=round(ne / min(ts), 1))
    if rank == 0:
        line = {
            "metric": "edges updated/sec, GNBlock fwd, 1M-edge batch", "value": round(value, 1), "unit": "edges/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 6), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_name, "dims": f"{din}=>{dout}", "edges_per_gpu": E, "nodes_per_gpu": N,
                       "graphs_per_gpu": G, "edges_whole_job": E_job, "parallelism": "single GPU", **extra},
            "roofline": roof, "cpu_baseline": cpu, "batch_ms": batch_ms,
        }
        if c_abi is not None:
            if "captured_us_per_step" in c_abi:
                # the C program times the headline's form (ONE gnx_block_forward_steps call, captured) and rounds 1-5's (K gnx_block_forward calls, captured)
                c_us = c_abi.get("steps_us_per_step", c_abi["captured_us_per_step"]) if steps_form else c_abi["captured_us_per_step"]
                line["c_abi_ms_per_step"] = round(c_us * 1e-3, 6)
                line["c_abi"] = {"vs_torch_captured": round(c_us * 1e-3 / ms_per_step, 4), "what": c_abi.get("steps_what" if steps_form and "steps_us_per_step" in c_abi else "captured_what"),
                                 "steps_form_ms_per_step": round(c_abi["steps_us_per_step"] * 1e-3, 6) if "steps_us_per_step" in c_abi else None,
                                 "two_launch_form_ms_per_step": round(c_abi["captured_us_per_step"] * 1e-3, 6),
                                 "reps_us": c_abi.get("captured_reps_us"), "event_us_per_step": c_abi.get("captured_event_us_per_step"),
                                 "model_forward_ms_per_step": round(c_abi["model_us_per_step"] * 1e-3, 6), "model_forward_what": c_abi.get("model_what"),
                                 "batch_ms": c_abi.get("batch_ms"), "steps": c_abi.get("steps"), "program": "tests/c/abi_bench.c --mode block"}
            else:
                line["c_abi_ms_per_step"], line["c_abi"] = None, c_abi
        if two_launch is not None:
            line["two_launch_form"] = two_launch
        if pipelined is not None:
            line["pipelined_two_streams"] = pipelined
        if chained is not None:
            line["chained_graph_update"] = chained
        if secondary is not None:
            line["secondary"] = secondary
        if dense is not None:
            line["cpu_dense_formulation_restatement"] = dense  # NOT the reference: its formulation restated in numpy (B2)
    # The JSON line must be the LAST thing on stdout: with NCCL_DEBUG=VERSION (set on the GPU boxes) RCCL writes its version
    # banner through C stdio, which on a pipe is only flushed at exit — after Python's print.  Every rank tears the process
    # group down and flushes C stdio first; rank 0 prints once the others are done.
    _flush_c_stdio()
    if rank == 0:
        emit(line, args, "default" if headline and secondary is not None else "block")


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


if __name__ == "__main__":
    main()

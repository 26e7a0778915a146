"""bench.py's result line (the task's contract + VERDICT r5 item 1): the LAST stdout line is the compact headline — the contract's keys, `roofline`,
`cpu_baseline`, numbers only, below 3 KB whatever was measured — and everything else (secondary configs, prose) goes to an earlier PREFIXED line and
to gpurun_out/bench_detail_*.json.  Assembled here from canned results: round 5's own 20.6-KB line (the one the driver could not parse) and an
8-rank sharded run."""
import argparse
import io
import json
import os
from contextlib import redirect_stdout

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def check_compact(c, n_gpus):
    raw = json.dumps(c)
    assert len(raw) < 4096 and len(raw) <= bench.COMPACT_MAX_BYTES, len(raw)
    for k in CONTRACT:
        assert k in c, k
    assert c["n_gpus"] == n_gpus and c["unit"] == "edges/s" and c["higher_is_better"] is True and c["data"] == "synthetic" and c["dtype"] == "f32"
    assert c["value"] > 0 and c["ms_per_step"] > 0 and c["vs_baseline"] is None
    assert "workload" in c["config"] and "model" not in c["config"]
    r = c["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 < r["frac"] <= 1 and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 or r["bound"] == "mfma"
    for v in c.values():  # no prose beyond the short fields
        assert not isinstance(v, str) or len(v) <= 200
    return raw


def r05_line():
    with open(os.path.join(ROOT, "profiles", "bench_r05_default.json")) as f:
        line = json.load(f)
    assert len(json.dumps(line)) > 16000  # the line that was too long for the driver
    return line


def test_compact_headline_of_the_round_5_line_is_below_3_kb():
    line = r05_line()
    c = bench.compact_line(line, "gpurun_out/bench_detail_default.json")
    check_compact(c, 1)
    assert c["value"] == line["value"] and c["ms_per_step"] == line["ms_per_step"] and c["steps"] == line["steps"] and c["warmup"] == line["warmup"]
    assert c["roofline"]["kernel"] == "k_block_wave" and c["roofline"]["kernel_us"] == line["roofline"]["kernel_us"]
    assert c["roofline"]["algorithmic_bytes"] == 60000580 and c["roofline"]["frac_whole_step"] == line["roofline"]["frac_whole_step"]
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] == line["cpu_baseline"]["cores"] and c["cpu_baseline"]["value"] > 0
    assert "secondary" not in c and "batch_ms" not in c and "c_abi" not in c
    assert c["c_abi_ms_per_step"] == line["c_abi_ms_per_step"] and c["chained_ms"] == line["chained_graph_update"]["ms_per_step"]
    s = bench.secondary_summary(line["secondary"])
    assert set(s) == {k for k, v in line["secondary"].items() if isinstance(v, dict)}
    assert s["c4"]["roof_bound"] == "mfma" and s["c5w_one_gpu"]["roof_frac"] > 0.6 and "what" not in json.dumps(s)
    assert len(json.dumps(s)) < 6000


def canned_sharded(world, E_total, roof):
    per = [[E_total // world, 590_000 // world, 4096 // world] for _ in range(world)]
    single = {"n_gpus": 1, "ms_per_step": 0.0221, "value": E_total / 22.1e-6, "unit": "edges/s", "graphs": 4096, "edges": E_total, "what": "x" * 300}
    return {"E_job": E_total, "G_job": 4096, "seed": 5, "M": 20, "nsets": 64, "dt": 20 * 9.1e-6, "dt_without": 20 * 8.7e-6, "reps": [20 * 9.0e-6, 20 * 9.1e-6, 20 * 9.4e-6],
            "reps_without": [20 * 8.6e-6, 20 * 8.7e-6, 20 * 8.8e-6], "per_rank": per, "single": single, "E": per[0][0], "N": per[0][1], "G": per[0][2], "roof": roof}


@pytest.mark.parametrize("world", [2, 8])
def test_compact_headline_of_a_sharded_run_is_below_3_kb(world):
    sec = r05_line()["secondary"]
    roof = dict(sec["c5_force_dist"]["roofline"], counts="c" * 400, kernel_us_source="k" * 400, all_kernels_us={"k_block_wave": 9.0, "k_graph_t": 4.0})
    roofw = dict(sec["c5w_one_gpu"]["roofline"], counts="c" * 400)
    args = argparse.Namespace(scaling="strong", full_line=False)
    din, dout = bench.DIMS["readme"]
    line = bench.sharded_line(args, canned_sharded(world, 1_000_000, roof), canned_sharded(world, 8_000_000, roofw), 20, 5, world, din, dout)
    assert len(json.dumps(line)) > 3500  # the whole line is NOT what gets printed last
    c = bench.compact_line(line, f"gpurun_out/bench_detail_sharded_n{world}.json")
    check_compact(c, world)
    assert c["scaling"] == "strong" and c["config"]["edges_whole_job"] == 1_000_000 and c["config"]["graphs_whole_job"] == 4096
    assert c["config"]["graphs_per_gpu"] == 4096 // world and "configs[4]" in c["config"]["workload"]
    assert c["with_allgather_ms"] == c["ms_per_step"] and 0 < c["without_allgather_ms"] <= c["with_allgather_ms"] and c["single_gpu_same_workload_ms"] == 0.0221
    assert c["roofline"]["whole_job"]["peak"] == 8000.0 * world and c["c5w"]["value"] > c["value"]
    assert c["cpu_baseline"] is None


def test_emit_prints_the_compact_line_last_and_keeps_the_whole_line_in_a_file(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    line = r05_line()
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(line, argparse.Namespace(full_line=False), "default")
    out = buf.getvalue().strip().splitlines()
    assert len(out) == 2 and out[0].startswith("bench-secondary (not the result line): ") and not out[0].lstrip().startswith("{")
    c = json.loads(out[-1])
    check_compact(c, 1)
    assert len(out[-1]) <= bench.COMPACT_MAX_BYTES
    with open(os.path.join(str(tmp_path), c["detail"])) as f:
        assert json.load(f) == line  # nothing is lost: the detail file is the whole line
    json.loads(out[0].split(": ", 1)[1])
    # the children of collect_secondary (and the detail tests) ask for the whole line
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(line, argparse.Namespace(full_line=True), "default")
    assert json.loads(buf.getvalue()) == line


def test_compact_line_survives_pathological_prose():
    line = r05_line()
    line["config"]["workload"] = "w" * 5000
    line["config"]["launch"] = "l" * 5000
    line["cpu_baseline"]["sample"] = "s" * 5000
    line["metric"] = "m" * 1000
    check_compact(bench.compact_line(line, None), 1)


def test_live_traffic_falls_back_without_a_gpu_and_the_compact_line_says_where_the_figure_is_from(monkeypatch):
    """roofline.traffic is measured in the driver's own run (two rocprofv3 counter passes over a child); when that cannot be done — no profiler,
    no GPU (here), a pass that overruns — the call returns None with the reason and the committed profile's figure stays; the compact line names
    the origin either way and stays under its cap."""
    import bench
    got, rec = bench.live_traffic("k_block_wave", ["--gpus", "1", "--steps", "2", "--warmup", "1"], timeout_s=120)
    assert got is None and rec["live"] is False and rec["why"]
    monkeypatch.setattr(__import__("shutil"), "which", lambda name: None)
    monkeypatch.setattr(bench.os.path, "exists", lambda p, _e=bench.os.path.exists: False if p.endswith("rocprofv3") else _e(p))
    got, rec = bench.live_traffic("k_block_wave", [], timeout_s=5)
    assert got is None and rec == {"live": False, "why": "rocprofv3 not found"}
    monkeypatch.undo()
    line = json.load(open(os.path.join(ROOT, "profiles", "bench_r06_default.json")))
    line["roofline"]["traffic"] = 79676633.1
    line["roofline"]["traffic_source"] = {"live": True, "how": "x" * 500, "committed_profile_figure": 79681095.9}
    c = bench.compact_line(line, "gpurun_out/bench_detail_default.json")
    check_compact(c, 1)
    assert c["roofline"]["traffic"] == 79676633.1 and c["roofline"]["traffic_measured"].startswith("this run")
    line["roofline"]["traffic_source"] = {"file": "profiles/traffic_readme.json", "commit": None}
    assert bench.compact_line(line)["roofline"]["traffic_measured"].startswith("profiles/")

"""The library's default forms beside foreign matrix kernels, its own turn-taking switched off.

Round 5 (profiles/r05_mfma_mix_hazard.log): kernels on v_mfma_f32_32x32x2f32 came out wrong now and then while a dense bf16 matrix kernel was resident
through another queue.  From round 6 on no default path issues that instruction (csrc/gnx_x6_mma.h: k_rows_gemm, k_ffn_fused and k_dw_gemm carry their
products as six bf16 terms; the fp32 forms run only where a call's flags ask).  tests/overlap_probe.py runs in a process of its own with
GNX_ALLOW_OVERLAP=1 (the DeviceTurn guard off; the variable is read once per process): a GNCore forward + backward loop beside a torch bf16 GEMM loop on
another stream, and two captured graphs replayed concurrently on two streams — every result bit-identical to the serial run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(*argv):
    env = dict(os.environ, GNX_ALLOW_OVERLAP="1")
    for k in ("GNX_FFN_FP32", "GNX_EDGE_FP32", "GNX_PROJ_FP32", "GNX_EDGE_NARROW_FP32"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "overlap_probe.py"), *argv], capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("big", ["0", "1"])
def test_default_forms_are_exact_beside_a_bf16_gemm_loop_and_under_concurrent_graph_replays(big):
    """small batch (1200 nodes: the general kernels — k_rows_gemm, k_ffn_fused, k_dw_gemm, all on six bf16 terms now) and big batch (4500 nodes: the
    prepared six-term kernels); forward AND backward; 0 of the runs may differ from the serial results"""
    r = _probe("default", "40", big)
    assert r["overlap_allowed"] == "1" and r["forward_backward_runs"] == 40 and r["graph_replay_pairs"] == 40
    assert r["forward_backward_wrong"] == 0, r
    assert r["graph_replays_wrong"] == 0, r

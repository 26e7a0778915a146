"""The library's forms beside foreign matrix kernels and beside each other (profiles/r06_overlap_hazard.log).

Round 5: k_rows_gemm / k_ffn_fused came out wrong now and then — row pairs ~1 % off — while a dense bf16 matrix kernel was resident through another
queue; the fp32 matrix instruction was blamed and the library's matrix-core calls were serialised per device.  Round 6 took that instruction off every
default path (csrc/gnx_x6_mma.h), found the damage unchanged, and bisected it to ONE site: the LayerNorm-on-load branch of those kernels used an LDS
read of the row statistics right behind the compiler's counted wait, and on a CU shared with another kernel's workgroups the last 16 lanes of a wave
got stale values.  With the guard there (GNX_LN_GUARD) every form is exact with overlapping calls, so the turn-taking is off by default
(GNX_TAKE_TURNS=1 brings it back).  tests/overlap_probe.py, a process of its own: a GNCore forward + backward loop beside a torch bf16 GEMM loop on
another stream, and two captured graphs replayed concurrently on two streams — every result bit-identical to the serial run — for the default forms,
the statistics-table forms (GNX_FLAG_LN_ON_LOAD: the branch that failed 54-96 times in 120) and the fp32-instruction forms (round 5's victims)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(*argv):
    env = dict(os.environ)
    for k in ("GNX_FFN_FP32", "GNX_EDGE_FP32", "GNX_PROJ_FP32", "GNX_EDGE_NARROW_FP32", "GNX_TAKE_TURNS"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "overlap_probe.py"), *argv], capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("forms,big", [("default", "0"), ("default", "1"), ("flags:0x20000", "0"), ("flags:0x21000", "0"), ("fp32", "0")],
                         ids=["default-small", "default-big", "ln-on-load", "ln-on-load-one-stream", "fp32-forms"])
def test_forms_are_exact_beside_a_bf16_gemm_loop_and_under_concurrent_graph_replays(forms, big):
    """small batch (1200 nodes: the general kernels — k_rows_gemm, k_ffn_fused, k_dw_gemm) and big batch (4500 nodes: the prepared six-term
    kernels); forward AND backward; calls overlap (no turn-taking); 0 of the runs may differ from the serial results"""
    r = _probe(forms, "40", big)
    assert r["turn_taking"] == "off" and r["forward_backward_runs"] == 40 and r["graph_replay_pairs"] == 40
    assert r["forward_backward_wrong"] == 0, r
    assert r["graph_replays_wrong"] == 0, r

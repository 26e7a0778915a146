"""CPU-side checks of the C-ABI boundary: libgnx.so loads, exports every symbol include/gnx.h declares, validates
arguments before touching the GPU, and fails loudly (no fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    import graphnets_jl_amd as gn
    return gn._lib.load()


def _declared():
    with open(os.path.join(ROOT, "include", "gnx.h")) as f:
        return re.findall(r"^GNX_API [\w\s\*]+?\b(gnx_\w+)\(", f.read(), flags=re.M)


def test_every_declared_symbol_is_exported_and_bound(lib):
    import graphnets_jl_amd as gn
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gnx.h but not exported by libgnx.so"
    assert set(names) == set(gn._lib.SIGNATURES), "python binding and header disagree"
    assert lib.gnx_version() == 130
    # ... and with the same number of parameters (ctypes would only notice at call time)
    with open(os.path.join(ROOT, "include", "gnx.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    for m in re.finditer(r"GNX_API [\w\s\*]+?\b(gnx_\w+)\(([^;]*?)\);", text, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(gn._lib.SIGNATURES[name][1]), f"{name}: header declares {n} parameters, the binding {len(gn._lib.SIGNATURES[name][1])}"


def test_struct_layouts_match_header():
    import graphnets_jl_amd as gn
    L = gn._lib
    assert C.sizeof(L.Dense) == 24 and C.sizeof(L.BlockParams) == 24 + 3 * 24 + 8  # (+ the `prepared` pointer)
    assert C.sizeof(L.CoreParams) == 104 + 6 * 16 + 3 * 48 + 8 + 8
    assert C.sizeof(L.GraphsInfo) == 64 and C.sizeof(L.ProfileEntry) == 72


def test_argument_validation_happens_before_any_gpu_work(lib):
    h = C.c_void_p(None)
    nn = (C.c_int64 * 1)(2)
    bad = np.array([[1, 2], [0, 1]], dtype=np.int64)
    ptrs = (C.c_void_p * 1)(bad.ctypes.data)
    assert lib.gnx_graphs_create_dense(ptrs, nn, 0, 2, 1, C.byref(h)) == -2   # GNX_ERR_NO_GRAPHS (checks.jl:8)
    assert lib.gnx_graphs_create_dense(ptrs, nn, 1, 2, 1, C.byref(h)) == -4   # GNX_ERR_ADJ_VALUE (pad.jl:30)
    assert b"0 or 1" in lib.gnx_last_error()
    assert lib.gnx_graphs_create_dense(ptrs, nn, 1, 9, 1, C.byref(h)) == -1   # bad elem kind
    cp = np.array([0, 1, 3], dtype=np.int64)
    rv = np.array([0, 1, 0], dtype=np.int64)                                   # unsorted inside column 1
    cpp, rvp = (C.c_void_p * 1)(cp.ctypes.data), (C.c_void_p * 1)(rv.ctypes.data)
    assert lib.gnx_graphs_create_csc(cpp, rvp, nn, 1, 0, C.byref(h)) == -7    # GNX_ERR_CSC
    assert lib.gnx_block_workspace_bytes(None, None, 1) == 0


def test_no_silent_cpu_fallback(lib):
    """Without a GPU a well-formed request must fail with a HIP error (> 0), never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p(None)
    nn = (C.c_int64 * 1)(2)
    ok = np.array([[1, 0], [1, 1]], dtype=np.int64)
    ptrs = (C.c_void_p * 1)(ok.ctypes.data)
    assert lib.gnx_graphs_create_dense(ptrs, nn, 1, 2, 1, C.byref(h)) > 0
    assert not h.value


def test_runtime_specialised_kernel_source_compiles_for_gfx950(lib):
    """The kernel text embedded in libgnx.so (csrc/gnx_device.h + gnx_wave_kernel.h) compiles with hiprtc for an
    unlisted width set — no GPU needed for the compile step; ineligible width sets are refused."""
    import graphnets_jl_amd as gn
    L = gn._lib
    n = C.c_size_t(0)
    assert lib.gnx_jit_precompile(C.byref(L.BlockParams(7, 3, 2, 5, 6, 1)), 128, C.byref(n)) == 0, lib.gnx_last_error()
    assert n.value > 4096
    assert lib.gnx_jit_precompile(C.byref(L.BlockParams(40, 3, 2, 5, 6, 1)), 128, C.byref(n)) == -6   # GNX_ERR_DIMS: too wide (> 32)
    assert lib.gnx_jit_precompile(C.byref(L.BlockParams(24, 4, 0, 3, 4, 5)), 128, C.byref(n)) == 0    # mid widths are eligible
    assert lib.gnx_jit_precompile(C.byref(L.BlockParams(16, 16, 16, 16, 16, 16)), 128, C.byref(n)) == -6  # too many weights: MFMA path
    assert lib.gnx_jit_precompile(C.byref(L.BlockParams(7, 3, 2, 5, 6, 1)), 100, C.byref(n)) == -1
    # the narrow GNCore's one-launch FeedForward kernel (hand-streamed scalar weights, packed FMAs: inline asm) for another width triple
    n.value = 0
    assert lib.gnx_jit_precompile_core_post(6, 4, 2, C.byref(n)) == 0, lib.gnx_last_error()
    assert n.value > 1000
    assert lib.gnx_jit_precompile_core_post(17, 4, 2, C.byref(n)) == -6
    st = (C.c_int64 * 4)()
    assert lib.gnx_jit_stats(st) == 0 and st[0] >= 1 and st[2] == 0


def _make_c_tests():
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
    return os.path.join(ROOT, "tests", "c", "_build")


def test_c_language_binding_compiles_links_and_validates():
    """tests/c/abi_smoke.c — include/gnx.h consumed by a C compiler (gcc, -Wall -Werror), struct layout by the compiler, linked
    against libgnx.so like a `ccall` host: symbol resolution, version, argument validation (no GPU needed for these)."""
    import subprocess
    out = subprocess.run([os.path.join(_make_c_tests(), "abi_smoke"), "--symbols"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "symbols ok" in out.stdout


def test_host_shim_under_address_sanitizer():
    """libgnx's host translation units built with g++ -fsanitize=address,undefined and driven through adjacency /
    CSC validation, tile construction, failed handle creation, model validation, the run-time specialiser and the profiler."""
    import subprocess
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([os.path.join(_make_c_tests(), "host_asan_driver")], capture_output=True, text=True, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "host_asan_driver: ok" in out.stdout


@pytest.mark.gpu
def test_readme_example_1_through_the_c_abi_from_c():
    """The full C program: README example 1 built, run on the GPU and checked against an inline double-precision evaluation."""
    import subprocess
    out = subprocess.run([os.path.join(_make_c_tests(), "abi_smoke")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "max |hip - double|" in out.stdout


@pytest.mark.gpu
def test_c_abi_bench_runs_both_modes_without_python_in_the_loop(tmp_path):
    """tests/c/abi_bench.c — the throughput program bench.py quotes as `c_abi_ms_per_step`: BASELINE configs[1] through
    gnx_block_forward (captured by the C program) and through gnx_model_forward, and the 4-layer configs[3] model as one gnx_model.
    Small sizes here (it is the bench line that runs the full ones); a graph handed over as a file takes the --csc path."""
    import json
    import subprocess
    exe = os.path.join(_make_c_tests(), "abi_bench")
    rng = np.random.default_rng(5)
    N, E = 3000, 30000
    k = np.sort(rng.choice(N * N, E, replace=False))
    cp = np.zeros(N + 1, dtype=np.int64)
    np.add.at(cp, k // N + 1, 1)
    path = str(tmp_path / "g.bin")
    with open(path, "wb") as f:
        f.write(np.array([N, E], dtype=np.int64).tobytes() + np.cumsum(cp).astype(np.int64).tobytes() + (k % N).astype(np.int64).tobytes())
    for argv, keys in ((["--mode", "block", "--steps", "16", "--warmup", "4", "--csc", path], ("captured_us_per_step", "steps_us_per_step", "model_us_per_step")),
                       (["--mode", "block", "--steps", "8", "--nodes", "2000", "--edges", "16000"], ("captured_us_per_step", "steps_us_per_step", "model_us_per_step")),
                       (["--mode", "c4", "--steps", "3", "--warmup", "2", "--csc", path], ("model_us_per_step",)),
                       (["--mode", "c4", "--steps", "3", "--warmup", "2", "--csc", path, "--core-dims", "10,5,3"], ("model_us_per_step",))):
        out = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        line = json.loads(out.stdout.strip().splitlines()[-1])
        for key in keys:
            assert 0.5 < line[key] < 1e6, line
        assert np.isfinite(line["gf_out0"]) and line["batch_ms"] > 0

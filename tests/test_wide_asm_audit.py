"""Static audit of the matrix-core GEMM's device code (gnx_wide.hip compiled to gfx950 assembly here, no GPU needed) for the two
properties its speed AND its correctness rest on and that only the .s shows:

1. no register in scratch memory in the kernels of the hot launches (projected edge update, node update, projections, encoder /
   decoder loaders): a scratch reload is a vector-memory operation — its wait drains every global load in flight;
2. the source rows that the NL = 3 epilogue requests with loads the compiler does not track (inline asm) are not read, copied or
   spilled between the loads and the counted wait that retires them (the destination counts as written at the asm statement, so a
   compiler-inserted copy would read a register whose data has not landed)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "graphnets.jl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def wide_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    # the assembly (30 MB) is kept under gpurun_out/ (git-ignored, and not part of the snapshot that travels to a GPU box), keyed by the hash of
    # everything the translation unit includes: a two-minute compile once per source state instead of once per test session
    import glob
    import hashlib
    srcs = [os.path.join(CSRC, "gnx_wide.hip")] + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "gnx.h")]
    key = hashlib.sha256(b"".join(open(f, "rb").read() for f in srcs)).hexdigest()[:16]
    obj = os.path.join(ROOT, "gpurun_out", ".cache")
    try:
        os.makedirs(obj, exist_ok=True)
    except OSError:
        pass
    out = os.path.join(obj, f"gnx_wide.audit.{key}.s") if os.path.isdir(obj) and os.access(obj, os.W_OK) else str(tmp_path_factory.mktemp("asm") / "gnx_wide.s")
    if not os.path.exists(out):
        for old in glob.glob(os.path.join(obj, "gnx_wide.audit.*.s")):
            os.remove(old)
        tmp = out + ".tmp%d" % os.getpid()
        cmd = [HIPCC, "-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, "gnx_wide.hip"), "-o", tmp, "-O3", "--offload-arch=gfx950", "-std=c++17",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-fno-gpu-rdc"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        os.replace(tmp, out)
    text = open(out).read()
    kernels = {}
    for m in re.finditer(r"^(_ZN3gnx11k_rows_gemmI\w+):\s*(?:;.*)?$", text, re.M):
        name = m.group(1)
        end = text.index(".amdhsa_kernel " + name, m.end())
        meta_end = text.index(".end_amdhsa_kernel", end)
        kernels[name] = (text[m.end():end], text[end:meta_end])
    assert kernels, "no k_rows_gemm instantiation found in the assembly"
    return kernels


def _args(name):
    """template arguments (BN, VEC4, KC, NL, TRANS, LD) from the mangled name"""
    m = re.search(r"k_rows_gemmILi(\d+)ELb([01])ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)E", name)
    return tuple(int(x) for x in m.groups())


def _x6(name):
    """the seventh template argument: the six-term form (round 6: the default; 0 = v_mfma_f32_32x32x2f32, only where a call's flags ask)"""
    return int(re.search(r"k_rows_gemmILi\d+ELb[01]ELi\d+ELi\d+ELb[01]ELi\d+ELb([01])E", name).group(1))


def test_hot_gemm_kernels_keep_no_register_in_scratch_memory(wide_asm):
    checked = 0
    for name, (_, meta) in wide_asm.items():
        bn, vec4, kc, nl, trans, ld = _args(name)
        if not vec4:
            continue  # element-output kernels with epilogue operands (rare widths) still spill: not on any benchmarked path
        if bn == 64 and nl == 2:
            continue  # 64-column two-stream epilogue at four workgroups per CU (small launches only)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        if _x6(name):
            # the six-term form holds three bf16 parts of every fragment (~25 more live registers): the 64-column loaders with row sums (the
            # node updates of config 4's encoder / decoder at four workgroups per CU) stay spill-free, the 128-column kernels at three
            # workgroups per CU keep a bounded handful of registers in scratch memory (measured against two workgroups per CU without any:
            # profiles/r06_ab_wide_wpe.log)
            if bn == 64 and ld >= 1:
                assert scratch == 0, (name, scratch)
            if bn == 128 and (nl in (0, 3) or ld >= 1):
                assert scratch <= 64, (name, scratch)
            checked += 1
            continue
        if bn == 128 and nl == 0 and ld == 0 and not trans:
            assert scratch == 0, (name, scratch)
        if nl == 3 or ld >= 1:
            assert scratch == 0, (name, scratch)
        checked += 1
    assert checked >= 20


def test_untracked_source_row_loads_are_not_touched_before_their_wait(wide_asm):
    audited = 0
    for name, (body, _) in wide_asm.items():
        if _args(name)[3] != 3:
            continue
        lines = body.split("\n")
        loads = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l and i > 0 and "s_nop 4" in lines[i - 1]]
        waits = [i for i, l in enumerate(lines) if "s_cmp_eq_u32" in l and i + 2 < len(lines) and "s_waitcnt vmcnt" in lines[i + 2]]
        assert loads and waits, name
        wait = waits[-1]
        assert all(i < wait for i in loads), name
        regs = {}
        for i in loads:
            m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", lines[i])
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                regs[r] = i
        for i in range(loads[0], wait):
            if i in loads:
                continue
            code = lines[i].split(";")[0]
            used = {int(x) for x in re.findall(r"\bv(\d+)\b", code)}
            for m in re.finditer(r"v\[(\d+):(\d+)\]", code):
                used |= set(range(int(m.group(1)), int(m.group(2)) + 1))
            for r in used & set(regs):
                assert i < regs[r], f"{name}: '{code.strip()}' touches v{r} between its untracked load and the wait"
        audited += 1
    assert audited >= 2


def test_lds_dma_of_the_destination_rows_is_retired_before_the_barrier_that_publishes_it(wide_asm):
    """NL = 3: the tile's destination projections are written into LDS by `global_load_lds` from ONE wave and read by the
    others.  The source has no explicit wait — the design relies on the `s_waitcnt vmcnt(0)` the compiler places in front of the
    next workgroup barrier.  This walks the control-flow graph of the compiled kernel from every LDS-DMA instruction: on EVERY path
    the first `s_barrier` reached must come after an `s_waitcnt` with `vmcnt(0)` (and no later LDS-DMA).  A compiler update that
    drops or weakens that wait turns into a test failure instead of a silent race."""
    audited = 0
    for name, (body, _) in wide_asm.items():
        if _args(name)[3] != 3:
            continue
        lines = [l.split(";")[0].rstrip() for l in body.split("\n")]
        label_at = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                label_at[m.group(1)] = i
        dma = [i for i, l in enumerate(lines) if "global_load_lds" in l]
        assert dma, name

        def successors(i):
            code = lines[i].strip()
            if code.startswith("s_endpgm"):
                return []
            m = re.match(r"s_branch\s+(\.LBB\w+)", code)
            if m:
                return [label_at[m.group(1)]]
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)", code)
            if m:
                return [label_at[m.group(1)], i + 1]
            assert not re.match(r"s_setpc|s_swappc|s_call", code), (name, code)  # no indirect control flow expected
            return [i + 1] if i + 1 < len(lines) else []

        barriers_checked = 0
        seen = set()
        stack = [(i + 1, False) for i in dma]
        while stack:
            i, waited = stack.pop()
            while i < len(lines):
                if (i, waited) in seen:
                    break
                seen.add((i, waited))
                code = lines[i].strip()
                if "global_load_lds" in code:
                    waited = False
                elif code.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", code):
                    waited = True
                elif code.startswith("s_barrier"):
                    assert waited, f"{name}: s_barrier at .s line {i} reachable from an LDS-DMA without s_waitcnt vmcnt(0) in between"
                    barriers_checked += 1
                    break  # published: later barriers are not this audit's business
                nxt = successors(i)
                if not nxt:
                    break
                for j in nxt[:-1]:
                    stack.append((j, waited))
                i = nxt[-1]
        assert barriers_checked >= 1, name
        audited += 1
    assert audited >= 2

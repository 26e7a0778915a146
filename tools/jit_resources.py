#!/usr/bin/env python3
"""Compile the run-time specialised kernels for the given width sets (no GPU needed) and print their register / LDS /
scratch usage from the code-object notes.   python tools/jit_resources.py 16,8,4:8,16 12,12,4:12,12 ...  [--ept 2]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ept = int(sys.argv[sys.argv.index("--ept") + 1]) if "--ept" in sys.argv else 2
    with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None) as d:
        os.environ["GNX_JIT_CACHE"] = d
        import graphnets_jl_amd as gn
        L = gn._lib
        lib = L.load()
        for spec in args:
            i, o = spec.split(":")
            dims = [int(v) for v in i.split(",")] + [int(v) for v in o.split(",")]
            dims += [1] * (6 - len(dims))
            n = C.c_size_t(0)
            rc = lib.gnx_jit_precompile(C.byref(L.BlockParams(*dims)), 64 * ept, C.byref(n))
            if rc:
                print(spec, "FAILED", lib.gnx_last_error().decode()[:500])
                continue
            f = [x for x in os.listdir(d) if x.endswith(".bin")][0]
            b = open(os.path.join(d, f), "rb").read()
            j = b.index(b"\n", b.index(b"\n") + 1)
            co = os.path.join(d, "k.co")
            open(co, "wb").write(b[j + 1:])
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for ln in notes.splitlines():
                ln = ln.strip()
                for key in (".name:", ".vgpr_count:", ".sgpr_count:", ".sgpr_spill_count:", ".vgpr_spill_count:",
                            ".private_segment_fixed_size:", ".group_segment_fixed_size:"):
                    if ln.startswith(key):
                        cur[key] = ln.split()[-1]
                if ln.startswith(".wavefront_size:"):
                    print(f"{spec:24s} {cur.get('.name:', '')[:28]:28s} vgpr {cur.get('.vgpr_count:')} (spill {cur.get('.vgpr_spill_count:')})  "
                          f"sgpr {cur.get('.sgpr_count:')} (spill {cur.get('.sgpr_spill_count:')})  scratch {cur.get('.private_segment_fixed_size:')} B  "
                          f"lds {cur.get('.group_segment_fixed_size:')} B")
                    cur = {}
            os.remove(os.path.join(d, f))


if __name__ == "__main__":
    main()

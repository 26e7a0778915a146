#!/bin/bash
# usage: tools/jit_disasm.sh de dn dg oe on og [ept]   -> gpurun_out/jitc/kb.s (k_block_wave ISA of that width set)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/jitc && rm -f gpurun_out/jitc/*.bin
GNX_JIT_CACHE=$PWD/gpurun_out/jitc python - "$@" <<'PY'
import sys, ctypes as C
sys.path.insert(0, '.')
import graphnets_jl_amd as gn
L = gn._lib; lib = L.load()
d = [int(v) for v in sys.argv[1:7]]; ept = int(sys.argv[7]) if len(sys.argv) > 7 else 2
n = C.c_size_t(0); rc = lib.gnx_jit_precompile(C.byref(L.BlockParams(*d)), 64 * ept, C.byref(n))
assert rc == 0, lib.gnx_last_error()
PY
cd gpurun_out/jitc
python - <<'PY'
import glob
f = glob.glob('*.bin')[0]; b = open(f, 'rb').read(); j = b.index(b'\n', b.index(b'\n') + 1); open('k.co', 'wb').write(b[j + 1:])
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d k.co > k.s
awk '/k_block_wave/{f=1} /k_graph_t/{f=0} f' k.s > kb.s
for pat in s_load v_writelane v_readlane v_fma v_pk_fma s_waitcnt ds_read ds_write global_load global_store s_cbranch; do echo "$pat $(grep -c "$pat" kb.s)"; done

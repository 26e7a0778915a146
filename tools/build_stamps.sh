#!/bin/bash
# Diagnostic build of libgnx with per-phase shader-clock stamps in k_rows_gemm: graphnets.jl_amd/libgnx_stamps.so
# (run with GNX_LIB_PATH=graphnets.jl_amd/libgnx_stamps.so GNX_WIDE_STAMPS=1; the shipped library executes no stamp)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/graphnets.jl_amd/csrc
python3 $R/graphnets.jl_amd/build.py > /dev/null
/opt/rocm/bin/hipcc -x hip -c $C/gnx_wide.hip -o /tmp/gnx_wide_stamps.o -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C -fvisibility=hidden -fno-gpu-rdc -DGNX_WIDE_STAMPS_BUILD
OBJS=$(ls $C/_obj/*.o | grep -v gnx_wide.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/graphnets.jl_amd/libgnx_stamps.so $OBJS /tmp/gnx_wide_stamps.o -ldl
echo $R/graphnets.jl_amd/libgnx_stamps.so

#!/bin/bash
# A/B build of ONE source file with extra flags: tools/build_variant.sh <tag> <file.hip> <flags...>  ->  graphnets.jl_amd/libgnx_<tag>.so
# (run with GNX_LIB_PATH=graphnets.jl_amd/libgnx_<tag>.so)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/graphnets.jl_amd/csrc
tag=$1; src=$2; shift 2
python3 $R/graphnets.jl_amd/build.py > /dev/null
/opt/rocm/bin/hipcc -x hip -c $C/$src -o /tmp/gnx_variant_$tag.o -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C -fvisibility=hidden -fno-gpu-rdc "$@"
OBJS=$(ls $C/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/graphnets.jl_amd/libgnx_$tag.so $OBJS /tmp/gnx_variant_$tag.o -ldl
echo $R/graphnets.jl_amd/libgnx_$tag.so

#!/bin/bash
# The COMMITTED (HEAD) version of one source file as a variant library, the baseline of a same-box A/B against the working tree:
#   tools/build_head_variant.sh <tag> <file.hip>   ->  graphnets.jl_amd/libgnx_<tag>.so   (every other object: the working tree's, built by build.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/graphnets.jl_amd/csrc
tag=$1; src=$2
git -C $R show HEAD:graphnets.jl_amd/csrc/$src > /tmp/gnx_head_$src
/opt/rocm/bin/hipcc -x hip -c /tmp/gnx_head_$src -o /tmp/gnx_variant_$tag.o -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C -fvisibility=hidden -fno-gpu-rdc
OBJS=$(ls $C/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/graphnets.jl_amd/libgnx_$tag.so $OBJS /tmp/gnx_variant_$tag.o -ldl
echo $R/graphnets.jl_amd/libgnx_$tag.so

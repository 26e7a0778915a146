#!/bin/bash
# A/B of several builds of libgnx on the GPU box (tools/build_variant.sh makes graphnets.jl_amd/libgnx_<tag>.so): us/step of the headline
# workloads per build, two repetitions each, interleaved.   Usage: tools/ab_libs.sh "<tag> <tag> ..." [extra bench args]  -> gpurun_out/ab_libs.log
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
LOG=$OUT/ab_libs.log; : > $LOG
tags=$1; shift
one() {  # name tag args...
  name=$1; tag=$2; shift 2
  lib=$REPO/graphnets.jl_amd/libgnx.so; [ "$tag" != base ] && lib=$REPO/graphnets.jl_amd/libgnx_$tag.so
  line=$(GNX_LIB_PATH=$lib python3 $REPO/bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | tail -1)
  echo "$name $tag $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("us/step", round(d["ms_per_step"]*1e3,2), "kernels", r.get("all_kernels_us"))' 2>&1)" | tee -a $LOG
}
for rep in 1 2; do
  for tag in base $tags; do
    one c2 $tag "$@"
    one hetero512 $tag --workload hetero "$@"
    one hetero4096 $tag --workload hetero --hetero-graphs 4096 "$@"
  done
done

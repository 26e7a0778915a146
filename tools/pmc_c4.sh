#!/bin/bash
# PMC counters of the C4 model's kernels (separate passes, kernel-trace only beside them): tools/pmc_c4.sh <tag> [ENV=..]...
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_c4_$TAG; mkdir -p $OUT
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --model c4 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_a -- $B > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_b -- $B > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_c -- $B > $OUT/c.log 2>&1
cd $REPO
python3 tools/summarize_prof.py $OUT $OUT/summary > $OUT/summary.txt 2>&1
python3 - <<PY
import json
d=json.load(open("$OUT/summary_pmc.json"))
for k,v in d.items():
    if "ffn" in k or "rows_gemm<128,true" in k:
        wc=v.get("SQ_WAVE_CYCLES",1)
        print(k, "launches", v.get("launches_sampled"))
        print("   mfma_util", round(v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1024/(v.get("GRBM_GUI_ACTIVE",1)/8),3),
              "wait_any", round(v.get("SQ_WAIT_ANY",0)/wc,3), "wait_inst", round(v.get("SQ_WAIT_INST_ANY",0)/wc,3), "active", round(v.get("SQ_ACTIVE_INST_ANY",0)/wc,3),
              "wait_inst_lds", round(v.get("SQ_WAIT_INST_LDS",0)/wc,3), "active_lds", round(v.get("SQ_ACTIVE_INST_LDS",0)/wc,3), "act_valu", round(v.get("SQ_ACTIVE_INST_VALU",0)/wc,3))
        print("   ", {x: round(y) for x,y in v.items() if x.startswith("SQ_") or x.startswith("GRBM")})
PY

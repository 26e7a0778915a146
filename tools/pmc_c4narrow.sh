#!/bin/bash
# Kernel trace + FETCH/WRITE + SQ counter passes of the README ex.3 model at its own widths, one row per kernel instantiation.
# -> gpurun_out/prof_<tag>/summary_{kernel_stats.csv,pmc.json}     Usage: tools/pmc_c4narrow.sh r02_c4narrow
TAG=${1:-r02_c4narrow}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --model c4 --core-dims 10,5,3 --steps 5 --warmup 2 --no-cpu-baseline --no-c-abi"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1
if [ "${GNX_PROF_SKIP_TRAFFIC:-0}" != "1" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
fi
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1
python3 $REPO/tools/summarize_prof.py $OUT $OUT/summary --full-names > $OUT/summary.txt 2>&1
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete; find "$OUT" -name "*_agent_info.csv" -delete
tail -80 $OUT/summary.txt

#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and, in SEPARATE passes, PMC counters of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]   → gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-c-abi $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
if [ "${GNX_PROF_EA:-0}" = "1" ]; then
# memory-side request classes of the L2s: total read requests by size, and the ones whose destination is DRAM
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$OUT/pmc_ea1" -- $BENCH > "$OUT/pmc_ea1.log" 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum --kernel-trace --output-format csv -d "$OUT/pmc_ea2" -- $BENCH > "$OUT/pmc_ea2.log" 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum --kernel-trace --output-format csv -d "$OUT/pmc_ea3" -- $BENCH > "$OUT/pmc_ea3.log" 2>&1
rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum --kernel-trace --output-format csv -d "$OUT/pmc_ea4" -- $BENCH > "$OUT/pmc_ea4.log" 2>&1
fi
if [ "${GNX_PROF_SQ:-0}" = "1" ]; then
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$OUT/pmc_sq1" -- $BENCH > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
fi
if [ "${GNX_PROF_MFMA:-0}" = "1" ]; then
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -- $BENCH > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_mix" -- $BENCH > "$OUT/pmc_mix.log" 2>&1
fi
python3 $REPO/tools/summarize_prof.py "$OUT" "$OUT/summary" ${GNX_PROF_DIMS:+--dims $GNX_PROF_DIMS} > "$OUT/summary.txt" 2>&1
# the raw traces are tens of MB per pass (bench.py's clock warm-up alone is thousands of launches): summaries stay, traces go
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete; find "$OUT" -name "*_agent_info.csv" -delete
tail -60 "$OUT/summary.txt"

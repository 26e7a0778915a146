#!/bin/bash
# One round's profile set on the GPU box: kernel-trace + FETCH/WRITE PMC passes of bench.py for the headline config (C2, README
# dims), the heterogeneous batches (C3: 512 graphs, C5: 4096 graphs) and the C4 model; summaries land in gpurun_out/prof_<tag>/.
# Usage: tools/profile_round.sh r02
set -u
R=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO" || exit 1
mkdir -p gpurun_out
GNX_PROF_EA=1 GNX_PROF_DIMS=readme bash tools/profile.sh ${R}_readme --steps 100 > gpurun_out/prof_${R}_readme.out 2>&1
GNX_PROF_DIMS=readme_hetero512 bash tools/profile.sh ${R}_hetero512 --steps 100 --workload hetero > gpurun_out/prof_${R}_hetero512.out 2>&1
GNX_PROF_DIMS=readme_hetero4096 bash tools/profile.sh ${R}_hetero4096 --steps 100 --workload hetero --hetero-graphs 4096 > gpurun_out/prof_${R}_hetero4096.out 2>&1
GNX_PROF_DIMS=readme_hetero4096_8000000 bash tools/profile.sh ${R}_hetero4096_8M --steps 40 --workload hetero --hetero-graphs 4096 --hetero-edges 8000000 > gpurun_out/prof_${R}_hetero4096_8M.out 2>&1
GNX_PROF_SQ=0 GNX_PROF_MFMA=1 GNX_PROF_DIMS=core bash tools/profile.sh ${R}_core --steps 20 --dims core > gpurun_out/prof_${R}_core.out 2>&1
# C4 model (configs[3]) and README ex.3 at its own widths: kernel trace + FETCH / WRITE passes -> whole-model traffic per forward
for M in c4 c4_10-5-3; do
  OUT=$REPO/gpurun_out/prof_${R}_$M; mkdir -p $OUT
  CD=""; [ "$M" = "c4_10-5-3" ] && CD="--core-dims 10,5,3"
  B="python3 $REPO/bench.py --model c4 --steps 3 --warmup 1 --no-cpu-baseline --no-c-abi --full-line $CD"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
   # matrix-core counters of the wide model (k_ffn_x6: bf16 MFMAs at 32 busy cycles each; k_ffn_fused / k_rows_gemm: fp32 MFMAs)
   [ "$M" = "c4" ] && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $B > $OUT/pmc_mfma.log 2>&1)
  python3 tools/summarize_prof.py $OUT $OUT/summary --model-traffic $M > $OUT/summary.txt 2>&1
  find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete; find "$OUT" -name "*_agent_info.csv" -delete
done
for t in readme hetero512 hetero4096 hetero4096_8M core c4 c4_10-5-3; do echo "== $t"; head -12 gpurun_out/prof_${R}_$t/summary_kernel_stats.csv; done
ls gpurun_out/prof_${R}_readme/ profiles/ | head -40

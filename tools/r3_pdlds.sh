#!/bin/bash
# round-3 A/B of the projected edge GEMM's destination-rows-through-LDS form (GNX_GEMM_PD_LDS=0: both gathered tables as epilogue operand streams): tests, block and C4 timing, per-phase stamps
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "wide or core or fullsize or block or model or chain or backward" > gpurun_out/pdlds_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/pdlds_tests.log
rm -f gpurun_out/ab_wide.log gpurun_out/ab_c4.log
timeout -k 10 300 bash tools/ab_wide.sh "pdlds GNX_GEMM_PD_LDS=1" "streams GNX_GEMM_PD_LDS=0" &&
timeout -k 10 300 bash tools/ab_c4.sh "pdlds GNX_GEMM_PD_LDS=1" "streams GNX_GEMM_PD_LDS=0"
for m in 1 0; do
  echo "PD_LDS=$m"
  GNX_GEMM_PD_LDS=$m GNX_LIB_PATH=graphnets.jl_amd/libgnx_stamps.so GNX_WIDE_STAMPS=1 timeout -k 10 300 python bench.py --dims core --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | grep "gnx stamps" | grep "edge" | tail -1
done > gpurun_out/stamps_pdlds.log 2>&1
cat gpurun_out/stamps_pdlds.log

#!/bin/bash
# round-4 GPU check: the whole GPU suite, smoke, then the driver's bench invocation (one GPU process at a time, steps joined so that a
# failed or killed step starts no further GPU step)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r4_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
S0=$(date +%s); timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; rc=$?; echo "bench rc=$rc wall=$(( $(date +%s) - S0 )) s"
[ $rc -eq 0 ] || { tail -20 gpurun_out/r4_bench_default.err; exit $rc; }
python tools/print_bench_line.py gpurun_out/r4_bench_default.json

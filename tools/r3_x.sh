cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/experiments/two_stream_pipeline.py 2>&1 | tail -4
for B in 0 64; do GNX_GEMM_BN=$B timeout -k 10 200 python bench.py --model c4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GNX_GEMM_BN=$B c4', d['ms_per_step'], d['kernel_us_one_forward'])"; done

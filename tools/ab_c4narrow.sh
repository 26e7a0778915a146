#!/bin/bash
# README ex.3 model (enc -> 2 x GNCore(10,5,3) -> dec, 1M-edge graph) at the wave-tile sizes the handle can be built with:
# us/step and per-kernel event times; then a rocprofv3 kernel trace of the default.   -> gpurun_out/ab_c4narrow/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab_c4narrow; mkdir -p $OUT
for cap in 128 64 256; do
  for rep in 1 2; do
    GNX_WTILE_E=$cap python3 $REPO/bench.py --model c4 --core-dims 10,5,3 --steps 50 --no-cpu-baseline 2> $OUT/err_$cap.log | tail -1 > $OUT/line_${cap}_$rep.json
    python3 -c "import json,sys; d=json.load(open('$OUT/line_${cap}_$rep.json')); print('WTILE_E=$cap rep$rep us/step', round(d['ms_per_step']*1e3,1), d.get('kernel_us_one_forward'))" | tee -a $OUT/summary.txt
  done
done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --model c4 --core-dims 10,5,3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/kt.log 2>&1)
python3 $REPO/tools/summarize_prof.py $OUT $OUT/summary > $OUT/summary_prof.txt 2>&1
head -20 $OUT/summary_kernel_stats.csv

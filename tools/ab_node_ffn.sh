#!/bin/bash
# same-box A/B of where a wide core's node FeedForward runs — WITH tools/experiments/node_ffn_ahead.patch applied (rejected, round 5): ahead of the block on the side stream, beside the edge
# kernel (the patch's default) against behind the block (GNX_NODE_FFN_BEHIND=1: the place it has); config 4 and one GNCore replay, interleaved.   -> gpurun_out/ab_node_ffn.log
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LOG=$REPO/gpurun_out/ab_node_ffn.log; : > $LOG
for rep in 1 2 3; do
  for mode in ahead behind; do
    [ $mode = behind ] && export GNX_NODE_FFN_BEHIND=1 || unset GNX_NODE_FFN_BEHIND
    line=$(python3 $REPO/bench.py --model c4 --no-cpu-baseline --no-c-abi --steps 10 2>/dev/null | tail -1)
    echo "c4 $mode $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms", d["ms_per_step"])' 2>&1)" | tee -a $LOG
    python3 $REPO/tools/experiments/core_replay_time.py 2>/dev/null | tail -1 | sed "s/^/core_replay $mode /" | tee -a $LOG
  done
done

#!/bin/bash
# timing-only ablations of tools/experiments/edge_ws.hip (results of the ablated builds are wrong by construction; their check is ignored)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  hipcc -O3 --offload-arch=gfx950 $v -o /tmp/edge_ws_v tools/experiments/edge_ws.hip 2>/dev/null || { echo "build failed: $v"; continue; }
  echo "== variant [$v]"
  timeout -k 5 120 /tmp/edge_ws_v 7813 | grep "k_edge_ws" | tail -2
done

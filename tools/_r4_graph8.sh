#!/bin/bash
# eager loop against the captured rollout at B = 8, with and without the side stream under the reference frames' nan_to_num
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for ns in 1 0; do
    for g in "--no-graph" "--graph"; do
      TANTE_SIDE_STREAM=$ns timeout -k 10 200 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-workloads --no-roofline $g 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side_stream=$ns $g', 'frames/s', d['value'], 'ms', d['ms_per_step'], d['config']['hip_graph'])"
    done
  done
done

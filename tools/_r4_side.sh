#!/bin/bash
# the reference frames' nan_to_num on a second stream under the rollout (TANTE_SIDE_STREAM=1) or on the main stream (=0, the default), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
for b in 8 4 1; do
  for i in 1 2 3; do
    for ns in 1 0; do
      g="--no-graph"; [ $b -lt 8 ] && g="--graph"
      TANTE_SIDE_STREAM=$ns timeout -k 10 200 python $R/bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-workloads --no-roofline $g 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b side_stream=$ns $g', 'frames/s', d['value'], 'ms', d['ms_per_step'])"
    done
  done
done

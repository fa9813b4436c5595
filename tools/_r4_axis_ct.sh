#!/bin/bash
# same-box A/B of the H+W propagator's channel-tile width on the rollout bench
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for ct in 0 16; do
    TANTE_AXIS_CT=$ct timeout -k 10 200 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-train --no-workloads 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ct=$ct', 'frames/s', d['value'], 'ms', d['ms_per_step'])"
  done
done

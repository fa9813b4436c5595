#!/bin/bash
# same-box A/B of the half-size block workgroups at small batches (cfg2 rollout)
R=${GRAFT_REPO_ROOT:-/root/repo}
for b in 1 2 4 6; do
  for i in 1 2; do
    for h in 0 1; do
      TANTE_FS_HALF=$h timeout -k 10 200 python $R/bench.py --batch $b --steps 12 --warmup 3 --no-cpu-baseline --no-train --no-workloads --no-roofline --graph 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b half=$h', 'frames/s', d['value'], 'ms', d['ms_per_step'])"
    done
  done
done

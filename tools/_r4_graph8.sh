#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for g in "" "--graph"; do
    timeout -k 10 200 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-workloads --no-roofline $g 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 graph=[$g]', 'frames/s', d['value'], 'ms', d['ms_per_step'])"
  done
done

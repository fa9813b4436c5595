#!/bin/bash
# does the timed region start before the clocks have settled?  warm-up 3 (the default) against 30 rollouts, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for w in 3 30; do
    timeout -k 10 200 python $R/bench.py --steps 10 --warmup $w --no-cpu-baseline --no-train --no-workloads --no-roofline --no-graph 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup=$w', 'frames/s', d['value'], 'ms', d['ms_per_step'])"
  done
done
for k in 10 40; do
  timeout -k 10 200 python $R/bench.py --steps $k --warmup 3 --no-cpu-baseline --no-train --no-workloads --no-roofline --no-graph 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps=$k warmup=3', 'frames/s', d['value'], 'ms', d['ms_per_step'])"
done

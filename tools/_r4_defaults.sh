#!/bin/bash
# the round's default forms against the forms they replaced, same box (after the side stream went): fused tail, T propagator inside the block launch
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in "" "TANTE_HEAD_ENC=0" "TANTE_FUSE_TPROP=0"; do
    env $v timeout -k 10 200 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-workloads --no-roofline --no-graph 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', 'frames/s', d['value'], 'ms', d['ms_per_step'])"
  done
done

#!/bin/bash
# same-box sweep of the weight-gradient job launch's workgroup count on the cfg3 train step
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for w in 512 384 768 1024; do
    TANTE_WGRAD_JOBS_WGS=$w timeout -k 10 300 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-workloads --no-graph --train-steps 10 --no-train-strong 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('jobs_wgs=$w', 'train ms', d['train']['ms_per_step'])"
  done
done

#!/bin/bash
# Same-box A/B of a HOST option on the cfg3 train step:  gpurun -- 'bash tools/ab_train_opt.sh TANTE_TRAIN_FUSED_TAIL 1 0'
# (interleaved rounds; prints the B = 8 step and the B = 64 step per value)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
opt=$1; shift
for i in $(seq ${AB_ROUNDS:-3}); do
  for v in "$@"; do
    env $opt=$v timeout -k 10 200 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --no-roofline --reps 1 ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('$opt=$v', 'train ms', t['ms_per_step'], 'strong ms', (t.get('strong') or {}).get('ms_per_step'))"
  done
done

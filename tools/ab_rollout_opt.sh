#!/bin/bash
# Same-box A/B of an option on the cfg2 rollout bench:  gpurun -- 'bash tools/ab_rollout_opt.sh TANTE_TAIL_INFER 0 1'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
opt=$1; shift
for i in $(seq ${AB_ROUNDS:-3}); do
  for v in "$@"; do
    env $opt=$v timeout -k 10 200 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-train --no-workloads ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; o=(r.get('others') or {})
he=[v for k,v in o.items() if 'head' in k or 'tail' in k]
print('$opt=$v', 'frames/s', d['value'], 'ms', d['ms_per_step'], 'block us', r.get('avg_launch_us'), 'head', [h.get('avg_launch_us') for h in he])"
  done
done

#!/bin/bash
# Same-box A/B of library variants on the cfg2 ROLLOUT (the headline):  gpurun -- 'bash tools/ab_rollout.sh product <variant> ...'
# (tools/_ab/lib_<variant>.so, built by tools/build_variant.sh; "product" = the shipped library).  Interleaved rounds.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq ${AB_ROUNDS:-2}); do
  for v in "$@"; do
    if [ "$v" == "product" ]; then lib=$R/tante_amd/lib/libtante_hip.so; else lib=$R/tools/_ab/lib_$v.so; fi
    [ -f "$lib" ] || { echo "missing $lib" >&2; exit 1; }
    env TANTE_LIB=$lib timeout -k 10 200 python $R/bench.py --steps 10 --warmup 3 --reps 5 --no-cpu-baseline --no-train --no-workloads 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'frames/s', d['value'], 'ms', d['ms_per_step'], 'block us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
  done
done

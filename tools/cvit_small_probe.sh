#!/bin/bash
# cfg4 at B = 1 with the block tails as three gemm_small launches (TANTE_CVIT_SMALL_ROWS=512) or as the chain (0): graph-replayed bench
# line + rocprofv3 kernel averages.   gpurun -- 'bash tools/cvit_small_probe.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 512; do
  for r in 1 2; do
    TANTE_CVIT_SMALL_ROWS=$v timeout -k 10 200 python3 $R/bench.py --config $R/configs/cvit_rb.yaml --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --graph 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SMALL_ROWS=$v graph', d['value'], 'frames/s', d['ms_per_step'], 'ms')"
  done
  rm -rf /tmp/cvp; TANTE_CVIT_SMALL_ROWS=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cvp -- python3 $R/bench.py --config $R/configs/cvit_rb.yaml --steps 10 --warmup 3 --reps 2 --no-cpu-baseline --no-roofline --graph > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/cvp/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print('   ', r['Name'].replace('(anonymous namespace)::','')[:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
done

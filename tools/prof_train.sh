#!/bin/bash
# Kernel-time summary of the cfg3 train step:  gpurun -- 'bash tools/prof_train.sh'  -> gpurun_out/prof_train/*kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 3 --no-train-strong > $OUT/bench.json 2> $OUT/err.log
f=$(ls -t $(find $OUT -name "*kernel_stats.csv") | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:26]:
    print(f"{100*int(r['TotalDurationNs'])/tot:5.1f}% {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:110]}")
PY

#!/bin/bash
# Same-box A/B of library variants on the cfg3 TRAIN STEP (B = 8 weak, and the strong B = 64 line):
#   gpurun -- 'bash tools/ab_train.sh product bf_v000 ...'   (tools/_ab/lib_<name>.so each, built by tools/build_variant.sh; "product" = the shipped
# library; "off" = the shipped library with the one-launch block backward switched off).  Interleaved rounds: box-to-box spread exceeds most changes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${AB_ROUNDS:-2}
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    extra=""
    if [ "$v" == "product" ]; then lib=$R/tante_amd/lib/libtante_hip.so
    elif [ "$v" == "off" ]; then lib=$R/tante_amd/lib/libtante_hip.so; extra="TANTE_TRAIN_FUSED_BLOCK_BWD=0"
    else lib=$R/tools/_ab/lib_$v.so; fi
    [ -f "$lib" ] || { echo "missing $lib" >&2; exit 1; }
    env $extra TANTE_LIB=$lib timeout -k 10 200 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --no-roofline --reps 1 ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('$v', 'train ms', t['ms_per_step'], 'strong ms', (t.get('strong') or {}).get('ms_per_step'))"
  done
done

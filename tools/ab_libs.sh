#!/bin/bash
# Same-box A/B of several builds on the rollout bench:  gpurun -- 'bash tools/ab_libs.sh base ring ...'  (tools/_ab/lib_<name>.so each, built by
# tools/build_variant.sh; interleaved rounds in one call because box-to-box spread (~5 %) exceeds most kernel changes).  The variant is
# selected through TANTE_LIB (tante_amd/_lib.py): the product library tante_amd/lib/libtante_hip.so is never overwritten.  "product" names it.
set -u
# AB_ARGS: bench.py arguments of the runs (default: the cfg2 rollout leg alone)
AB_ARGS=${AB_ARGS:---steps 8 --warmup 3 --no-cpu-baseline --no-train --no-workloads}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" == "product" ]; then lib=$R/tante_amd/lib/libtante_hip.so; else lib=$R/tools/_ab/lib_$v.so; fi
    [ -f "$lib" ] || { echo "missing $lib" >&2; exit 1; }
    TANTE_LIB=$lib timeout -k 10 200 python $R/bench.py $AB_ARGS 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$v', 'frames/s', d['value'], 'ms', d['ms_per_step'], 'dominant kernel us', r.get('avg_launch_us'), 'frac', r.get('frac'))"
  done
done

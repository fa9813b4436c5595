#!/bin/bash
# Same-box A/B of several builds on the rollout bench:  gpurun -- 'bash tools/ab_libs.sh base ring ...'  (tools/_ab/lib_<name>.so each;
# interleaved rounds in one call because box-to-box spread (~5 %) exceeds most kernel changes).  Restores the product library at the end.
R=$GRAFT_REPO_ROOT
cp $R/tante_amd/lib/libtante_hip.so /tmp/lib_product.so
for i in 1 2 3; do
  for v in "$@"; do
    cp $R/tools/_ab/lib_$v.so $R/tante_amd/lib/libtante_hip.so
    timeout -k 10 200 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'frames/s', d['value'], 'ms', d['ms_per_step'], 'block us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
  done
done
cp /tmp/lib_product.so $R/tante_amd/lib/libtante_hip.so

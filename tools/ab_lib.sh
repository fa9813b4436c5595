#!/bin/bash
# Same-box A/B of two builds of libtante_hip.so on the rollout bench (box-to-box variance is ~5 %, larger than most kernel tweaks).
# Put the two libraries at tools/_ab/lib_base.so and tools/_ab/lib_prio.so (git-ignored, but they travel to the GPU box), then
#   gpurun -- 'bash tools/ab_lib.sh'
# and restore tante_amd/lib/libtante_hip.so with `python -m tante_amd.build` afterwards.
for i in 1 2 3; do
  for v in base prio; do
    cp tools/_ab/lib_$v.so tante_amd/lib/libtante_hip.so
    timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-train > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
"
  done
done

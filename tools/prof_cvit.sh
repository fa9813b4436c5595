#!/bin/bash
# On the GPU box: kernel trace of the cfg4 forward at B = $1 (default 1) -> gpurun_out/cvit_b$1/ and a per-forward timeline of the last step.
B=${1:-1}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cvit_b$B -- python3 $R/bench.py --config $R/configs/cvit_rb.yaml --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/cvit_b$B.log 2>&1

#!/bin/bash
# Run on the MI355X box (via gpurun): the round's rocprofv3 evidence -> gpurun_out/r01/.  Copy what should be judged into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rollout -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train > $OUT/rollout_bench.json 2> $OUT/rollout.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 3 > $OUT/train_bench.json 2> $OUT/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cvit -- python3 $R/tools/cvit_time.py > $OUT/cvit_time.txt 2> $OUT/cvit.err
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train > /dev/null 2>&1
done
cd $R && python3 tools/summarize_profiles.py $OUT

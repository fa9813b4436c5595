#!/bin/bash
# Run on the MI355X box (via gpurun): the round's rocprofv3 evidence -> gpurun_out/r06/.  tools/summarize_profiles.py condenses it into
# the files that are committed under profiles/.   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel traces (never combined with counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rollout -- python3 $R/bench.py --no-workloads --steps 5 --warmup 2 --no-cpu-baseline --no-train > $OUT/rollout_bench.json 2> $OUT/rollout.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $R/bench.py --no-workloads --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 3 --no-train-strong > $OUT/train_bench.json 2> $OUT/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trl -- python3 $R/bench.py --config $R/configs/tante_trl.yaml --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trl_bench.json 2> $OUT/trl.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cvit -- python3 $R/bench.py --config $R/configs/cvit_rb.yaml --steps 5 --warmup 2 > $OUT/cvit_bench.json 2> $OUT/cvit.err
python3 $R/bench.py --config $R/configs/cvit_rb.yaml --batch 4 --steps 10 --warmup 3 > $OUT/cvit_b4_bench.json 2> $OUT/cvit_b4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fno -- python3 $R/bench.py --config $R/configs/tante_fno.yaml --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fno_bench.json 2> $OUT/fno.err
python3 $R/bench.py --config $R/configs/fno_vf.yaml --steps 3 --warmup 1 > $OUT/fno_vf_bench.json 2> $OUT/fno_vf.err
python3 $R/tools/dp_gloo_1gpu.py > $OUT/dp_gloo_1gpu.log 2>&1
TANTE_DIST_BACKEND=gloo TANTE_ALL_ON_GPU0=1 python3 $R/bench.py --gpus 2 --steps 3 --warmup 1 --train-steps 3 > $OUT/bench_gloo2_plumbing.json 2> $OUT/bench_gloo2_plumbing.err
python3 $R/bench.py --gpus 2 --steps 1 --warmup 0 > $OUT/bench_gpus2_refused.out 2>&1; echo "exit code $?" >> $OUT/bench_gpus2_refused.out
# 2. counters of the rollout, one pass per group (8 SQ slots; FETCH_SIZE and WRITE_SIZE cannot share a pass), kernel-trace only
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/$n -- python3 $R/bench.py --no-workloads --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train > /dev/null 2> $OUT/pmc_$n.err
done
# 2b. HBM counters of cfg5's spectral path (the truncated-DFT kernels) and of the train step (tools/pmc_train.sh: per-kernel MB and TB/s)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_fno/$c -- python3 $R/bench.py --config $R/configs/tante_fno.yaml --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/pmc_fno_$c.err
done
bash $R/tools/pmc_train.sh > $OUT/pmc_train.log 2>&1
cp $R/gpurun_out/pmc_train/summary.txt $OUT/r06_pmc_train.txt 2>/dev/null
# 2c. cfg4 at B = 4 and cfg5: per-kernel HBM bytes, TB/s and matrix-pipe occupancy (tools/pmc_bench.sh)
bash $R/tools/pmc_bench.sh cvit_b4 --config configs/cvit_rb.yaml --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_cvit_b4.log 2>&1
cp $R/gpurun_out/pmc_cvit_b4/summary.txt $OUT/r06_pmc_cvit_b4.txt 2>/dev/null
bash $R/tools/pmc_bench.sh fno --config configs/tante_fno.yaml --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_fno_all.log 2>&1
cp $R/gpurun_out/pmc_fno/summary.txt $OUT/r06_pmc_fno_kernels.txt 2>/dev/null
# 3. the counters condensed, put where bench.py reads them (this box's copy of profiles/), then the un-profiled reference line: its
#    roofline.traffic then cites counters taken from the same source tree (traffic_source.stale = false)
cd $R && python3 tools/summarize_profiles.py $OUT
cp $OUT/r06_pmc_rollout.json $OUT/r06_pmc_fno.json $R/profiles/ 2>/dev/null
python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench_full.json 2> $OUT/bench_full.err
python3 $R/bench.py --config $R/configs/tante_fno.yaml --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fno_bench.json 2> $OUT/fno2.err
# (the bench lines written under the profiler above carry its per-launch host overhead: replace them by un-profiled runs of the same commands)
python3 $R/bench.py --config $R/configs/tante_trl.yaml --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trl_bench.json 2> $OUT/trl2.err
python3 $R/bench.py --config $R/configs/cvit_rb.yaml --steps 10 --warmup 3 > $OUT/cvit_bench.json 2> $OUT/cvit2.err
python3 $R/bench.py --no-workloads --steps 5 --warmup 2 --no-cpu-baseline --no-train > $OUT/rollout_bench.json 2> $OUT/rollout2.err
cd $R && python3 tools/summarize_profiles.py $OUT
# gpurun merges gpurun_out/ back only while it stays under 64 MiB: the raw rocprofv3 trees are condensed above, drop them
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
rm -rf $R/gpurun_out/pmc_train $R/gpurun_out/pmc_cvit_b4 $R/gpurun_out/pmc_fno
du -sh $R/gpurun_out

#!/bin/bash
# Hardware counters of the fused block kernel on the rollout bench (run on the GPU box via gpurun):  bash tools/fs_pmc.sh [tag]
# Separate --pmc passes (8 SQ slots per pass; FETCH_SIZE / WRITE_SIZE cannot share one), kernel-trace only -> gpurun_out/pmc_<tag>.json
R=$GRAFT_REPO_ROOT
TAG=${1:-fs}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train > /dev/null 2> $OUT/$n.err
done
cd $R && python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for short in ("block_fs_kernel<2", "block_fs_kernel<1", "fused_block16_kernel", "axis_hw_kernel"):
            if short in k:
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for short in ("block_fs_kernel<2", "block_fs_kernel<1", "fused_block16_kernel", "axis_hw_kernel"):
            if short in r["Kernel_Name"]:
                dur[short].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"avg_ns_under_pmc": round(sum(dur[k]) / max(1, len(dur[k])), 1), "counters_avg_per_launch": {c: round(v, 1) for c, v in sorted(m.items())}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = round(m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024)   # gfx950: wide reads tallied at half
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {x: round(m.get(y, 0) / wc, 3) for x, y in (("parked", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"))}
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        e["mfma_busy_over_busy_x32"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["SQ_BUSY_CYCLES"] * 32), 4)
    res[k] = e
json.dump(res, open("$OUT.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY

#!/bin/bash
# HBM traffic per kernel of the cfg3 train step (two counter passes, kernel-trace only):  gpurun -- 'bash tools/pmc_train.sh'
# -> gpurun_out/pmc_train/summary.txt : kernel, launches, avg us, MB fetched + written per launch, TB/s
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_train
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  TANTE_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --no-workloads --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --train-steps 2 --no-train-strong > /dev/null 2> $OUT/$c.err
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def load(c):
    f = max(glob.glob(f"{out}/{c}/*/*counter_collection.csv"), key=lambda p: len(open(p).read()))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0][:56]
        agg[n][0] += 1; agg[n][1] += float(r["Counter_Value"])
    return agg
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
f = max(glob.glob(f"{out}/FETCH_SIZE/*/*kernel_trace.csv"), key=lambda p: len(open(p).read()))
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0][:56]
    dur[n][0] += 1; dur[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for n, (k, us) in dur.items():
    # units as in tools/summarize_profiles.py (MI355X_MICROARCH.md, HBM): KiB; FETCH_SIZE doubled for 16 B / lane streams (an upper
    # bound for kernels that read narrower)
    fb = fe[n][1] / max(1, fe[n][0]) * 1024 * 2 / 1e6 if n in fe else 0.0
    wb = wr[n][1] / max(1, wr[n][0]) * 1024 / 1e6 if n in wr else 0.0
    rows.append((us, n, k, us / k, fb, wb))
rows.sort(reverse=True)
with open(f"{out}/summary.txt", "w") as fo:
    for us, n, k, avg, fb, wb in rows[:60]:
        line = f"{n:58s} {k:5d} {avg:8.1f} us  fetch {fb:8.1f} MB  write {wb:8.1f} MB  {(fb + wb) / avg:6.2f} TB/s  total {us / 1e3:7.2f} ms"
        print(line); fo.write(line + "\n")
PY

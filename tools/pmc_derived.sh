#!/bin/bash
# rocprofv3's DERIVED metrics per kernel of one bench.py command (one --pmc pass per pair of metrics, kernel-trace only):
#   gpurun -- 'bash tools/pmc_derived.sh <tag> <bench.py arguments...>'   ->  gpurun_out/pmcd_<tag>/summary.txt
# MfmaUtil / VALUBusy / SALUBusy (% of busy cycles the pipe is active), MemUnitStalled, LdsUtil, LdsBankConflict, TA_BUSY_avr, VmemLatency,
# LdsLatency, MeanOccupancyPerCU -- as rocprofv3 defines them for gfx950 (rocprofv3 -L).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcd_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS=()
for a in "$@"; do case $a in configs/*) ARGS+=("$R/$a");; *) ARGS+=("$a");; esac; done
i=0
for c in "MfmaUtil VALUBusy" "SALUBusy MemUnitStalled" "LdsUtil LdsBankConflict" "TA_BUSY_avr MeanOccupancyPerCU" "VmemLatency LdsLatency"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p$i -- python3 $R/bench.py "${ARGS[@]}" > /dev/null 2> $OUT/p$i.err
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def name(r):
    return re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0][:56]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for d in sorted(glob.glob(f"{out}/p*")):
    fs = glob.glob(f"{d}/*/*counter_collection.csv")
    if not fs: continue
    for r in csv.DictReader(open(max(fs, key=lambda p: len(open(p).read())))):
        a = agg[name(r)][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
f = max(glob.glob(f"{out}/p0/*/*kernel_trace.csv"), key=lambda p: len(open(p).read()))
for r in csv.DictReader(open(f)):
    x = dur[name(r)]; x[0] += 1; x[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
cols = ["MfmaUtil", "VALUBusy", "SALUBusy", "MemUnitStalled", "LdsUtil", "LdsBankConflict", "TA_BUSY_avr", "MeanOccupancyPerCU", "VmemLatency", "LdsLatency"]
rows = sorted(dur.items(), key=lambda kv: -kv[1][1])[:24]
with open(f"{out}/summary.txt", "w") as fo:
    hdr = f"{'kernel':58s} {'n':>4s} {'avg us':>8s} " + " ".join(f"{c[:12]:>12s}" for c in cols)
    print(hdr); fo.write(hdr + "\n")
    for n, (k, us) in rows:
        vals = []
        for c in cols:
            a = agg[n].get(c)
            vals.append(f"{a[1] / a[0]:12.2f}" if a and a[0] else f"{'-':>12s}")
        line = f"{n:58s} {k:4d} {us / k:8.1f} " + " ".join(vals)
        print(line); fo.write(line + "\n")
PY

#!/bin/bash
# Per-kernel HBM traffic and matrix-pipe occupancy of one bench.py command (three counter passes, kernel-trace only, never combined with
# other trace domains):   gpurun -- 'bash tools/pmc_bench.sh <tag> <bench.py arguments...>'
#   e.g.  bash tools/pmc_bench.sh cvit_b4 --config configs/cvit_rb.yaml --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
# -> gpurun_out/pmc_<tag>/summary.txt : kernel, launches, avg us, MB fetched / written per launch, TB/s, MFMA-busy share of the kernel's
#    busy cycles (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32), as in tools/summarize_profiles.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS=()
for a in "$@"; do case $a in configs/*) ARGS+=("$R/$a");; *) ARGS+=("$a");; esac; done
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p$i -- python3 $R/bench.py "${ARGS[@]}" > /dev/null 2> $OUT/p$i.err
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def name(r):
    return re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0][:60]
def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    fs = glob.glob(f"{out}/{d}/*/*counter_collection.csv")
    if not fs: return agg
    f = max(fs, key=lambda p: len(open(p).read()))
    for r in csv.DictReader(open(f)):
        a = agg[name(r)][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
fe, wr, sq = load("p0"), load("p1"), load("p2")
f = max(glob.glob(f"{out}/p0/*/*kernel_trace.csv"), key=lambda p: len(open(p).read()))
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    d = dur[name(r)]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for n, (k, us) in dur.items():
    # KiB counters; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section): an upper bound for kernels that read narrower than 16 B / lane
    fb = fe[n]["FETCH_SIZE"][1] / max(1, fe[n]["FETCH_SIZE"][0]) * 1024 * 2 / 1e6 if n in fe else 0.0
    wb = wr[n]["WRITE_SIZE"][1] / max(1, wr[n]["WRITE_SIZE"][0]) * 1024 / 1e6 if n in wr else 0.0
    busy = sq[n]["SQ_BUSY_CYCLES"][1] if n in sq else 0.0
    mf = sq[n]["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (busy * 32) if busy else float("nan")
    wc = sq[n]["SQ_WAVE_CYCLES"][1] if n in sq else 0.0
    wait = sq[n]["SQ_WAIT_ANY"][1] / wc if wc else float("nan")
    rows.append((us, n, k, us / k, fb, wb, mf, wait))
rows.sort(reverse=True)
with open(f"{out}/summary.txt", "w") as fo:
    for us, n, k, avg, fb, wb, mf, wait in rows[:40]:
        line = (f"{n:62s} {k:5d} {avg:8.1f} us  fetch {fb:8.1f} MB  write {wb:8.1f} MB  {(fb + wb) / avg:6.2f} TB/s  mfma_busy {mf:5.3f}  "
                f"wave_wait {wait:5.3f}  total {us / 1e3:7.2f} ms")
        print(line); fo.write(line + "\n")
PY

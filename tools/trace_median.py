"""Median duration per kernel (in launch order, runs of identical names) from a rocprofv3 kernel-trace csv directory."""
import csv, glob, sys
for d in sys.argv[1:]:
    fs = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not fs:
        print(d, "no trace"); continue
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
    seq = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
    print(d)
    i = 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j][0] == seq[i][0]: j += 1
        if j - i >= 20:
            ds = sorted(x[1] for x in seq[i:j])
            print(f"   {ds[len(ds) // 2] / 1e3:8.2f} us  x{j - i:<4} {seq[i][0][:90]}")
        i = j

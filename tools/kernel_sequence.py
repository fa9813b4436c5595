#!/usr/bin/env python3
"""Print ONE repetition of a kernel trace in launch order:  python tools/kernel_sequence.py <kernel_trace.csv> <anchor substring> [occurrence]
The trace's launches between two consecutive occurrences of the anchor kernel (e.g. format_input) are listed with their durations and the
idle gap in front of each."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -2
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[occ], idx[occ + 1]
prev_end = None
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    d = (e - s) / 1e3
    tot += d
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{d:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', '?'):>8s} wg {r.get('Workgroup_Size_X', '?'):>5s} lds {r.get('LDS_Block_Size', '?'):>7s}  {name[:110]}")
    prev_end = e
print(f"{b - a} launches, {tot:.1f} us of kernel time, span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us")

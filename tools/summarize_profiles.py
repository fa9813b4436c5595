"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into the small files that are committed under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def stats(sub, dst, top=25):
    f = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return
    rows = list(csv.DictReader(open(f[0])))[:top]
    with open(os.path.join(out, dst), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


stats("rollout", "r01_rollout_kernel_stats.csv")
stats("train", "r01_train_kernel_stats.csv")
stats("cvit", "r01_cvit_kernel_stats.csv")

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for short in ("fused_block16_kernel", "fused_head_kernel", "axis_hw_kernel", "axis_mlp_kernel", "gemm_kernel"):
            if short in k:
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"launches_sampled": len(next(iter(d.values()))), "counters_avg_per_launch": {c: round(v, 1) for c, v in m.items()}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        # rocprofv3 reports KiB; gfx950 tallies wide (16 B/lane) streaming reads at half their bytes (MI355X_MICROARCH.md, HBM):
        # doubled for the kernels whose reads are such streams (block, head, T-propagator, GEMM), as-is for axis_hw's 64 B segments
        fetch = m["FETCH_SIZE"] * 1024 * (1 if k == "axis_hw_kernel" else 2)
        e["hbm_bytes_per_launch"] = round(fetch + m["WRITE_SIZE"] * 1024)
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {"parked_waitcnt_barrier": round(m.get("SQ_WAIT_ANY", 0) / wc, 3),
                                 "issue_stall": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                 "issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}
    res[k] = e
json.dump(res, open(os.path.join(out, "r01_pmc_rollout.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])

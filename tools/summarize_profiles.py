"""Condense the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/r06/) into the small files committed under profiles/."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r06"


def stats(sub, dst, top=30):
    f = glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return
    rows = list(csv.DictReader(open(f[0])))[:top]
    with open(os.path.join(out, dst), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


for sub in ("rollout", "train", "trl", "cvit", "fno"):
    stats(sub, f"{TAG}_{sub}_kernel_stats.csv")

SHORT = {"block_fs_kernel<2": "fused_block_kernel", "block_fs_kernel<1": "fused_block_kernel_T_letter", "fused_head_kernel": "fused_head_kernel", "head_enc_kernel": "head_enc_kernel",
         "axis_hw_": "axis_hw_kernel", "axis_mlp_vec_kernel": "axis_mlp_vec_kernel"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k, short in SHORT.items():
            if k in r["Kernel_Name"]:
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "pmc", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k, short in SHORT.items():
            if k in r["Kernel_Name"]:
                dur[short].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {"config": "tante_am.yaml",
       "kernel_source_sha16": hashlib.sha256(open(os.path.join(ROOT, "tante_amd", "csrc", "block_sliced.hip"), "rb").read()).hexdigest()[:16],
       "kernel_sources": {n: hashlib.sha256(open(os.path.join(ROOT, "tante_amd", "csrc", n), "rb").read()).hexdigest()[:16]
                          for n in ("block_fused.hip", "block_sliced.hip", "head_enc.hip")},
       "note": "separate rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train`; FETCH_SIZE / WRITE_SIZE "
               "in KiB, FETCH_SIZE doubled for the 16 B/lane streams (MI355X_MICROARCH.md, HBM); SQ_* wave-cycle counters count quad-cycles"}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = {"launches_sampled": len(next(iter(d.values()))), "avg_ns_under_pmc": round(sum(dur[k]) / max(1, len(dur[k])), 1),
         "counters_avg_per_launch": {c: round(v, 1) for c, v in sorted(m.items())}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = round(m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024)   # every kernel here reads 16 B per lane
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {"parked_waitcnt_barrier": round(m.get("SQ_WAIT_ANY", 0) / wc, 3), "issue_stall": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                 "issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        e["mfma_busy_over_sq_busy_x32"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["SQ_BUSY_CYCLES"] * 32), 4)
    res[k] = e
json.dump(res, open(os.path.join(out, f"{TAG}_pmc_rollout.json"), "w"), indent=1)

# cfg5: HBM bytes of the spectral path per SpectralLayer call (its five kernels; one idft_rows_conv[_x3] launch per call)
SPEC = ("dft_rows_kernel", "dft_cols_kernel", "spectral_mix_kernel", "idft_cols_kernel", "idft_rows_conv_kernel", "idft_rows_conv_x3_kernel")
fno = {"config": "tante_fno.yaml", "kernel_source_sha16": hashlib.sha256(open(os.path.join(ROOT, "tante_amd", "csrc", "spectral_dft.hip"), "rb").read()).hexdigest()[:16],
       "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --config configs/tante_fno.yaml --steps 2 --warmup 1 --no-cpu-baseline "
               "--no-roofline`; KiB; FETCH_SIZE doubled (16 B / lane or 128 B / half-wave streams: an upper bound for the narrower reads)", "kernels": {}}
tot_bytes, calls = 0.0, 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "pmc_fno", c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            for k in SPEC:
                if k in r["Kernel_Name"]:
                    e = fno["kernels"].setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})
                    e[c].append(float(r["Counter_Value"]))
for k, e in fno["kernels"].items():
    fb = sum(e["FETCH_SIZE"]) * 1024 * 2
    wb = sum(e["WRITE_SIZE"]) * 1024
    n_ = max(1, len(e["FETCH_SIZE"]))
    fno["kernels"][k] = {"launches_sampled": n_, "fetch_bytes_per_launch": round(fb / n_), "write_bytes_per_launch": round(wb / max(1, len(e["WRITE_SIZE"])))}
    tot_bytes += fb / n_ * n_ + wb / max(1, len(e["WRITE_SIZE"])) * n_
    if k in ("idft_rows_conv_kernel", "idft_rows_conv_x3_kernel"):      # one of the two per SpectralLayer call (fp32 form / split-bf16 form)
        calls += n_
if calls:
    fno["spectral_layer"] = {"hbm_bytes_per_call": round(tot_bytes / calls), "calls_sampled": calls}
    json.dump(fno, open(os.path.join(out, f"{TAG}_pmc_fno.json"), "w"), indent=1)
for name in ("rollout_bench.json", "train_bench.json", "trl_bench.json", "cvit_bench.json", "cvit_b4_bench.json", "fno_bench.json", "fno_vf_bench.json", "bench_gloo2_plumbing.json",
             "bench_full.json"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        lines = [ln for ln in open(p).read().strip().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(out, f"{TAG}_{name}"), "w").write(lines[-1] + "\n")
for name in ("dp_gloo_1gpu.log", "bench_gpus2_refused.out"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        keep = [ln for ln in open(p).read().splitlines() if ln.startswith("dp_gloo_1gpu") or "bench.py" in ln or ln.startswith("exit code")]
        open(os.path.join(out, f"{TAG}_{name}"), "w").write("\n".join(keep) + "\n")
print(json.dumps({k: (v if not isinstance(v, dict) or "counters_avg_per_launch" not in v else {kk: vv for kk, vv in v.items() if kk != "counters_avg_per_launch"})
                  for k, v in res.items()}, indent=1)[:2500])

#!/usr/bin/env python3
"""CPU-baseline protocol, steps 1 and 2 -- BUILD CONTAINER ONLY (needs /root/reference).

Step 1: time the REAL reference (zwu88/TANTE, imported with the inert-stub recipe of tests/golden/make_golden.py) on CPU for the
        headline workload: configs/tante_am.yaml = cfg2, order-3 Taylor, B = 8, fp32, eval, 1 warm-up + 3 repetitions.
Step 2: time the oracle (oracle/tante_oracle.py, what bench.py's cpu_baseline leg runs on the GPU box) on the SAME weights and
        inputs, check that the two agree numerically and run within +-10 % of each other.
Writes profiles/cpu_reference.json.  Nothing of the reference is copied: it is imported by path and only timed."""
import json
import os
import platform
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TANTE_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not found: this tool runs in the build container only")
    for name in ("torchinfo", "h5py", "wandb"):
        m = types.ModuleType(name)
        if name == "torchinfo":
            m.summary = lambda *a, **k: None
        sys.modules.setdefault(name, m)
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    sys.path.insert(0, REF)
    from models.tante import TANTE                      # the reference
    from data.dataset import TanteMetadata
    import tante_amd
    from oracle import tante_oracle as O

    threads = int(os.environ.get("TANTE_CPU_THREADS", str(len(os.sched_getaffinity(0)))))
    torch.set_num_threads(threads)
    cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_am.yaml"))
    wl, mk = cfg["workload"], cfg["model"]
    B, T, D, res = wl["batch_size"], wl["n_steps_input"], wl["n_fields"], tuple(wl["spatial_resolution"])
    md = TanteMetadata(dataset_name="synthetic", n_spatial_dims=2, spatial_resolution=res, field_names={0: [f"f{i}" for i in range(D)]},
                       boundary_condition_types=["periodic"], n_files=1, n_trajectories_per_file=[1], n_steps_per_trajectory=[16], n_fields=D)
    torch.manual_seed(cfg.get("seed", 211))
    kw = {k: v for k, v in mk.items() if k != "_target_"}
    kw["dropout"] = 0.0
    ref = TANTE(dset_metadata=md, **kw).eval()
    x = torch.randn(B, T, D, *res, generator=torch.Generator().manual_seed(211))
    w = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    ocfg = O.TanteCfg(mk["in_T"], D, res, taylor_order=mk.get("taylor_order", 1), frame_interval=mk.get("frame_interval", 1.0),
                      attn_axes=mk.get("attn_axes", "THWTHWTHW"), n_head=mk.get("n_head", 8), mlp_ratio=mk.get("mlp_ratio", 1.0),
                      embed_dim=mk.get("embed_dim", 256), patch_scale=mk.get("patch_scale", 32))

    def timed(fn, reps=3):
        with torch.no_grad():
            y = fn()                                    # warm-up
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                y = fn()
                ts.append(time.perf_counter() - t0)
        return y, ts

    y_ref, t_ref = timed(lambda: ref(x))
    O.set_fast(True)                                    # the form bench.py's cpu_baseline leg times
    y_orc, t_orc = timed(lambda: O.tante_forward(w, ocfg, x))
    O.set_fast(False)
    y_exp, t_exp = timed(lambda: O.tante_forward(w, ocfg, x), reps=1)
    err = max(float((y_orc - y_ref).abs().max() / y_ref.abs().max()), float((y_exp - y_ref).abs().max() / y_ref.abs().max()))
    m_ref, m_orc = sum(t_ref) / len(t_ref), sum(t_orc) / len(t_orc)
    out = {"workload": "configs/tante_am.yaml (cfg2: 256x256x11, order-3 Taylor THW-THW-THW, fp32, eval)", "batch": B,
           "cpu_model": cpu_model(), "threads": threads, "torch": torch.__version__,
           "reference": {"s_per_call": [round(t, 4) for t in t_ref], "mean_s": round(m_ref, 4), "frames_per_s": round(B / m_ref, 3)},
           "oracle": {"s_per_call": [round(t, 4) for t in t_orc], "mean_s": round(m_orc, 4), "frames_per_s": round(B / m_orc, 3)},
           "oracle_written_out_form_s": round(t_exp[0], 4),
           "oracle_over_reference_time": round(m_orc / m_ref, 4), "within_10_percent": bool(abs(m_orc / m_ref - 1) <= 0.10),
           "max_abs_diff_over_max": err}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "cpu_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

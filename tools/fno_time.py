"""Time the cfg5 spectral model (configs/tante_fno.yaml) forward in bf16 / fp32."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_fno.yaml")); wl = cfg["workload"]
H, W = wl["spatial_resolution"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=(H, W))
torch.manual_seed(211)
for mode in ("bf16", "fp32"):
    m = tante_amd.build_model(cfg, md).cuda().eval().set_compute(mode)
    for B in (1, 4):
        x = torch.randn(B, wl["n_steps_input"], wl["n_fields"], H, W, device="cuda")
        with torch.no_grad():
            for _ in range(2): m(x)
            torch.cuda.synchronize(); t = time.time()
            for _ in range(5): m(x)
            torch.cuda.synchronize()
        print(mode, "B", B, "ms/forward", (time.time() - t) / 5 * 1e3, flush=True)

"""Host-side cost of one cfg3 train step: cProfile over 5 steps (the GPU queue absorbs launches, so wall time of the un-synchronised loop
is the CPU's cost of issuing a step) + the top functions by cumulative time."""
import cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
dev = torch.device("cuda:0")
tcfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_trl.yaml")); twl = tcfg["workload"]
tmd = tante_amd.TanteMetadata(n_fields=twl["n_fields"], spatial_resolution=tuple(twl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(tcfg, tmd, dropout=float(tcfg["model"].get("dropout", 0.0))).to(dev).train().set_compute("bf16")
oc = tcfg["optimizer"]
opt = tante_amd.FlatAdamW(m.parameters(), lr=oc["lr"], weight_decay=oc["weight_decay"], max_norm=1.0)
B, n = twl["batch_size"], twl["n_steps_output"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, twl["n_steps_input"], *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(tmd)
for _ in range(2): tante_amd.train_step(m, opt, batch, fmt, n, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): tante_amd.train_step(m, opt, batch, fmt, n, 1)
t_issue = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print(f"issue {t_issue * 1e3:.1f} ms/step, complete {t_all * 1e3:.1f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tante_amd.train_step(m, opt, batch, fmt, n, 1)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)

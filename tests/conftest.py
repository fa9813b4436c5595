import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._rccl_child = None


def pytest_sessionstart(session):
    """The RCCL world-size-1 check (tests/rccl_world1_child.py) needs a process that initialises its process group BEFORE any other GPU
    call, and a process that has initialised the GPU must not start programs on this pool: so the child is started HERE, before this
    process has touched the GPU (torch.cuda.device_count() does not initialise it), when the selection includes the gpu tests; it runs
    beside the first tests and test_rccl_world1_all_reduce_beside_the_captured_graph collects its verdict."""
    import subprocess
    cfg = session.config
    expr = (cfg.getoption("markexpr") or "").strip()
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("TANTE_NO_RCCL_CHILD"):
        return
    try:
        if torch.cuda.device_count() < 1:
            return
    except Exception:      # noqa: BLE001
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    log = open(os.path.join(out_dir, "r06_rccl_world1.log"), "w")
    cfg._rccl_child = (subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py")], stdout=log,
                                        stderr=subprocess.STDOUT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")), log.name)


def load_golden(name):
    """-> (arrays dict of torch tensors, weights dict keyed without the 'w.' prefix)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs = {k: torch.from_numpy(z[k]) for k in z.files}
    return arrs


def split_prefix(arrs, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in arrs.items() if k.startswith(prefix)}


def rel_err(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b):
    """max |a-b| / max|b| -- a scale-relative max-norm error."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


G14_KW = dict(in_T=4, taylor_order=1, attn_axes="THWTHWTHW", n_head=8, embed_dim=256, patch_scale=8, dropout=0.0)
G14_FIELDS, G14_RES = 4, (64, 384)


def g14_setup():
    """The production-shape train-step fixture (tests/golden/make_golden.py:g14) stores no weights and no inputs -- 4.2 M parameters
    and 6 MB of fields do not fit a small fixture -- only their checksums: both are rebuilt here from the same seeded CPU generators
    the generator script used with the reference (manual_seed(14) before the constructor; Generator(1414) for the fields), and the
    checksums prove the rebuilt tensors are the reference's.  -> (model on CPU, batch, golden arrays)"""
    import tante_amd
    g = load_golden_raw("g14_trainstep_wide")
    torch.manual_seed(14)
    md = tante_amd.TanteMetadata(n_fields=G14_FIELDS, spatial_resolution=G14_RES)
    m = tante_amd.TANTE(dset_metadata=md, **G14_KW)
    gen = torch.Generator().manual_seed(1414)
    batch = {"input": torch.randn(2, 4, 64, 384, 4, generator=gen), "output": torch.randn(2, 4, 64, 384, 4, generator=gen)}
    names = [str(n) for n in g["param_names"]]
    params = dict(m.named_parameters())
    assert names == list(params.keys()), "parameter order differs from the reference's named_parameters()"
    wn = np.array([float(params[n].detach().double().norm()) for n in names])
    assert np.allclose(wn, g["w_norm"], rtol=1e-6, atol=1e-12), "same-seed initialisation differs from the reference's"
    chk = np.array([float(batch["input"].double().sum()), float(batch["input"].double().pow(2).sum()),
                    float(batch["output"].double().sum()), float(batch["output"].double().pow(2).sum())])
    assert np.allclose(chk, g["in_sum"], rtol=1e-9), "seeded input fields differ from the generator script's"
    return m, batch, g, names


def load_golden_raw(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


# ---- parity report: every comparison of the GPU suite leaves {test, rel, max, tol} behind -----------------------------------------
PARITY = []


def record_parity(rel, mx, tol, mode, note=""):
    """Called by the parity tests' `close()`: remembers the measured errors next to the bar they were held to, so the headroom of
    every (possibly widened) tolerance is visible after a green run (parity_report.json, also copied under gpurun_out/)."""
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    PARITY.append({"test": test, "mode": mode, "rel": float(rel), "max": float(mx), "tol": float(tol), "note": note})


def pytest_sessionfinish(session, exitstatus):
    if not PARITY:
        return
    import json
    worst = {}
    for r in PARITY:
        k = r["test"]
        if k not in worst or r["rel"] / r["tol"] > worst[k]["rel"] / worst[k]["tol"]:
            worst[k] = r
    rows = sorted(worst.values(), key=lambda r: -r["rel"] / r["tol"])
    out = {"n_comparisons": len(PARITY), "n_tests": len(rows), "exitstatus": int(exitstatus),
           "worst_margin_first": [dict(r, margin=round(r["tol"] / max(r["rel"], 1e-30), 2)) for r in rows]}
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "parity_report.json"), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

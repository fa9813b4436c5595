"""Time tante_axis_wgrad at cfg3's three propagator shapes:  python tools/axis_wgrad_time.py   (TANTE_AXIS_WGRAD_WGS=n to vary the grid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tante_amd  # noqa
from tante_amd import kernels as K, _lib as L


WS = None


def axis_wgrad(U, V, outer, n, inner, dW, db, acc):
    global WS
    if os.environ.get("NO_WS"):
        L.check(L.lib().tante_axis_wgrad(U.data_ptr(), V.data_ptr(), outer, n, inner, dW.data_ptr(), db.data_ptr(), int(acc), K._stream()), 'tante_axis_wgrad')
        return
    if WS is None:
        WS = torch.zeros(L.lib().tante_axis_wgrad_workspace_bytes() // 4, device=U.device)
    L.check(L.lib().tante_axis_wgrad_ws(U.data_ptr(), V.data_ptr(), outer, n, inner, dW.data_ptr(), db.data_ptr(), int(acc), WS.data_ptr(), WS.numel() * 4,
                                        K._stream()), 'tante_axis_wgrad')


dev = torch.device("cuda:0")
for name, outer, n, inner in (("H", 32, 16, 48 * 256), ("W", 32 * 16, 48, 256), ("T", 8, 4, 16 * 48 * 256)):
    U = torch.randn(outer, n, inner, device=dev); V = torch.randn(outer, n, inner, device=dev)
    dW = torch.zeros(n, n, device=dev); db = torch.zeros(n, device=dev)
    for _ in range(3): axis_wgrad(U, V, outer, n, inner, dW, db, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): axis_wgrad(U, V, outer, n, inner, dW, db, True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    dW.zero_(); db.zero_()
    axis_wgrad(U, V, outer, n, inner, dW, db, False)
    ref = torch.einsum("oai,oji->aj", U.double(), V.double())
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    errb = float((db.double() - U.double().sum((0, 2))).abs().max() / U.double().sum((0, 2)).abs().max())
    wsz = float(WS.abs().max()) if WS is not None else 0.0
    print(f"axis {name}: n = {n:2d}  {us:6.1f} us  {2 * U.numel() * 4 / us / 1e6:5.2f} TB/s   rel err dW {err:.1e} db {errb:.1e}  workspace max {wsz}")

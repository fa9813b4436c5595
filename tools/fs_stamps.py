#!/usr/bin/env python3
"""Where the feature-sliced block kernel (block_sliced.hip) spends its cycles: phase SHARES from in-kernel s_memtime stamps.

Needs the diagnostic library (python -m tante_amd.build --ablate -> tools/_ab/libtante_ablate.so, built with -DTANTE_ABLATE: the only build
in which a stamp executes).  Run on the GPU box:  python tools/fs_stamps.py [L] [causal].  Read the shares, not the absolute time: the
stamps' own waits forbid overlaps the product kernel has (cdna_hip_programming.md 7, In-kernel stamps)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libtante_ablate.so")
import tante_amd  # noqa: E402
from tante_amd import kernels as K  # noqa: E402

NAMES = ["token table", "table barrier", "LN1 (loads+stats+write)", "barrier 1", "q gemm", "k gemm", "v gemm", "attention", "residual loads",
         "barrier 2", "out-proj gemm", "LN2 stats + barrier 3", "LN2 normalise + barrier 4", "fc1 gemm", "GELU + barrier 5", "fc2 gemm"]


def main():
    Lq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    causal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
    ntok = 8 * 4 * 32 * 32
    x = torch.randn(ntok // Lq, Lq, 256, device=dev)
    seq = K.dense_seq(ntok // Lq, Lq)
    st = blk._packed_fused()
    lib = L.lib()
    groups = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    L.set_option("TANTE_FS_GROUPS", groups)
    nwg = 4096
    stamps = torch.zeros(nwg * 8 * 20, dtype=torch.int64, device=dev)
    lib.tante_fs_set_stamps.argtypes = [C.c_void_p]
    y = x.clone().view(-1, 256)
    for _ in range(3):
        K.block_fused(y, st, 256, 8, 256, seq, causal, 1e-5)          # warm-up without stamps
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.block_fused(y, st, 256, 8, 256, seq, causal, 1e-5)
    e1.record()
    torch.cuda.synchronize()
    print(f"un-stamped: {e0.elapsed_time(e1) * 100:.1f} us per launch")
    lib.tante_fs_set_stamps(stamps.data_ptr())
    K.block_fused(y, st, 256, 8, 256, seq, causal, 1e-5)
    torch.cuda.synchronize()
    lib.tante_fs_set_stamps(None)
    raw = stamps.cpu().numpy().reshape(nwg * 8, 20)
    ran = raw[:, 0] != 0
    wg = np.repeat(np.arange(nwg), 8)[ran]
    raw = raw[ran].astype(np.int64)
    hw, xcc = raw[:, 16], raw[:, 17] & 0xf
    cu = (xcc << 8) | ((hw >> 8) & 0xff)          # XCC | SE_ID SH_ID CU_ID
    simd = (hw >> 4) & 3
    clk = (raw[:, 15] - raw[:, 0]) / np.maximum(1, raw[:, 19] - raw[:, 18]) * 100e6
    print(f"shader clock (s_memtime / s_memrealtime x 100 MHz): median {np.median(clk) / 1e9:.3f} GHz")
    by_cu = {}
    for c, g in zip(cu, wg):
        by_cu.setdefault(int(c), set()).add(int(g))
    sizes = np.array([len(v) for v in by_cu.values()])
    pairs = [sorted(v) for v in by_cu.values() if len(v) == 2]
    print(f"{len(by_cu)} CUs in use, workgroups per CU: min {sizes.min()} max {sizes.max()}; sample pairs {pairs[:6]}; "
          f"pair index distance histogram {np.unique([b - a for a, b in pairs], return_counts=True)}")
    s = raw[:, :16]
    d = np.diff(s, axis=1)                                            # (wave, 15)
    tot = s[:, 15] - s[:, 0]
    print(f"{s.shape[0]} waves; stamped span per wave (entry -> fc2 gemm done): median {np.median(tot):.0f} cycles, max {tot.max()}")
    print(f"first entry -> last stamp over the whole grid: {s[:, 15].max() - s[:, 0].min()} cycles; entry spread {s[:, 0].max() - s[:, 0].min()}")
    for k in range(15):
        print(f"  {NAMES[k + 1]:<28s} median {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}   share {np.median(d[:, k]) / np.median(tot) * 100:5.1f} %")
    if groups == 2 or (groups == 0 and ntok // 64 > 256):
        # paired form: group = virtual block & 1; group 1 runs one barrier segment behind group 0
        g1 = (wg & 1) == 1
        print(f"paired form: span median group0 {np.median(tot[~g1]):.0f} group1 {np.median(tot[g1]):.0f}; "
              f"group-0 entry -> group-1 last stamp (per CU pair, median): {np.median(s[g1, 15]) - np.median(s[~g1, 0]):.0f}")
        for k in range(15):
            print(f"  {NAMES[k + 1]:<28s} g0 {np.median(d[~g1, k]):8.0f}   g1 {np.median(d[g1, k]):8.0f}   "
                  f"start offset g1-g0 {np.median(s[g1, k]) - np.median(s[~g1, k]):8.0f}")
        return
    # the two workgroups of a CU sit in hardware wave slots of different parity (HW_ID bits 3:0); the product build gives the odd ones
    # issue priority (FS_PRIO): how far apart do the two run?
    odd = (hw & 1) == 1
    if odd.any() and (~odd).any():
        print(f"wave slot parity: {odd.sum()} odd / {(~odd).sum()} even; span median odd {np.median(tot[odd]):.0f} even {np.median(tot[~odd]):.0f}")
        for k in range(15):
            print(f"  {NAMES[k + 1]:<28s} odd {np.median(d[odd, k]):8.0f}   even {np.median(d[~odd, k]):8.0f}   "
                  f"start offset odd-even {np.median(s[odd, k]) - np.median(s[~odd, k]):8.0f}")


if __name__ == "__main__":
    main()

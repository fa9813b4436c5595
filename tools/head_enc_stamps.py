#!/usr/bin/env python3
"""Where the fused rollout tail (head_enc_kernel, head_enc.hip) spends its time: in-kernel s_memtime stamps (100 MHz ticks).
Needs the diagnostic library (python -m tante_amd.build --ablate).  Run on the GPU box:  python tools/head_enc_stamps.py [B]"""
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

NAMES = {1: "prologue: W1/W3/W1e DMA, rows of order 0"}
for o in range(3):
    b = 2 + 6 * o
    NAMES.update({b: f"o{o} staging + barrier A", b + 1: f"o{o} stage 1", b + 2: f"o{o} pack rows + vmcnt(0)", b + 3: f"o{o} barrier B",
                  b + 4: f"o{o} stages 2+3", b + 5: f"o{o} barrier C"})
NAMES.update({26: "vmcnt + barrier C'", 27: "frame sum + stores", 28: "enc stage 1", 29: "enc stage 2", 30: "fragment stores + vmcnt(0) (acknowledged)",
              31: "barrier + counter", 33: "last arriver: fragments back + stage 3 over K = 512 (taps 1..3 streamed)", 34: "last arriver: z stores"})

def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_am.yaml"))
    m = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
    xs = [torch.randn(B * 4 * 1024, 256, device=dev) for _ in range(3)]
    last = torch.randn(B, 1, 11, 256, 256, device=dev)
    out = torch.zeros(B, 1, 11, 256, 256, device=dev)
    z = torch.zeros(B, 1024, 256, device=dev)
    ph = [d.packed_head() for d in m.decoders]
    pe = m.encoder.packed_head_enc()

    def run(enc=True):
        K.head_enc_fused(xs, 1024, 4 * 1024 * 256, 256, 3 * 1024 * 256, B, 32, 32, 256, 11, ph, [1.0, 0.5, 1 / 6], out, out[0].numel(), last, 0,
                         last[0].numel(), enc_stream=pe if enc else None, z=z if enc else None)
    lib = L.lib()
    lib.tante_head_enc_set_stamps.argtypes = [C.c_void_p]
    nwg = 1024
    for enc in (True, False):
        for _ in range(3):
            run(enc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(enc)
        e1.record()
        torch.cuda.synchronize()
        print(f"enc={enc}: un-stamped {e0.elapsed_time(e1) * 50:.1f} us per launch")
    stamps = torch.zeros(nwg * 8 * 40, dtype=torch.int64, device=dev)
    lib.tante_head_enc_set_stamps(stamps.data_ptr())
    run(True)
    torch.cuda.synchronize()
    lib.tante_head_enc_set_stamps(None)
    raw = stamps.cpu().numpy().reshape(nwg * 8, 40).astype(np.int64)
    raw = raw[raw[:, 0] != 0]
    t0 = raw[:, 0].min()
    print(f"{len(raw)} waves; ticks are 10 ns (s_memtime); wave start spread {raw[:, 0].max() - t0}")
    keys = sorted(NAMES)
    prev = 0
    for k in keys:
        col = raw[:, k]
        ok = col != 0
        if not ok.any():
            continue
        pk = prev
        while pk > 0 and not ((raw[:, pk] != 0) & ok).any():
            pk -= 1
        both = ok & (raw[:, pk] != 0)
        d = (col - raw[:, pk])[both]
        print(f"  [{k:2d}] {NAMES[k]:45s} {d.mean():8.1f} ticks  (min {d.min():6d} max {d.max():6d})   reached at {(col[ok] - t0).mean():8.1f} (max {(col[ok] - t0).max()}) by {ok.sum()} waves")
        prev = k


if __name__ == "__main__":
    main()

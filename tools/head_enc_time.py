#!/usr/bin/env python3
"""Launch time of the fused rollout tail (head_enc_kernel) on cfg2's shape, with and without the re-encoding half; TANTE_LIB selects a
variant library (tools/build_variant.sh).  python tools/head_enc_time.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd  # noqa: E402
from tante_amd import kernels as K  # noqa: E402


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
    res = []
    for enc in (True, False):
        for _ in range(5):
            run(enc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run(enc)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 20)
    print(f"{os.environ.get('TANTE_LIB', 'product')}  B={B}  TILES={os.environ.get('TANTE_HEAD_TILES', '2')}: enc {res[0]:.1f} us, head only {res[1]:.1f} us (looped, back to back)")


if __name__ == "__main__":
    main()

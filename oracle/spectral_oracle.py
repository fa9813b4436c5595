"""CPU ORACLE for the spectral operator path (models/enc_dec_fno.py).  TEST INFRASTRUCTURE ONLY -- same rules as
tante_oracle.py.  Pinned against the g12_* fixtures (outputs of the reference itself)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .tante_oracle import W, Tensor, gelu_erf, real_conv2d, real_transconv2d, sub

# models/enc_dec_fno.py:39-46 -- its OWN patch_scale table (two stages), not the three-stage one of enc_dec_cnn.py
PATCH_MAP_FNO = {64: (8, 8), 32: (8, 4), 16: (4, 4), 8: (4, 2), 4: (2, 2), 2: (2, 1)}


def spectral_layer(w: W, x: Tensor, modes1: int, modes2: int) -> Tensor:
    """SpectralLayer.forward (enc_dec_fno.py:213-222): rfft2(ortho) -> low-mode complex contraction with ONE weight shared by the
    top band [:m1] and the bottom band [-m1:] (bottom written second, so it wins where they overlap, l.203-210) ->
    irfft2(s=(H, W), ortho), plus a 1x1 conv of x."""
    B, Cin, H, Wd = x.shape
    wt = torch.complex(w["weight_re"], w["weight_im"]) if "weight_re" in w else w["weight"]
    Cout = wt.shape[1]
    xf = torch.fft.rfft2(x, dim=(-2, -1), norm="ortho")
    Wf = xf.shape[-1]
    m1, m2 = min(modes1, H), min(modes2, Wf)
    yf = torch.zeros(B, Cout, H, Wf, dtype=torch.cfloat)
    if m1 > 0 and m2 > 0:
        ww = wt[:, :, :m1, :m2]
        yf[:, :, :m1, :m2] = torch.einsum("bcij,coij->boij", xf[:, :, :m1, :m2], ww)
        yf[:, :, -m1:, :m2] = torch.einsum("bcij,coij->boij", xf[:, :, -m1:, :m2], ww)
    y = torch.fft.irfft2(yf, s=(H, Wd), dim=(-2, -1), norm="ortho")
    return F.conv2d(x, w["w0.weight"], w["w0.bias"]) + y


def enc_fno(w: W, x: Tensor, patch_scale: int, overlap: float, modes) -> Tensor:
    """enc_FNO.forward (enc_dec_fno.py:258-273): spectral -> GELU -> RealConv2d(P0) -> GELU -> spectral(modes // P0) -> GELU ->
    RealConv2d(P1) -> 'b t h w c'."""
    B, T, D, H, Wd = x.shape
    P = PATCH_MAP_FNO[patch_scale]
    m1, m2 = modes
    z = x.reshape(B * T, D, H, Wd)
    z = gelu_erf(spectral_layer(sub(w, "enc_spectral_1."), z, m1, m2))
    z = gelu_erf(real_conv2d(z, w["enc_conv_1.conv.weight"], w["enc_conv_1.conv.bias"], P[0], overlap))
    z = gelu_erf(spectral_layer(sub(w, "enc_spectral_2."), z, m1 // P[0], m2 // P[0]))
    z = real_conv2d(z, w["enc_conv_2.conv.weight"], w["enc_conv_2.conv.bias"], P[1], overlap)
    return z.reshape(B, T, *z.shape[1:]).permute(0, 1, 3, 4, 2)


def dec_fno(w: W, x: Tensor, patch_scale: int, overlap: float, modes) -> Tensor:
    """dec_FNO.forward (enc_dec_fno.py:306-323)."""
    B, T, Hp, Wp, C = x.shape
    P = PATCH_MAP_FNO[patch_scale]
    m1, m2 = modes
    z = x.permute(0, 1, 4, 2, 3).reshape(B * T, C, Hp, Wp)
    z = gelu_erf(real_transconv2d(z, w["dec_conv_1.deconv.weight"], w["dec_conv_1.deconv.bias"], P[1], overlap))
    z = gelu_erf(spectral_layer(sub(w, "dec_spectral_1."), z, m1 // P[0], m2 // P[0]))
    z = gelu_erf(real_transconv2d(z, w["dec_conv_2.deconv.weight"], w["dec_conv_2.deconv.bias"], P[0], overlap))
    z = spectral_layer(sub(w, "dec_spectral_2."), z, m1, m2)
    return z.reshape(B, T, -1, z.shape[-2], z.shape[-1])


def fno_wrapper(w: W, x: Tensor, modes1: int, modes2: int, n_layers: int = 4) -> Tensor:
    """`models.FNO.forward` (models/fno.py:102-106): 'b t c h w -> b (t c) h w' -> operator -> 'b c h w -> b 1 c h w'.
    **PARITY UNPINNED** for the operator in the middle: the reference delegates it to `neuralop.models.FNO` (fno.py:4, 94-100), which is
    not vendored, not version-pinned and not installed here, and no reference test or fixture touches it.  What is restated is the
    published operator (Li et al. 2021) that tante_amd/fno.py builds: pointwise lifting MLP (GELU) -> n_layers x [spectral_layer,
    GELU on all but the last] -> pointwise projection MLP (GELU), over the pinned `spectral_layer` above.  The I/O contract around it
    IS the reference's."""
    B, T, C_, H, Wd = x.shape
    z = x.reshape(B, T * C_, H, Wd).permute(0, 2, 3, 1)
    z = F.linear(gelu_erf(F.linear(z, w["model.lifting.fc1.weight"], w["model.lifting.fc1.bias"])),
                 w["model.lifting.fc2.weight"], w["model.lifting.fc2.bias"]).permute(0, 3, 1, 2)
    for i in range(n_layers):
        z = spectral_layer(sub(w, f"model.fno_blocks.{i}."), z, modes1, modes2)
        if i + 1 < n_layers:
            z = gelu_erf(z)
    z = z.permute(0, 2, 3, 1)
    z = F.linear(gelu_erf(F.linear(z, w["model.projection.fc1.weight"], w["model.projection.fc1.bias"])),
                 w["model.projection.fc2.weight"], w["model.projection.fc2.bias"])
    return z.permute(0, 3, 1, 2).unsqueeze(1)

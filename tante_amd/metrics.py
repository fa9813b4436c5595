"""Losses / metrics of the harness (trainer/metrics.py) on the GPU.

One HIP pass (tante_metric_sums) reduces (pred, ref) to three sums per (b, t, c) over the spatial axes; every
metric class of the reference is a closed form of those, evaluated on the tiny (B, T, C) result.  `pred` may be
any strided view (e.g. the channels-last view of the channels-first rollout buffer): no permute copy is made.
Shapes and semantics follow the reference: inputs channels-last (B, T, H, W, C); Metric.forward(x, y, rt[, eps, n]).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _spatial_strides(x: torch.Tensor):
    """(B, T, *spatial, C) -> (pb, pt, ps, pc, HW) if the spatial axes collapse to one stride."""
    sp_shape, sp_stride = x.shape[2:-1], x.stride()[2:-1]
    ps = sp_stride[-1]
    for i in range(len(sp_shape) - 2, -1, -1):
        if sp_stride[i] != sp_stride[i + 1] * sp_shape[i + 1]:
            return None
    hw = 1
    for d in sp_shape:
        hw *= d
    return x.stride(0), x.stride(1), ps, x.stride(-1), hw


def metric_sums(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """-> (B, T, C, 5) = {sum (x-y)^2, sum y^2, sum y, sum (y-p)^2, sum (y-p)} over the spatial axes, p = y at the first pixel."""
    if not (x.is_cuda and y.is_cuda):
        raise RuntimeError("tante_amd metrics run on the GPU only (no CPU fallback)")
    if x.shape != y.shape or x.dtype != torch.float32 or y.dtype != torch.float32:
        raise ValueError("metrics expect two fp32 tensors of the same (B, T, ..., C) shape")
    y = y.contiguous()
    st = _spatial_strides(x)
    if st is None:
        x = x.contiguous()
        st = _spatial_strides(x)
    pb, pt, ps, pc, hw = st
    B, T, Cc = x.shape[0], x.shape[1], x.shape[-1]
    sums = torch.empty(B, T, Cc, 5, dtype=torch.float32, device=x.device)
    L.check(L.lib().tante_metric_sums(x.data_ptr(), pb, pt, ps, pc, y.data_ptr(), B, T, hw, Cc, sums.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "tante_metric_sums")
    return sums


def _n_spatial(x):
    n = 1
    for d in x.shape[2:-1]:
        n *= d
    return n


class Metric(torch.nn.Module):
    """trainer/metrics.py:18-51."""

    def forward(self, *args, **kwargs):
        assert len(args) >= 3, "At least three arguments required (x, y, rt)"
        x, y, rt = args[:3]
        eps, n = (args[3], args[4]) if len(args) >= 5 else (0.5, 2)
        loss = self.eval(x, y, **kwargs)
        if rt is not None:
            return loss.mean() + self.eval_rt(rt, eps, n)
        return loss


class MSE(Metric):
    @staticmethod
    def eval(x, y):
        return metric_sums(x, y)[..., 0] / _n_spatial(x)                          # (B, T, C), metrics.py:53-60

    @staticmethod
    def eval_rt(rt, eps=0.5, n=2.0):
        """Step-size band regulariser, metrics.py:62-80."""
        r = torch.mean(rt)
        up, down = min(1 + eps, 4), max(1 + eps, 4)
        loss = 0
        if r < up:
            loss = loss + 5e-3 * (up - r) ** n
        if r > down:
            loss = loss + 1e-1 * (r - down) ** n
        return loss


def _m2(s, n):
    """Centred second moment sum (y - mean)^2 per (b, t, c) from the SHIFTED sums, in float64 (tiny tensors), clamped at 0."""
    z2, z1 = s[..., 3].double(), s[..., 4].double()
    return (z2 - z1 * z1 / n).clamp_min(0.0)


def _nmse_from(s, n, eps, norm_mode):
    mse = s[..., 0] / n
    if norm_mode == "norm":
        norm = s[..., 1] / n
    elif norm_mode == "std":
        norm = (_m2(s, n) / (n - 1)).float()                                       # torch.std(...)**2 (unbiased)
    else:
        raise ValueError(f"Invalid norm_mode: {norm_mode}")
    return mse / (norm + eps)


class NMSE(Metric):
    @staticmethod
    def eval(x, y, eps=1e-7, norm_mode="norm"):
        return _nmse_from(metric_sums(x, y), _n_spatial(x), eps, norm_mode)       # metrics.py:82-98


class RMSE(Metric):
    @staticmethod
    def eval(x, y):
        return torch.sqrt(MSE.eval(x, y))


class NRMSE(Metric):
    @staticmethod
    def eval(x, y, eps=1e-7, norm_mode="norm"):
        return torch.sqrt(NMSE.eval(x, y, eps=eps, norm_mode=norm_mode))


class VMSE(Metric):
    @staticmethod
    def eval(x, y):
        return NMSE.eval(x, y, norm_mode="std")


class VRMSE(Metric):
    @staticmethod
    def eval(x, y):
        return NRMSE.eval(x, y, norm_mode="std")                                  # metrics.py:158-164


class L2RE(Metric):
    @staticmethod
    def eval(x, y, eps=1e-7):
        s = metric_sums(x, y)                                                     # norms over (T, H, W) per (B, C)
        return torch.sqrt(s[..., 0].sum(dim=1)) / (torch.sqrt(s[..., 1].sum(dim=1)) + eps)   # metrics.py:100-111


class NNMSE(Metric):
    @staticmethod
    def eval(x, y, eps=1e-7, norm_mode="norm"):
        s = metric_sums(x, y)
        n, Cc = _n_spatial(x), x.shape[-1]
        mse_c = (s[..., 0] / n).mean(dim=-1)                                      # mean over C of MSE
        if norm_mode == "norm":
            norm = s[..., 1].sum(dim=-1) / (n * Cc)
        elif norm_mode == "std":
            # std over (spatial, C) jointly: combine the per-channel centred moments (Chan et al.) -- mean_c = pivot_c + z1_c / n
            tot = n * Cc
            piv = y.reshape(y.shape[0], y.shape[1], -1, Cc)[:, :, 0, :].double()
            mean_c = piv + s[..., 4].double() / n
            mean = mean_c.mean(dim=-1, keepdim=True)
            norm = ((_m2(s, n) + n * (mean_c - mean) ** 2).sum(dim=-1) / (tot - 1)).float()
        else:
            raise ValueError(f"Invalid norm_mode: {norm_mode}")
        return mse_c / (norm + eps)                                               # metrics.py:114-130


def mse_mean_grad(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """d/dx of  MSE(x, y).mean()  (the train loss, trainer.py:189) as a contiguous channels-last tensor."""
    if not (x.is_cuda and y.is_cuda):
        raise RuntimeError("tante_amd metrics run on the GPU only (no CPU fallback)")
    y = y.contiguous()
    st = _spatial_strides(x)
    if st is None:
        x = x.contiguous()
        st = _spatial_strides(x)
    pb, pt, ps, pc, hw = st
    B, T, Cc = x.shape[0], x.shape[1], x.shape[-1]
    g = torch.empty_like(y)
    L.check(L.lib().tante_mse_grad(x.data_ptr(), pb, pt, ps, pc, y.data_ptr(), B, T, hw, Cc, 2.0 / x.numel(), g.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "tante_mse_grad")
    return g

"""`models.FNO` -- the baseline Fourier neural operator wrapper of the reference (models/fno.py:63-106, configs/fno.yaml:21-26).

What the reference fixes at this boundary is the SURFACE: the constructor `(in_T, dset_metadata, modes1, modes2, modes3,
hidden_channels, gradient_checkpointing)`, the attributes it sets (`dim_in = n_fields * in_T`, `dim_out = n_fields`, `n_modes`,
`n_spatial_dims`, `.model`), and the I/O contract of `forward`: `b t c h w -> b (t c) h w -> model -> b 1 c h w` (fno.py:102-106), so
that the sliding-window rollout loops re-feed it one frame per call.  The arithmetic INSIDE `.model` is `neuralop.models.FNO`
(fno.py:4, 94-100) -- a third-party package that is neither vendored nor version-pinned by the reference (absent from
requirements.txt) nor installed here: **parity unpinned** at that boundary.  `.model` is therefore the published FNO
(Li et al. 2021: pointwise lifting MLP -> 4 Fourier layers `act(spectral(x) + W x)` -> pointwise projection MLP) assembled from the
in-repo HIP `SpectralLayer` (models/enc_dec_fno.py:184-222, pinned by fixture G12) and the MFMA row GEMM; its parameter names are
this file's own, not neuralop's.  Inference and training (autograd nodes of autograd.py); no CPU fallback.

**What the reference's own forward does with the blocks.**  `NeuralOpsCheckpointWrapper.forward` (models/fno.py:48-51; tfno.py:46-49)
calls `self.optional_checkpointing(self.fno_blocks, x, layer_idx, ...)` and DISCARDS the result (no `x =`), and it skips neuralop's
positional embedding: as shipped, the reference's FNO computes `projection(lifting(x))` -- two pointwise MLPs, no Fourier layer acts on
the output.  That is evidently a bug of the wrapper, not the model the paper's table reports, so the default here is the PUBLISHED
operator (the Fourier layers are applied); `FNO(..., reference_as_written=True)` reproduces the shipped behaviour (lifting ->
projection only; the spectral layers keep their parameters and receive zero gradient, as in the reference).  Numbers quoted for
`fno.yaml` / `fno_vf.yaml` (bench.py, profiles/) are for the published operator.  Neither form can load a reference checkpoint:
the parameter names and the internals of `neuralop.models.FNO` are not available here; `load_state_dict` refuses neuralop-named
state dicts with an explicit error instead of a silent `strict=False` mismatch."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from . import kernels as K
from . import stages as S
from .attn_backbone import _PackCache, resolve_compute
from .spectral import SpectralLayer


class _ChannelMLP(nn.Module):
    """Two pointwise (1x1) layers with a GELU between them, on channels-last rows."""

    def __init__(self, cin: int, chid: int, cout: int):
        super().__init__()
        self.fc1 = nn.Linear(cin, chid)
        self.fc2 = nn.Linear(chid, cout)
        self._cache = _PackCache()

    def _packed(self, compute: int):
        return self._cache.get(compute, [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias],
                               lambda: [S.pack_linear_chunks(l.weight.detach(), l.bias.detach(), compute) for l in (self.fc1, self.fc2)])

    def run(self, rows: torch.Tensor, compute: int) -> torch.Tensor:
        p1, p2 = self._packed(compute)
        h = S.linear_chunks(rows, p1, K.act_torch_dtype(compute), L.ACT_GELU_ERF)
        return S.linear_chunks(h, p2, torch.float32)

    def run_train(self, rows: torch.Tensor, compute: int) -> torch.Tensor:
        from .autograd import ActFn, LinearFn
        h = LinearFn.apply(rows, self.fc1.weight, self.fc1.bias, None, compute, torch.float32)
        h = ActFn.apply(h, L.ACT_GELU_ERF, K.act_torch_dtype(compute))
        return LinearFn.apply(h, self.fc2.weight, self.fc2.bias, None, compute, torch.float32)


class FourierOperator2d(nn.Module):
    """(B, Cin, H, W) -> (B, Cout, H, W): lifting -> n_layers x act(SpectralLayer) (no activation after the last) -> projection."""

    def __init__(self, n_modes, in_channels: int, out_channels: int, hidden_channels: int, n_layers: int = 4,
                 lifting_ratio: int = 2, projection_ratio: int = 2, gradient_checkpointing: bool = False, skip_blocks: bool = False):
        super().__init__()
        self.skip_blocks = skip_blocks                            # models/fno.py:48-51 as written: the blocks' outputs are dropped
        if len(n_modes) != 2:
            raise NotImplementedError("the HIP spectral layer is two-dimensional (rfft2): n_spatial_dims = 3 is not built")
        self.n_modes, self.n_layers = tuple(n_modes), n_layers
        self.in_channels, self.out_channels, self.hidden_channels = in_channels, out_channels, hidden_channels
        self.gradient_checkpointing = gradient_checkpointing      # accepted for the surface; nothing is re-computed here
        self.lifting = _ChannelMLP(in_channels, lifting_ratio * hidden_channels, hidden_channels)
        self.fno_blocks = nn.ModuleList(SpectralLayer(hidden_channels, hidden_channels, n_modes[0], n_modes[1]) for _ in range(n_layers))
        self.projection = _ChannelMLP(hidden_channels, projection_ratio * hidden_channels, out_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected (B, {self.in_channels}, H, W), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("tante_amd has no CPU path: move the model and its input to the GPU")
        B, _, H, W = x.shape
        compute = resolve_compute(None)
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        rows = x.float().permute(0, 2, 3, 1).reshape(B * H * W, self.in_channels)            # layout change only
        if not train:
            rows = rows.detach().contiguous()
            z = self.lifting.run(rows, compute).view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
            for i, blk in enumerate(self.fno_blocks if not self.skip_blocks else ()):
                z = blk.run(z, L.ACT_GELU_ERF if i + 1 < self.n_layers else L.ACT_NONE, compute)
            rows = z.permute(0, 2, 3, 1).reshape(B * H * W, self.hidden_channels)
            return self.projection.run(rows, compute).view(B, H, W, -1).permute(0, 3, 1, 2)
        from .autograd import ActFn, SpectralLayerFn
        z = self.lifting.run_train(rows.contiguous(), compute).view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        for i, blk in enumerate(self.fno_blocks if not self.skip_blocks else ()):
            z = SpectralLayerFn.apply(z, blk.weight, blk.w0.weight, blk.w0.bias, blk.modes1, blk.modes2)
            if i + 1 < self.n_layers:
                z = ActFn.apply(z, L.ACT_GELU_ERF, torch.float32)
        rows = z.permute(0, 2, 3, 1).reshape(B * H * W, self.hidden_channels).contiguous()
        return self.projection.run_train(rows, compute).view(B, H, W, -1).permute(0, 3, 1, 2)


class FNO(nn.Module):
    """models/fno.py:63-106 -- same constructor, same attributes, same forward contract."""

    def __init__(self, in_T: int, dset_metadata, modes1: int = 16, modes2: int = 16, modes3: int = 16, hidden_channels: int = 64,
                 gradient_checkpointing: bool = False, reference_as_written: bool = False):
        super().__init__()
        self.reference_as_written = reference_as_written
        n_channel = dset_metadata.n_fields
        self.dim_in = n_channel * in_T
        self.dim_out = n_channel
        self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3
        self.hidden_channels = hidden_channels
        self.initialized = False
        self.n_spatial_dims = dset_metadata.n_spatial_dims if dset_metadata else 2
        self.gradient_checkpointing = gradient_checkpointing
        if self.n_spatial_dims == 2:
            self.n_modes = (modes1, modes2)
        elif self.n_spatial_dims == 3:
            self.n_modes = (modes1, modes2, modes3)
        else:
            raise ValueError(f"n_spatial_dims must be 2 or 3, got {self.n_spatial_dims}")
        self.model = FourierOperator2d(self.n_modes, self.dim_in, self.dim_out, hidden_channels,
                                       gradient_checkpointing=gradient_checkpointing, skip_blocks=reference_as_written)
        self.in_T = in_T
        self.output_length = 1

    # key fragments only neuralop.models.FNO's state dict has (its spectral weights, channel-MLP convolutions, skip connections)
    _NEURALOP_KEYS = ("fno_blocks.convs", "fno_blocks.fno_skips", "fno_blocks.channel_mlp", "lifting.fcs", "projection.fcs", "positional_embedding")

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        foreign = [k for k in state_dict if any(f in k for f in self._NEURALOP_KEYS)]
        if foreign:
            raise RuntimeError("tante_amd.FNO cannot load a neuralop.models.FNO state dict (e.g. %r): the reference delegates this model to the "
                               "third-party `neuralop` package, whose parameterisation is not reproduced here (tante_amd/fno.py, parity "
                               "unpinned); train this model with tante_amd instead" % foreign[0])
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if input.dim() != 5 or input.shape[1] * input.shape[2] != self.dim_in:
            raise ValueError(f"expected (B, {self.in_T}, {self.dim_out}, H, W), got {tuple(input.shape)}")
        B, T, C_, H, W = input.shape
        x = input.reshape(B, T * C_, H, W)               # 'b t c ... -> b (t c) ...'
        return self.model(x).unsqueeze(1)                # 'b c ... -> b 1 c ...'

"""tante_amd -- MI355X-native implementation of the TANTE Taylor-expansion rollout path.

Host code mirrors the reference's operator surface (models.TANTE, models.attn_backbone.*,
models.enc_dec_cnn.*, trainer rollout semantics, configs/*.yaml); the arithmetic lives in
libtante_hip.so (include/tante_hip.h, tante_amd/csrc/*.hip).  There is no CPU fallback.
"""
from .tante import TANTE, TanteMetadata, enc_CNN, dec_CNN, film, interprator, t_series, Patch_map  # noqa: F401
from .attn_backbone import Attn_Backbone, TransformerBlock  # noqa: F401
from .spectral import SpectralLayer, enc_FNO, dec_FNO  # noqa: F401
from .cvit import CViT  # noqa: F401
from .fno import FNO  # noqa: F401
from .rollout import (DefaultChannelsFirstFormatter, DefaultChannelsLastFormatter, rollout_model,  # noqa: F401
                      rollout_adaptive, GraphedRollout)
from .config import instantiate, load_config, build_model  # noqa: F401
from . import metrics, optim, dist  # noqa: F401
from .metrics import MSE, NMSE, RMSE, NRMSE, VMSE, VRMSE, L2RE, NNMSE  # noqa: F401
from .optim import FlatAdamW, warmup_cosine_lr  # noqa: F401
from .train import GraphedTrainStep, train_step, train_step_adaptive, train_step_cvit  # noqa: F401
from . import harness  # noqa: F401
from .options import set_option, get_option  # noqa: F401
from .harness import LinearWarmupCosineAnnealingLR, SyntheticDataModule, save_checkpoint, load_checkpoint  # noqa: F401

__all__ = ["TANTE", "TanteMetadata", "enc_CNN", "dec_CNN", "film", "interprator", "t_series", "Attn_Backbone",
           "TransformerBlock", "DefaultChannelsFirstFormatter", "DefaultChannelsLastFormatter", "rollout_model",
           "rollout_adaptive", "GraphedRollout", "instantiate", "load_config", "build_model", "CViT", "FNO", "SpectralLayer", "enc_FNO", "dec_FNO", "set_option", "get_option"]

"""Load the reference's configs/*.yaml unchanged (hydra / omegaconf are not needed: the `model:` block is
a `_target_` dotted path plus constructor kwargs -- train.py:35, eval.py:38)."""
from __future__ import annotations

import importlib
from typing import Any, Dict

import yaml

# the reference's dotted paths -> this package
_TARGET_ALIASES = {
    "models.TANTE": "tante_amd.tante.TANTE",
    "models.tante.TANTE": "tante_amd.tante.TANTE",
    "models.CViT": "tante_amd.cvit.CViT",
    "models.cvit.CViT": "tante_amd.cvit.CViT",
    "models.FNO": "tante_amd.fno.FNO",
    "models.fno.FNO": "tante_amd.fno.FNO",
    "models.enc_dec_fno.SpectralLayer": "tante_amd.spectral.SpectralLayer",
    "models.attn_backbone.Attn_Backbone": "tante_amd.attn_backbone.Attn_Backbone",
    "models.attn_backbone.TransformerBlock": "tante_amd.attn_backbone.TransformerBlock",
}


def load_config(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return yaml.safe_load(f)


def instantiate(node: Dict[str, Any], **extra):
    """hydra.utils.instantiate for a flat `_target_` node."""
    node = dict(node)
    target = node.pop("_target_")
    target = _TARGET_ALIASES.get(target, target)
    mod, _, name = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), name)
    node.update(extra)
    return cls(**node)


def build_model(cfg: Dict[str, Any], dset_metadata, **overrides):
    node = dict(cfg["model"])
    node.update(overrides)
    return instantiate(node, dset_metadata=dset_metadata)

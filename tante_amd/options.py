"""The package's A/B switches, in ONE place.

Every switch chooses between forms that compute the same function (a fused launch or the launches it replaces, a side stream or the main
stream, ...); none is needed to use the package.  There are two kinds:
  * library options -- launch heuristics of libtante_hip.so (`_lib.LIB_OPTIONS`), set with tante_set_option;
  * host options -- module-level flags of this package, registered here by the module that owns them.
`tante_amd.set_option(name, value)` sets either kind at run time; `get_option(name)` reads it.  The process environment is read in this
file only, ONCE per switch at import time (`TANTE_<NAME>=<int>`), so that the measurement scripts under tools/ can flip a switch from the
command line; nothing else in the package looks at os.environ for behaviour (dist.py reads the torchrun rank variables, build.py HIPCC).
"""
from __future__ import annotations

import os
import sys
from typing import Dict, Tuple

_REGISTRY: Dict[str, Tuple[str, str, type]] = {}      # option name -> (module, attribute, type)


def register(name: str, default, module: str, attr: str):
    """Called at import by the module that owns the switch: returns its initial value (the environment's, else `default`)."""
    kind = type(default)
    _REGISTRY[name] = (module, attr, kind)
    raw = os.environ.get(name)
    if raw is None or raw == "":
        return default
    try:
        return kind(float(raw)) if kind in (int, float) else (raw != "0")
    except ValueError:
        raise ValueError(f"{name}={raw!r}: expected {'a number' if kind in (int, float) else '0 or 1'}") from None


def _load_owners():
    """The modules that own switches register them when they are imported; make sure all of them have been."""
    import importlib
    for m in ("attn_backbone", "autograd", "tante", "train_forward", "rollout", "cvit", "stages"):
        importlib.import_module("tante_amd." + m)


def host_options():
    _load_owners()
    return sorted(_REGISTRY)


EPOCH = [0]      # bumped by every set_option: captured graphs (rollout.GraphedRollout) re-capture when a switch changed under them


def set_option(name: str, value) -> None:
    """Set a host option (a registered module flag) or a library option (tante_set_option) by its TANTE_* name."""
    from . import _lib
    EPOCH[0] += 1
    if name not in _REGISTRY:
        _load_owners()
    if name in _REGISTRY:
        module, attr, kind = _REGISTRY[name]
        setattr(sys.modules[module], attr, kind(value) if kind is not bool else bool(int(value)))
        return
    if name in _lib.LIB_OPTIONS:
        _lib.set_option(name, int(value))
        return
    raise KeyError(f"unknown option {name!r}: host options {host_options()}; library options {list(_lib.LIB_OPTIONS)}")


def get_option(name: str):
    from . import _lib
    if name not in _REGISTRY:
        _load_owners()
    if name in _REGISTRY:
        module, attr, _ = _REGISTRY[name]
        return getattr(sys.modules[module], attr)
    if name in _lib.LIB_OPTIONS:
        return _lib.get_option(name, 0)
    raise KeyError(f"unknown option {name!r}")

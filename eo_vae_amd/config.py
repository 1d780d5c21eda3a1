"""Minimal stand-ins for the two config libraries the reference's entry points use
(omegaconf YAML loading with ${a.b} interpolation, hydra.utils.instantiate on `_target_`).
Neither library is installed on the MI355X image; PyYAML is."""
from __future__ import annotations

import importlib
import json
import os
import re
from typing import Any

import yaml

_INTERP = re.compile(r'\$\{([^${}]+)\}')


class _Loader(yaml.SafeLoader):
    pass


# YAML 1.1 does not read "1e-4" as a float; OmegaConf does.
_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    re.compile(r'''^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$''', re.X),
    list('-+0123456789.'))


def _lookup(root: dict, dotted: str):
    node = root
    for part in dotted.strip().split('.'):
        if isinstance(node, list):
            node = node[int(part)]
        else:
            node = node[part]
    return node


def _resolve(node, root, depth=0):
    if depth > 20:
        raise ValueError('config interpolation is recursive')
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        whole = _INTERP.fullmatch(node)
        if whole:
            return _resolve(_lookup(root, whole.group(1)), root, depth + 1)
        if _INTERP.search(node):
            return _INTERP.sub(lambda m: str(_resolve(_lookup(root, m.group(1)), root, depth + 1)), node)
    return node


def load_config(path: str, resolve: bool = True) -> dict:
    """YAML or JSON file -> plain dict with ${a.b} interpolations resolved."""
    if not os.path.exists(path):
        raise FileNotFoundError(f'Config file not found: {path}')
    with open(path) as f:
        text = f.read()
    data = json.loads(text) if path.endswith('.json') else yaml.load(text, Loader=_Loader)
    if resolve and isinstance(data, (dict, list)):
        data = _resolve(data, data)
    return data


# `_target_` strings of the reference's YAML files resolve to this package's implementations
TARGET_ALIASES = {
    'lightning.Trainer': 'eo_vae_amd.trainer.Trainer',
    'eo_vae.datasets.terramesh_datamodule.TerraMeshDataModule': 'eo_vae_amd.data.SyntheticDataModule',
}


def locate(dotted: str):
    dotted = TARGET_ALIASES.get(dotted, dotted)
    module, _, name = dotted.rpartition('.')
    if not module:
        raise ValueError(f'_target_ {dotted!r} is not a dotted path')
    return getattr(importlib.import_module(module), name)


def instantiate(cfg: Any, **overrides):
    """hydra.utils.instantiate semantics for the subset the configs use: recursive `_target_`
    construction, keyword overrides replace (already-built) sub-objects."""
    if isinstance(cfg, list):
        return [instantiate(c) for c in cfg]
    if not isinstance(cfg, dict):
        return cfg
    if '_target_' not in cfg:
        return {k: instantiate(v) for k, v in cfg.items()}
    kwargs = {}
    for k, v in cfg.items():
        if k == '_target_' or k in overrides:
            continue
        kwargs[k] = instantiate(v) if isinstance(v, dict) and '_target_' in v else v
    kwargs.update(overrides)
    return locate(cfg['_target_'])(**kwargs)

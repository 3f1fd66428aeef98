"""Config plumbing for `experiment=cDDPM/DDPM_cond_spark_2D` without Hydra / OmegaConf.

When hydra is installed the reference's own entry (`run.py` -> `@hydra.main(config_path="configs/",
config_name="config.yaml")`, reference run.py:20-56) composes the config and instantiates
`_target_: src.models.DDPM_2D.DDPM_2D` (configs/model/DDPM_2D.yaml:1); nothing here is needed.
Neither package is in this image, so this module composes the few files that fix the UNet shape with
PyYAML: configs/config.yaml defaults -> configs/model/DDPM_2D.yaml, configs/datamodule/IXI.yaml, then the
experiment overlay (`# @package _global_`), then `key=value` overrides; `${datamodule.cfg.x}`
interpolations are resolved. It reads the reference's YAML files where they lie (path given by the caller):
no config text is copied into this repository.
"""
from __future__ import annotations

import copy
import os
import re
from typing import Any, Dict, Iterable, Optional

import yaml

_INTERP = re.compile(r"^\$\{([A-Za-z0-9_.]+)\}$")


def _load(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _lookup(root: Dict[str, Any], dotted: str):
    cur: Any = root
    for part in dotted.split("."):
        if not isinstance(cur, dict) or part not in cur:
            raise KeyError(dotted)
        cur = cur[part]
    return cur


def _resolve(node: Any, root: Dict[str, Any], depth: int = 0) -> Any:
    if depth > 10:
        raise RecursionError("interpolation loop in config")
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node.strip())
        if m:
            try:
                return _resolve(_lookup(root, m.group(1)), root, depth + 1)
            except KeyError:
                return node      # left as written (e.g. ${oc.env:...}); the path does not read such keys
    return node


def _parse_value(text: str) -> Any:
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def compose(config_dir: str, experiment: str = "cDDPM/DDPM_cond_spark_2D", overrides: Iterable[str] = ()) -> Dict[str, Any]:
    """Compose `config.yaml` + the groups its defaults list names + the experiment overlay + overrides."""
    root = _load(os.path.join(config_dir, "config.yaml"))
    groups: Dict[str, Optional[str]] = {}
    for item in root.pop("defaults", []) or []:
        if isinstance(item, dict):
            for g, name in item.items():
                groups[g.lstrip("/").replace("override ", "")] = name
    exp_path = os.path.join(config_dir, "experiment", experiment + ("" if experiment.endswith(".yaml") else ".yaml"))
    exp = _load(exp_path)
    for item in exp.pop("defaults", []) or []:
        if isinstance(item, dict):
            for g, name in item.items():
                groups[g.replace("override ", "").strip().lstrip("/")] = name
    cfg: Dict[str, Any] = {}
    for g, name in groups.items():
        if g in ("experiment", "hparams_search") or not name:
            continue
        path = os.path.join(config_dir, g, name if str(name).endswith(".yaml") else f"{name}.yaml")
        if os.path.exists(path):
            _merge(cfg, {g: _load(path)})
    _merge(cfg, root)
    _merge(cfg, exp)        # '# @package _global_': merged at the root
    for ov in overrides:
        key, _, val = ov.partition("=")
        node = cfg
        parts = key.lstrip("+").split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_value(val)
    return _resolve(cfg, cfg)


def instantiate_model(cfg: Dict[str, Any], encoder=None):
    """what `hydra.utils.instantiate(cfg.model)` does for `_target_: src.models.DDPM_2D.DDPM_2D`
    (reference src/train.py:98), bound to this package's DDPM_2D."""
    from .DDPM_2D import DDPM_2D, AttrDict

    model_cfg = cfg["model"]
    target = model_cfg.get("_target_", "")
    if not target.endswith("DDPM_2D.DDPM_2D"):
        raise NotImplementedError(f"only the cDDPM target src.models.DDPM_2D.DDPM_2D is provided, got {target!r}")
    return DDPM_2D(AttrDict(model_cfg["cfg"]), prefix=None, encoder=encoder)

"""A small stand-in for the parts of Hydra / OmegaConf the reference's model configs use (neither package is installed):
defaults-list composition with config groups and packages (``group: option``, ``group@pkg: option``, ``- option``,
``override group: option``, ``_self_``), dotted command-line overrides (``a.b=v``, ``+a.b=v``, ``group/sub=option``,
``group@pkg=option``), relative / absolute ``${...}`` interpolation with the ``hydra:runtime.cwd`` resolver, and
``instantiate`` (``_target_``, ``_partial_``, ``_recursive_``).  TEST INFRASTRUCTURE ONLY: it exists so that the drop-in
tests can build objects from the reference's own ``configs/model/**.yaml`` exactly as ``hydra.utils.instantiate`` would
(demo/demo.py:107-111, sam_pt/vos_eval/eval.py)."""
from __future__ import annotations

import copy
import functools
import importlib
import os
import re
from typing import Any, Dict, List, Optional

import yaml


def _load(path: str) -> dict:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg: dict, path: str, value: Any):
    keys = [k for k in path.split(".") if k]
    node = cfg
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    if isinstance(value, dict) and isinstance(node.get(keys[-1]), dict):
        _merge(node[keys[-1]], value)
    else:
        node[keys[-1]] = value


def compose(config_dir: str, group: str, option: str, choices: Optional[Dict[str, str]] = None) -> dict:
    """Config of ``<config_dir>/<group>/<option>.yaml`` with its defaults list resolved recursively.  ``choices`` maps a
    defaults key ("point_tracker", "sam@sam_predictor.sam_model", "image_encoder", with the group path relative to the
    file that lists it or absolute from config_dir) to another option, like Hydra's ``group=option`` overrides."""
    choices = choices or {}

    def build(grp: str, opt: str) -> dict:
        raw = _load(os.path.join(config_dir, grp, opt + ".yaml"))
        defaults: List = raw.pop("defaults", [])
        out: dict = {}
        self_done = False
        # an `override g: o` entry replaces the option of a group introduced by an earlier (included) file
        overrides = {}
        for d in defaults:
            if isinstance(d, dict):
                (k, v), = d.items()
                if k.startswith("override "):
                    overrides[k[len("override "):].strip()] = v
        for d in defaults:
            if d == "_self_":
                _merge(out, raw)
                self_done = True
            elif isinstance(d, str):                                   # another option of the same group
                _merge(out, build_with(grp, d, overrides))
            else:
                (k, v), = d.items()
                if k.startswith("override "):
                    continue
                sub, _, pkg = k.partition("@")
                v = choices.get(k, choices.get(os.path.join(grp, sub), v))
                node = build(os.path.join(grp, sub), v)
                _set_path(out, pkg if pkg else sub, node)
        if not self_done:
            _merge(out, raw)
        return out

    def build_with(grp: str, opt: str, overrides: dict) -> dict:
        saved = dict(choices)
        for k, v in overrides.items():
            choices.setdefault(k, v)
        try:
            return build(grp, opt)
        finally:
            choices.clear()
            choices.update(saved)

    return build(group, option)


def apply_overrides(cfg: dict, overrides: List[str]) -> dict:
    """``a.b.c=value`` / ``+a.b=value`` (YAML-typed values) on a composed config."""
    for o in overrides:
        key, _, val = o.partition("=")
        key = key.lstrip("+")
        _set_path(cfg, key, yaml.safe_load(val))
    return cfg


_INTERP = re.compile(r"\$\{\s*([^}]+?)\s*\}")


def resolve(cfg: dict, cwd: str = "/nonexistent") -> dict:
    """Resolve ``${...}`` in place: ``${hydra:runtime.cwd}``, absolute ``${a.b}``, relative ``${.a}`` / ``${..a}`` (one
    leading dot = the node holding the value, each further dot one level up, as OmegaConf)."""

    def lookup(path_keys: List[str], expr: str):
        if expr.startswith("hydra:"):
            return cwd if expr == "hydra:runtime.cwd" else None
        dots = len(expr) - len(expr.lstrip("."))
        rest = [k for k in expr.lstrip(".").split(".") if k]
        base = path_keys[:len(path_keys) - dots] if dots else []
        node = cfg
        for k in base + rest:
            node = node[int(k)] if isinstance(node, list) else node[k]
        return node

    def walk(node, keys):
        items = node.items() if isinstance(node, dict) else enumerate(node)
        for k, v in list(items):
            here = keys + [str(k)]
            if isinstance(v, (dict, list)):
                walk(v, here)
            elif isinstance(v, str) and "${" in v:
                m = _INTERP.fullmatch(v.strip())
                for _ in range(8):                                      # chains of interpolations
                    if m:
                        v = lookup(here, m.group(1))
                    else:
                        v = _INTERP.sub(lambda mm: str(lookup(here, mm.group(1))), v)
                    if not (isinstance(v, str) and "${" in v):
                        break
                    m = _INTERP.fullmatch(v.strip())
                node[k] = v

    walk(cfg, [])
    return cfg


def _locate(target: str):
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node: Any, **kwargs) -> Any:
    """hydra.utils.instantiate on a resolved plain-dict config."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    recursive = node.get("_recursive_", True)
    partial = node.get("_partial_", False)
    args = {k: (instantiate(v) if recursive else copy.deepcopy(v)) for k, v in node.items()
            if k not in ("_target_", "_recursive_", "_partial_", "_convert_")}
    args.update(kwargs)
    fn = _locate(node["_target_"])
    return functools.partial(fn, **args) if partial else fn(**args)

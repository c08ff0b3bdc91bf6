"""YAML-template + JSON-override configs with ``${...}`` interpolation: the subset of OmegaConf the reference's
runners use (``OmegaConf.load`` / ``create`` / ``merge`` / ``to_yaml``, attribute access, nested groups, lazy
``${key}`` / ``${group.key}`` interpolation -- ``i2vgen-xl/run_group_pnp_edit.py:81,195`` and
``configs/group_*/template.yaml``).  OmegaConf itself is not installed here; PyYAML is."""
from __future__ import annotations

import copy
import re
from typing import Any

import yaml

_PAT = re.compile(r"\$\{([^{}]+)\}")


class Config:
    """Nested dict with attribute access; string values are interpolated lazily against the ROOT config."""

    def __init__(self, data=None, root: "Config" = None):
        object.__setattr__(self, "_data", {})
        object.__setattr__(self, "_root", root if root is not None else self)
        for k, v in (data or {}).items():
            self._data[k] = self._wrap(v)

    def _wrap(self, v):
        if isinstance(v, Config):
            return Config(v.to_container(resolve=False), self._root)
        if isinstance(v, dict):
            return Config(v, self._root)
        return v

    # -- access
    def _resolve(self, v, depth=0):
        if isinstance(v, str) and "${" in v:
            if depth > 32:
                raise ValueError(f"interpolation cycle in {v!r}")
            m = _PAT.fullmatch(v)
            if m:  # whole value is one reference: keep the referenced type (e.g. a list)
                return self._resolve(self._root._lookup(m.group(1)), depth + 1)
            return _PAT.sub(lambda mm: str(self._resolve(self._root._lookup(mm.group(1)), depth + 1)), v)
        return v

    def _lookup(self, dotted: str):
        node: Any = self
        for part in dotted.strip().split("."):
            if not isinstance(node, Config) or part not in node._data:
                raise KeyError(f"interpolation key '{dotted}' not found")
            node = node._data[part]
        return node

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"Missing key {k}")

    def __getitem__(self, k):
        return self._resolve(self._data[k])

    def __setattr__(self, k, v):
        self._data[k] = self._wrap(v)

    __setitem__ = __setattr__

    def __contains__(self, k):
        return k in self._data

    def get(self, k, default=None):
        return self[k] if k in self._data else default

    def keys(self):
        return self._data.keys()

    def items(self):
        return [(k, self[k]) for k in self._data]

    def __iter__(self):
        return iter(self._data)

    def to_container(self, resolve=True):
        out = {}
        for k, v in self._data.items():
            if isinstance(v, Config):
                out[k] = v.to_container(resolve)
            else:
                out[k] = self._resolve(v) if resolve else v
        return out

    def __repr__(self):
        return f"Config({self.to_container(resolve=False)!r})"


def _rebind(cfg: Config, root: Config):
    object.__setattr__(cfg, "_root", root)
    for v in cfg._data.values():
        if isinstance(v, Config):
            _rebind(v, root)


class OmegaConf:
    """Drop-in for the four OmegaConf calls the reference makes."""

    @staticmethod
    def load(path) -> Config:
        with open(path) as f:
            return Config(yaml.safe_load(f) or {})

    @staticmethod
    def create(obj=None) -> Config:
        return Config(copy.deepcopy(obj) if obj is not None else {})

    @staticmethod
    def merge(*cfgs) -> Config:
        def merge2(a: dict, b: dict):
            for k, v in b.items():
                if isinstance(v, dict) and isinstance(a.get(k), dict):
                    merge2(a[k], v)
                else:
                    a[k] = copy.deepcopy(v)
            return a
        acc: dict = {}
        for c in cfgs:
            merge2(acc, c.to_container(resolve=False) if isinstance(c, Config) else dict(c))
        out = Config(acc)
        _rebind(out, out)
        return out

    @staticmethod
    def from_dotlist(items) -> Config:
        """``["a.b=1", "c=text"]`` -> nested config; values parsed as YAML scalars (``consisti2v/run_pnp_edit.py:138-140``)."""
        acc: dict = {}
        for it in items:
            key, _, val = str(it).partition("=")
            node = acc
            parts = key.strip().split(".")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = yaml.safe_load(val) if val != "" else None
        return Config(acc)

    @staticmethod
    def to_yaml(cfg: Config, resolve: bool = False) -> str:
        return yaml.safe_dump(cfg.to_container(resolve=resolve), sort_keys=False)

    @staticmethod
    def to_container(cfg: Config, resolve: bool = False):
        return cfg.to_container(resolve=resolve)

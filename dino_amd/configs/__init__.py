"""Config loading with the reference's schema (dino/configs/default.yaml) and call pattern
(dino/configs/__init__.py:6-11, dino/utils/config.py:24-40) on PyYAML -- omegaconf is not a
dependency: nested attribute access, file-over-default merge, `a.b=c` dot-list overrides and
`${a.b}` interpolation are implemented here."""
import copy
import os
import re

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))


class Config(dict):
    """dict with attribute access, recursively."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


# PyYAML's default resolver reads "1e-5" as a string (it wants "1.0e-5"); accept both.
class _Loader(yaml.SafeLoader):
    pass


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*)(?:\.[0-9_]*)?[eE][-+]?[0-9]+$"), list("-+0123456789"))


def _load_yaml(path):
    with open(path) as f:
        return yaml.load(f, Loader=_Loader) or {}


def load_config(config_name: str):
    return _wrap(_load_yaml(os.path.join(_HERE, config_name + ".yaml")))


def merge(base, over):
    """Recursive merge: `over` wins; dicts merge key-wise, everything else is replaced."""
    out = copy.deepcopy(base)
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def from_dotlist(opts):
    """["a.b=1", "c=x"] -> nested dict with YAML-typed values."""
    out = {}
    for item in opts or []:
        key, _, val = item.partition("=")
        node = out
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.load(val, Loader=_Loader) if val != "" else None
    return out


_INTERP = re.compile(r"\$\{([^}]+)\}")


def resolve(cfg):
    """Replace ${a.b.c} references (whole-string or embedded) by the referenced value."""
    root = cfg

    def lookup(path):
        node = root
        for p in path.split("."):
            node = node[p]
        return node

    def walk(x):
        if isinstance(x, dict):
            return {k: walk(v) for k, v in x.items()}
        if isinstance(x, list):
            return [walk(v) for v in x]
        if isinstance(x, str):
            m = _INTERP.fullmatch(x)
            if m:
                return walk(lookup(m.group(1)))
            return _INTERP.sub(lambda mm: str(lookup(mm.group(1))), x)
        return x
    return walk(cfg)


default_config = load_config("default")


def get_cfg_from_file(config_file, opts=None):
    """default <- file <- dot-list overrides, interpolations resolved (utils/config.py:24-40)."""
    cfg = merge(default_config, _load_yaml(config_file) if config_file else {})
    cfg = merge(cfg, from_dotlist(opts))
    return _wrap(resolve(cfg))

"""Python-file configs with `_base_` inheritance and dotted overrides.

A small stand-in for the part of mmcv.Config the reference's test path uses
(tools/test.py:67-70: Config.fromfile + merge_from_dict); mmcv is not a dependency.
"""
import ast
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return ConfigDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return type(o)(_wrap(v) for v in o)
    return o


def _merge(base, over):
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = v
    return out


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg', _wrap(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(filename)
        with open(filename) as f:
            src = f.read()
        ast.parse(src)          # syntax check with a clear error
        scope = {'__file__': filename}
        exec(compile(src, filename, 'exec'), scope)
        cfg = {k: v for k, v in scope.items()
               if not k.startswith('__') and not callable(v) and type(v).__name__ != 'module'}
        base = cfg.pop('_base_', None)
        if base:
            merged = {}
            for b in ([base] if isinstance(base, str) else base):
                merged = _merge(merged, Config._file2dict(os.path.join(os.path.dirname(filename), b)))
            cfg = _merge(merged, cfg)
        return cfg

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename)

    def merge_from_dict(self, options):
        """dotted keys, e.g. {'model.generator.vsr': True} (mmcv DictAction semantics)."""
        over = {}
        for full, v in options.items():
            d = over
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, '_cfg', _wrap(_merge(self._cfg, over)))

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __getitem__(self, k):
        return self._cfg[k]

    def __contains__(self, k):
        return k in self._cfg

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def to_dict(self):
        return self._cfg


def parse_cfg_options(pairs):
    """['a.b=1', 'c=True'] -> dict (argparse helper, like mmcv.DictAction)."""
    out = {}
    for kv in pairs or []:
        k, v = kv.split('=', 1)
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
        out[k] = v
    return out

"""Minimal registry + build_from_cfg with the reference's names.

mmcv is not a dependency of this build; these mirror the tiny part of
mmcv.utils.Registry / build_from_cfg that the reference's hot-path boundary uses
(/root/reference/mmedit/models/registry.py:1-8, mmedit/models/builder.py:8-60).
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    @property
    def module_dict(self):
        return self._module_dict

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if not force and key in self._module_dict:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        raise KeyError(f'`cfg` must contain the key "type", but got {cfg}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    else:
        obj_cls = obj_type
    return obj_cls(**args)


# the reference uses ONE registry under four aliases (models/registry.py:5-8)
MODELS = Registry('model')
BACKBONES = MODELS
COMPONENTS = MODELS
LOSSES = MODELS
DATASETS = Registry('dataset')
PIPELINES = Registry('pipeline')


def build(cfg, registry, default_args=None):
    """models/builder.py:8-23."""
    if isinstance(cfg, list):
        import torch.nn as nn
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    """models/builder.py:25-31."""
    return build(cfg, BACKBONES)


def build_component(cfg):
    return build(cfg, COMPONENTS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """models/builder.py:52-60."""
    return build(cfg, MODELS, dict(train_cfg=train_cfg, test_cfg=test_cfg))

"""ORACLE -- test infrastructure; runs ONLY in the build container.

Imports the *actual reference* hot path from /root/reference on CPU so that
golden vectors can be generated from it (oracle/gen_golden.py).  The reference
needs mmcv-full, which is not installed; the few mmcv names its hot-path
modules touch at import time are provided here as in-process stand-ins in
sys.modules.  None of this -- nor any reference source or
bytecode -- travels to the GPU box: only the produced .npz outputs do.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get('PNP_REFERENCE_ROOT', '/root/reference')


class _Registry:
    def __init__(self, name='models', **kwargs):
        self.name = name
        self._m = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._m[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._m.get(key)


def _build_from_cfg(cfg, registry, default_args=None):
    cfg = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            cfg.setdefault(k, v)
    typ = cfg.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    return cls(**cfg)


def _kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


class _ModulatedDeformConv2d(nn.Module):
    """Parameter container with the mmcv ctor signature; the op itself is absent."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deform_groups=1, bias=True):
        super().__init__()
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, k
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels // groups, *k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None


def _absent(*a, **k):
    raise RuntimeError('mmcv op not available in the oracle shim')


def install():
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_pnp_shim', False):
        return
    mmcv = types.ModuleType('mmcv')
    mmcv._pnp_shim = True
    mmcv.__version__ = '1.5.0'
    mmcv.build_from_cfg = _build_from_cfg
    cnn = types.ModuleType('mmcv.cnn')
    cnn.ConvModule = type('ConvModule', (nn.Module,), {})
    cnn.MODELS = _Registry('model')
    cnn.kaiming_init = _kaiming_init
    cnn.constant_init = _constant_init
    cnn.build_activation_layer = _absent
    cnn.build_conv_layer = _absent
    cnn.build_norm_layer = _absent
    cnn.normal_init = _absent
    cnn.xavier_init = _absent
    runner = types.ModuleType('mmcv.runner')
    runner.load_checkpoint = _absent
    runner.auto_fp16 = lambda *a, **k: (lambda f: f)
    ops = types.ModuleType('mmcv.ops')
    ops.ModulatedDeformConv2d = _ModulatedDeformConv2d
    ops.modulated_deform_conv2d = _absent
    ops.DeformConv2d = type('DeformConv2d', (nn.Module,), {})
    ops.DeformConv2dPack = type('DeformConv2dPack', (nn.Module,), {})
    ops.deform_conv2d = _absent
    utils = types.ModuleType('mmcv.utils')
    utils.Registry = _Registry
    utils.build_from_cfg = _build_from_cfg
    utils.get_logger = lambda *a, **k: __import__('logging').getLogger('mmedit')
    pw = types.ModuleType('mmcv.utils.parrots_wrapper')
    pw._BatchNorm = torch.nn.modules.batchnorm._BatchNorm
    mmcv.cnn, mmcv.runner, mmcv.ops, mmcv.utils = cnn, runner, ops, utils
    for name, mod in (('mmcv', mmcv), ('mmcv.cnn', cnn), ('mmcv.runner', runner), ('mmcv.ops', ops),
                      ('mmcv.utils', utils), ('mmcv.utils.parrots_wrapper', pw)):
        sys.modules[name] = mod

    def ns(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
        sys.modules[name] = m
        return m

    ns('mmedit', 'mmedit')
    ns('mmedit.models', 'mmedit/models')
    common = ns('mmedit.models.common', 'mmedit/models/common')
    ns('mmedit.models.backbones', 'mmedit/models/backbones')
    ns('mmedit.models.backbones.sr_backbones', 'mmedit/models/backbones/sr_backbones')
    mu = types.ModuleType('mmedit.utils')
    mu.get_root_logger = lambda *a, **k: __import__('logging').getLogger('mmedit')
    sys.modules['mmedit.utils'] = mu
    importlib.import_module('mmedit.models.registry')
    for sub in ('flow_warp', 'upsample', 'sr_backbone_utils'):
        m = importlib.import_module(f'mmedit.models.common.{sub}')
        for k, v in vars(m).items():
            if not k.startswith('_'):
                setattr(common, k, v)


def reference_generator_class():
    install()
    mod = importlib.import_module('mmedit.models.backbones.sr_backbones.iconvsr_ipb_par')
    return mod.IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par


def reference_flow_warp():
    install()
    return importlib.import_module('mmedit.models.common.flow_warp').flow_warp


def reference_modules():
    """(sr_backbone_utils, domain_aware, basicvsr_net) modules of the reference."""
    install()
    reference_generator_class()
    return (importlib.import_module('mmedit.models.common.sr_backbone_utils'),
            importlib.import_module('mmedit.models.backbones.sr_backbones.domain_aware'),
            importlib.import_module('mmedit.models.backbones.sr_backbones.basicvsr_net'))


def reference_loader_class():
    """LoadImageFromFileList_ipb (mmedit/datasets/pipelines/loading_ipb.py:221-397), importable with
    a disk FileClient and a PIL-backed imfrombytes standing in for mmcv's (file plumbing only; the
    MV / partition rasterisation arithmetic is the reference's own numpy code)."""
    install()
    import io
    import numpy as np
    from PIL import Image
    mmcv = sys.modules['mmcv']
    if 'mmcv.fileio' not in sys.modules:
        fileio = types.ModuleType('mmcv.fileio')

        class FileClient:
            def __init__(self, backend='disk', **kwargs):
                assert backend == 'disk'

            def get(self, filepath):
                with open(filepath, 'rb') as f:
                    return f.read()
        fileio.FileClient = FileClient
        mmcv.fileio = fileio
        sys.modules['mmcv.fileio'] = fileio

        def imfrombytes(content, flag='color', channel_order='bgr', backend=None):
            img = np.array(Image.open(io.BytesIO(content)).convert('RGB'))
            return img if channel_order == 'rgb' else img[..., ::-1].copy()
        mmcv.imfrombytes = imfrombytes
        core = types.ModuleType('mmedit.core')
        core.__path__ = []
        mask = types.ModuleType('mmedit.core.mask')
        for n in ('bbox2mask', 'brush_stroke_mask', 'get_irregular_mask', 'random_bbox'):
            setattr(mask, n, _absent)
        sys.modules['mmedit.core'] = core
        sys.modules['mmedit.core.mask'] = mask
        for name, rel in (('mmedit.datasets', 'mmedit/datasets'), ('mmedit.datasets.pipelines', 'mmedit/datasets/pipelines')):
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
            sys.modules[name] = m
    mod = importlib.import_module('mmedit.datasets.pipelines.loading_ipb')
    return mod.LoadImageFromFileList_ipb


def reference_metrics():
    """(psnr, tensor2img, BasicVSR) of the reference itself: mmedit/core/evaluation/metrics.py:170-215,
    mmedit/core/misc.py:9-74, mmedit/models/restorers/basicvsr.py:14 (its `evaluate`, :119-153).  Name-only stand-ins for
    what those files import and the PSNR path never calls: cv2, torchvision.utils.make_grid (4-D batches only),
    MATLABLikeResize (NIQE only).  No arithmetic is supplied from here."""
    reference_loader_class()          # sets up the mmedit.core / mmedit.datasets stand-in packages
    if 'cv2' not in sys.modules:
        cv2 = types.ModuleType('cv2')
        for n in ('filter2D', 'getGaussianKernel', 'BORDER_REPLICATE', 'resize'):
            setattr(cv2, n, _absent)
        sys.modules['cv2'] = cv2
    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tvu = types.ModuleType('torchvision.utils')
        tvu.make_grid = _absent
        tv.utils = tvu
        sys.modules['torchvision'] = tv
        sys.modules['torchvision.utils'] = tvu
    if 'mmedit.datasets.pipelines.matlab_like_resize' not in sys.modules:
        mlr = types.ModuleType('mmedit.datasets.pipelines.matlab_like_resize')
        mlr.MATLABLikeResize = type('MATLABLikeResize', (), {})
        sys.modules['mmedit.datasets.pipelines.matlab_like_resize'] = mlr
    core = sys.modules['mmedit.core']
    core.__path__ = [os.path.join(REFERENCE_ROOT, 'mmedit', 'core')]
    if 'mmedit.core.evaluation' not in sys.modules:
        ev = types.ModuleType('mmedit.core.evaluation')
        ev.__path__ = [os.path.join(REFERENCE_ROOT, 'mmedit', 'core', 'evaluation')]
        sys.modules['mmedit.core.evaluation'] = ev
    metrics = importlib.import_module('mmedit.core.evaluation.metrics')
    misc = importlib.import_module('mmedit.core.misc')
    core.psnr, core.ssim, core.tensor2img = metrics.psnr, metrics.ssim, misc.tensor2img
    if 'mmedit.models.restorers' not in sys.modules:
        r = types.ModuleType('mmedit.models.restorers')
        r.__path__ = [os.path.join(REFERENCE_ROOT, 'mmedit', 'models', 'restorers')]
        sys.modules['mmedit.models.restorers'] = r
    basicvsr = importlib.import_module('mmedit.models.restorers.basicvsr')
    return metrics.psnr, misc.tensor2img, basicvsr.BasicVSR


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'mmedit'))

"""pnp_vcve_amd -- MI355X-native (gfx950) BAE/CAA forward hot path of PnP-VCVE.

Importing the package registers the generator under the reference's registry name
(mmedit/models/registry.py BACKBONES).  Nothing here touches the GPU at import time.
"""
from .registry import (BACKBONES, COMPONENTS, DATASETS, LOSSES, MODELS, PIPELINES, Registry,  # noqa: F401
                       build_backbone, build_component, build_from_cfg, build_loss, build_model)
from . import generator  # noqa: F401  (registers the class)
from . import torch_ops  # noqa: F401  (registers torch.ops.pnpvcve.*)

__version__ = '0.1.0'

"""PyTorch custom-op registration of the hot path (`torch.ops.pnpvcve.*`).

north_star asks for the HIP kernels to be "invoked from Python via PyTorch-ROCm custom ops": these are thin
`torch.library` registrations over the C ABI (include/pnpvcve.h) -- torch carries the tensors, the device memory and
the current stream; the arithmetic is libpnpvcve_hip.so.  Each op has a fake (meta) implementation so that shape
propagation / tracing works without a GPU; there is no CPU implementation (a CPU tensor raises, like the rest of the
package).

    torch.ops.pnpvcve.flow_warp(x, flow)                        mmedit/models/common/flow_warp.py:6-50
    torch.ops.pnpvcve.mv_warp(feat_hwc, flow_x, flow_y)         iconvsr_mv.py:17-18 in the fused path's layout
    torch.ops.pnpvcve.psnr_sse(a, b, crop_border)               core/evaluation/metrics.py:200-215 statistic
    torch.ops.pnpvcve.generator_forward(handle, lrs, mvs, par, side)   iconvsr_ipb_par.py:44-149

`handle` is the integer id under which a generator module registered itself (register_generator); `side` is the CPU
float tensor (3, n, t) = (slices, QPs, base_QPs).  generator.forward() goes through this op.
"""
import weakref

import torch

from . import ops

_GENERATORS = weakref.WeakValueDictionary()
_NEXT_ID = [1]


def register_generator(module):
    hid = _NEXT_ID[0]
    _NEXT_ID[0] += 1
    _GENERATORS[hid] = module
    return hid


@torch.library.custom_op('pnpvcve::flow_warp', mutates_args=())
def flow_warp(x: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    return ops.flow_warp(x, flow)


@flow_warp.register_fake
def _(x, flow):
    return torch.empty_like(x)


@torch.library.custom_op('pnpvcve::mv_warp', mutates_args=())
def mv_warp(feat: torch.Tensor, flow_x: torch.Tensor, flow_y: torch.Tensor) -> torch.Tensor:
    return ops.mv_warp_nhwc(feat, flow_x, flow_y)


@mv_warp.register_fake
def _(feat, flow_x, flow_y):
    return torch.empty_like(feat)


@torch.library.custom_op('pnpvcve::psnr_sse', mutates_args=())
def psnr_sse(a: torch.Tensor, b: torch.Tensor, crop_border: int) -> torch.Tensor:
    """per-frame PSNR (float64, CPU) of the uint8-rounded frames; a, b (..., c, h, w)"""
    return ops.psnr_frames(a, b, crop_border)


@psnr_sse.register_fake
def _(a, b, crop_border):
    return torch.empty(a.shape[:-3], dtype=torch.float64)


@torch.library.custom_op('pnpvcve::generator_forward', mutates_args=())
def generator_forward(handle: int, lrs: torch.Tensor, mvs: torch.Tensor, par: torch.Tensor,
                      side: torch.Tensor) -> torch.Tensor:
    m = _GENERATORS.get(handle)
    if m is None:
        raise RuntimeError(f'pnpvcve::generator_forward: unknown generator handle {handle}')
    return m._forward_native(lrs, mvs, par, side)


@generator_forward.register_fake
def _(handle, lrs, mvs, par, side):
    m = _GENERATORS.get(handle)
    s = 4 if (m is not None and m.vsr) else 1
    n, t, _, h, w = lrs.shape
    return lrs.new_empty((n, t, 3, h * s, w * s))

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

    torch.ops.pnpvcve.conv3x3(srcs, packed_w, bias, gamma, packed_w1x1, par, residual, act)
                                                                basicvsr_net.py:484, sr_backbone_utils.py:304-333 halves, iconvsr.py:365
    torch.ops.pnpvcve.expert_mix(experts, attention)            Dynamic_conv2d_se.forward's mm(attention, weight), sr_backbone_utils.py:198-199
    torch.ops.pnpvcve.bae_block(x, w2, b2, gamma, w1x1, par, w1, b1)   ResidualBlockNoBNDynamic_drt.forward, :304-333
    torch.ops.pnpvcve.pixel_shuffle_conv(x, packed, act)        PixelShufflePack.forward, common/upsample.py:40-51

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


@torch.library.custom_op('pnpvcve::conv3x3', mutates_args=())
def conv3x3(srcs: list[torch.Tensor], packed_w: list[torch.Tensor], bias: torch.Tensor | None, gamma: torch.Tensor | None,
            packed_w1x1: torch.Tensor | None, par: torch.Tensor | None, residual: torch.Tensor | None,
            act: int) -> torch.Tensor:
    return ops.conv3x3(srcs, packed_w, bias=bias, gamma=gamma, packed_w1x1=packed_w1x1, par=par, residual=residual, act=act)


@conv3x3.register_fake
def _(srcs, packed_w, bias, gamma, packed_w1x1, par, residual, act):
    h, w = srcs[0].shape[:2]
    return srcs[0].new_empty((h, w, 64))


@torch.library.custom_op('pnpvcve::expert_mix', mutates_args=())
def expert_mix(experts: torch.Tensor, attention: torch.Tensor) -> torch.Tensor:
    """(E,64,64,3,3) expert bank + (E,) attention -> the packed image of sum_e attention[e] * experts[e]"""
    return ops.pack_conv3x3(experts, ew=attention)


@expert_mix.register_fake
def _(experts, attention):
    return experts.new_empty((9 * 4096,))


@torch.library.custom_op('pnpvcve::bae_block', mutates_args=())
def bae_block(x: torch.Tensor, w2_packed: torch.Tensor, b2: torch.Tensor | None, gamma: torch.Tensor | None,
              w1x1_packed: torch.Tensor | None, par: torch.Tensor | None, w1_packed: torch.Tensor,
              b1: torch.Tensor | None) -> torch.Tensor:
    return ops.bae_block(x, w2_packed, b2, gamma, w1x1_packed, par, w1_packed, b1)


@bae_block.register_fake
def _(x, w2_packed, b2, gamma, w1x1_packed, par, w1_packed, b1):
    return torch.empty_like(x)


@torch.library.custom_op('pnpvcve::pixel_shuffle_conv', mutates_args=())
def pixel_shuffle_conv(x: torch.Tensor, packed: torch.Tensor, act: int) -> torch.Tensor:
    return ops.pixel_shuffle_conv(x, packed, act)


@pixel_shuffle_conv.register_fake
def _(x, packed, act):
    h, w, c = x.shape
    return x.new_empty((2 * h, 2 * w, c))

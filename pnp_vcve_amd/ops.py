"""Tensor-level wrappers over the C ABI (device pointers + the current HIP stream).

PyTorch is plumbing here: it owns device memory and the stream; every computation
runs in libpnpvcve_hip.so.  All functions require CUDA(HIP) fp32 contiguous tensors and
raise otherwise -- there is deliberately no CPU path.
"""
import ctypes

import torch

from . import _native


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _on_device_of_first_tensor(fn):
    """Run the wrapper with the first tensor argument's device current: the launch goes to torch's current stream of
    THAT device (a process may hold tensors on several GPUs), like generator.forward does."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        first = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if first is None:
            first = next((a[0] for a in args if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor)), None)
        if first is not None and first.is_cuda:
            with torch.cuda.device(first.device):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapped


def _chk(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA/HIP tensor: the PnP-VCVE hot path has no CPU fallback')
    if t.dtype != torch.float32:
        raise TypeError(f'{name} must be float32, got {t.dtype}')
    return t.contiguous()


_WARP_MODES = {'bilinear': 0, 'nearest': 1}       # F.grid_sample modes flow_warp.py:47 can receive through flow_inter


@_on_device_of_first_tensor
def flow_warp(x, flow, interpolation='bilinear', padding_mode='zeros', align_corners=True):
    """Drop-in for mmedit.models.common.flow_warp (flow_warp.py:6-50).
    x (n,c,h,w), flow (n,h,w,2) in pixels."""
    if x.size()[-2:] != flow.size()[1:3]:
        raise ValueError(f'The spatial sizes of input ({x.size()[-2:]}) and '
                         f'flow ({flow.size()[1:3]}) are not the same.')
    if interpolation not in _WARP_MODES or padding_mode != 'zeros' or not align_corners:
        raise NotImplementedError("only 'bilinear' | 'nearest' / zeros / align_corners=True (what the generator can ask for)")
    x, flow = _chk(x, 'x'), _chk(flow, 'flow')
    n, c, h, w = x.shape
    out = torch.empty_like(x)
    _native.check(_native.lib().pnp_flow_warp_nchw_mode_f32(_ptr(x), _ptr(flow), _ptr(out), n, c, h, w, _WARP_MODES[interpolation],
                                                            _stream()), 'pnp_flow_warp_nchw_mode_f32')
    return out


@_on_device_of_first_tensor
def mv_warp_nhwc(feat, flow_x, flow_y, interpolation='bilinear'):
    """feat (h,w,c) pixel-major; flow_x/flow_y (h,w)."""
    feat, flow_x, flow_y = _chk(feat, 'feat'), _chk(flow_x, 'flow_x'), _chk(flow_y, 'flow_y')
    h, w, c = feat.shape
    out = torch.empty_like(feat)
    _native.check(_native.lib().pnp_mv_warp_nhwc_mode_f32(_ptr(feat), _ptr(flow_x), _ptr(flow_y), _ptr(out), h, w, c,
                                                          _WARP_MODES[interpolation], _stream()), 'pnp_mv_warp_nhwc_mode_f32')
    return out


@_on_device_of_first_tensor
def nchw_to_nhwc(x):
    x = _chk(x, 'x')
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), device=x.device, dtype=x.dtype)
    _native.check(_native.lib().pnp_nchw_to_nhwc_f32(_ptr(x), _ptr(out), n, c, h, w, _stream()), 'nchw_to_nhwc')
    return out


@_on_device_of_first_tensor
def nhwc_to_nchw(x):
    x = _chk(x, 'x')
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), device=x.device, dtype=x.dtype)
    _native.check(_native.lib().pnp_nhwc_to_nchw_f32(_ptr(x), _ptr(out), n, c, h, w, _stream()), 'nhwc_to_nchw')
    return out


@_on_device_of_first_tensor
def caa_predict(q_ew, q_gamma, w1, b1, w2, b2, v1=None, v2=None, softmax=True):
    """Base_Predictor + SEModule (domain_aware.py:172-183, 210-222).
    q_ew/q_gamma: python sequences of floats (host).  Returns (ew (count,E), gamma (count,64))."""
    count = len(q_ew)
    E = w2.shape[0]
    dev = w1.device
    ew = torch.empty((count, E), device=dev, dtype=torch.float32)
    gamma = torch.empty((count, 64), device=dev, dtype=torch.float32)
    qa = (ctypes.c_float * count)(*[float(v) for v in q_ew])
    qg = (ctypes.c_float * count)(*[float(v) for v in q_gamma])
    ts = [_chk(t, 'param') for t in (w1, b1, w2, b2)]
    vs = [(_chk(v, 'param') if v is not None else None) for v in (v1, v2)]
    _native.check(_native.lib().pnp_caa_predict_f32(qa, qg, count, E, int(bool(softmax)), _ptr(ts[0]), _ptr(ts[1]),
                                                    _ptr(ts[2]), _ptr(ts[3]), _ptr(vs[0]), _ptr(vs[1]), _ptr(ew),
                                                    _ptr(gamma), _stream()), 'pnp_caa_predict_f32')
    return ew, gamma


@_on_device_of_first_tensor
def pack_conv3x3(weight, cbase=0, csrc=64, ew=None):
    """OIHW (cout,cin,3,3) [or (E,cout,cin,3,3) with ew (E,)] -> packed image for input
    channels [cbase, cbase+csrc)."""
    weight = _chk(weight, 'weight')
    E = 1
    if weight.dim() == 5:
        E = weight.shape[0]
        ew = _chk(ew, 'ew')
    cout, cin = weight.shape[-4], weight.shape[-3]
    L = _native.lib()
    dst = torch.empty(int(L.pnp_packed_conv_floats(64 if csrc == 64 else 4)), device=weight.device, dtype=torch.float32)
    _native.check(L.pnp_pack_conv3x3_f32(_ptr(weight), _ptr(ew), E, cout, cin, cbase, csrc, _ptr(dst), _stream()),
                  'pnp_pack_conv3x3_f32')
    return dst


@_on_device_of_first_tensor
def pack_conv1x1(weights):
    """three (64,64,1,1) weights -> one tensor of 3 chunks (conv16x16, conv16x8, conv8x8)."""
    L = _native.lib()
    n = int(L.pnp_packed_conv_floats(4))
    dst = torch.empty(3 * n, device=weights[0].device, dtype=torch.float32)
    for j, w in enumerate(weights):
        w = _chk(w, 'w1x1')
        _native.check(L.pnp_pack_conv1x1_f32(_ptr(w), ctypes.c_void_p(dst.data_ptr() + 4 * n * j), _stream()),
                      'pnp_pack_conv1x1_f32')
    return dst


@_on_device_of_first_tensor
def par_tile_flags(par):
    """(3,h,w) partition planes -> int32 (tiles_y, tiles_x) of 8x16 tiles: bit j set iff plane j is nonzero in the tile."""
    par = _chk(par, 'par')
    if par.dim() != 3 or par.shape[0] != 3:
        raise ValueError('par must be (3,h,w)')
    h, w = par.shape[1:]
    out = torch.empty(((h + 7) // 8, (w + 15) // 16), device=par.device, dtype=torch.int32)
    _native.check(_native.lib().pnp_par_tile_flags_f32(_ptr(par), ctypes.c_void_p(out.data_ptr()), h, w, _stream()),
                  'pnp_par_tile_flags_f32')
    return out


@_on_device_of_first_tensor
def f16_image(packed):
    """fp32 packed weight image (whole chunks) -> fp16 image for conv3x3(..., fp16=True)."""
    packed = _chk(packed, 'packed')
    nchunks, rem = divmod(packed.numel(), 4096)
    if rem:
        raise ValueError('packed image must be whole 4096-float chunks')
    dst = torch.empty(packed.numel(), device=packed.device, dtype=torch.float16)
    _native.check(_native.lib().pnp_f16_image_from_f32(_ptr(packed), ctypes.c_void_p(dst.data_ptr()), nchunks, _stream()),
                  'pnp_f16_image_from_f32')
    return dst


@_on_device_of_first_tensor
def f16x3_image(packed):
    """fp32 packed weight image -> its split image for conv3x3_f16x3: hi = fp16(w), lo = fp16((w - hi) * 2048), interleaved per
    k-step (twice the halfs of f16_image)."""
    packed = _chk(packed, 'packed')
    nchunks, rem = divmod(packed.numel(), 4096)
    if rem:
        raise ValueError('packed image must be whole 4096-float chunks')
    dst = torch.empty(2 * packed.numel(), device=packed.device, dtype=torch.float16)
    _native.check(_native.lib().pnp_f16x3_image_from_f32(_ptr(packed), ctypes.c_void_p(dst.data_ptr()), nchunks, _stream()),
                  'pnp_f16x3_image_from_f32')
    return dst


@_on_device_of_first_tensor
def conv3x3_f16x3(srcs, packed_w, bias=None, gamma=None, packed_w1x1=None, par=None, par_flags=None, residual=None, act=0,
                  trace=None, scaled_w1x1=False, tile_queue=None):
    """conv3x3 in split fp16 (pnp_conv3x3_f16x3, PNP_PREC_F16X3): packed_w / packed_w1x1 are the fp32 images of conv3x3 (their
    split images are made here) or, for 64-channel sources / the 1x1 branches, float16 tensors that already are f16x3_image()
    results.  fp32 sources, fp32 result at fp32-level accuracy.  tile_queue: 16 zeroed int32 (the generator's per-context queue:
    blocks draw their tiles from it instead of walking a static share; left zeroed by every launch)."""
    srcs = [_chk(s, 'src') for s in srcs]
    h, w = srcs[0].shape[:2]
    n = len(srcs)

    def split(p, wide=True):
        if p is None or not wide:
            return None
        if p.dtype == torch.float16:
            if not p.is_cuda or not p.is_contiguous():
                raise TypeError('split weight images must be contiguous CUDA float16 tensors (ops.f16x3_image)')
            return p
        return f16x3_image(p)

    # the RGB frame's image is split too (k = 4 tap + channel: two 32-deep chunks): with it the whole conv is ONE launch; the fp32
    # image still travels for the traced variant, where the RGB link stays on the exact fp32 kernel
    x3 = [split(p) for p in packed_w]
    # scaled_w1x1 (include/pnpvcve_debug.h): the branch images followed by the same images x 1/255, as pnp_generator_pack lays them
    # out -- tiles whose partition values are all 0 or exactly 1/255 (par_flags bits 3..5) then take the masked-operand fast path
    if scaled_w1x1:
        if packed_w1x1 is None or packed_w1x1.dtype != torch.float32:
            raise TypeError('scaled_w1x1: packed_w1x1 must be the fp32 images of pack_conv1x1')
        packed_w1x1 = torch.cat([packed_w1x1.reshape(-1), packed_w1x1.reshape(-1) * (torch.ones((), device=packed_w1x1.device) / 255.0)])
    p_x3 = split(packed_w1x1)
    packed_w = [(_chk(p, 'packed_w') if p.dtype == torch.float32 else None) for p in packed_w]
    vp = lambda ts: (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts])
    out = torch.empty((h, w, 64), device=srcs[0].device, dtype=torch.float32)
    sc = (ctypes.c_int * n)(*[s.shape[2] for s in srcs])
    keep = [(_chk(t, 'arg') if t is not None else None) for t in (bias, gamma, par, residual)]
    one = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    if tile_queue is not None and (tile_queue.dtype != torch.int32 or tile_queue.numel() < 16 or not tile_queue.is_cuda):
        raise TypeError('tile_queue: 16 int32 on the device')
    if trace is None and not scaled_w1x1 and tile_queue is None:
        _native.check(_native.lib().pnp_conv3x3_f16x3(n, vp(srcs), sc, vp(packed_w), vp(x3), _ptr(keep[0]), _ptr(keep[1]),
                                                      one(p_x3), _ptr(keep[2]), one(par_flags), _ptr(keep[3]), act,
                                                      _ptr(out), h, w, _stream()), 'pnp_conv3x3_f16x3')
    else:       # include/pnpvcve_debug.h: in-kernel timeline, scaled branch images, tile queue
        _native.check(_native.lib().pnp_conv3x3_f16x3_ex(n, vp(srcs), sc, vp(packed_w), vp(x3), _ptr(keep[0]), _ptr(keep[1]),
                                                         one(p_x3), _ptr(keep[2]), one(par_flags), _ptr(keep[3]), act,
                                                         _ptr(out), h, w, int(bool(scaled_w1x1)), one(tile_queue), one(trace), _stream()),
                      'pnp_conv3x3_f16x3_ex')
    return out


@_on_device_of_first_tensor
def conv3x3(srcs, packed_w, bias=None, gamma=None, packed_w1x1=None, par=None, residual=None, act=0, fp16=False,
            variant=None, par_flags=None, trace=None):
    """Fused conv over pixel-major sources [(h,w,64) or (h,w,4)].  See include/pnpvcve.h.
    fp16=True: packed_w / packed_w1x1 are f16_image() tensors and the MFMA operands are fp16.
    variant / par_flags / trace: include/pnpvcve_debug.h (kernel selection, per-tile branch flags, timeline buffer)."""
    if fp16:
        if variant is not None or par_flags is not None:
            raise ValueError('conv3x3(fp16=True): `variant` and `par_flags` select among the fp32 kernels only '
                             '(pnp_conv3x3_f16_ex takes neither)')
        return _conv3x3_f16(srcs, packed_w, bias, gamma, packed_w1x1, par, residual, act, trace)
    srcs = [_chk(s, 'src') for s in srcs]
    h, w = srcs[0].shape[:2]
    n = len(srcs)
    out = torch.empty((h, w, 64), device=srcs[0].device, dtype=torch.float32)
    sp = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    sc = (ctypes.c_int * n)(*[s.shape[2] for s in srcs])
    wp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in packed_w])
    keep = [(_chk(t, 'arg') if t is not None else None) for t in (bias, gamma, packed_w1x1, par, residual)]
    if variant is None and par_flags is None and trace is None:
        _native.check(_native.lib().pnp_conv3x3_f32(n, sp, sc, wp, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]),
                                                    _ptr(keep[3]), _ptr(keep[4]), act, _ptr(out), h, w, _stream()),
                      'pnp_conv3x3_f32')
    else:
        _native.check(_native.lib().pnp_conv3x3_f32_ex(n, sp, sc, wp, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]),
                                                       _ptr(keep[3]), _ptr(keep[4]), act, _ptr(out), h, w,
                                                       int(variant or 0), _ptr(par_flags), _ptr(trace), _stream()),
                      'pnp_conv3x3_f32_ex')
    return out


def _conv3x3_f16(srcs, packed_w, bias, gamma, packed_w1x1, par, residual, act, trace=None):
    srcs = [_chk(s, 'src') for s in srcs]
    for p in list(packed_w) + ([packed_w1x1] if packed_w1x1 is not None else []):
        if not p.is_cuda or p.dtype != torch.float16 or not p.is_contiguous():
            raise TypeError('fp16 conv: weight images must be contiguous CUDA float16 tensors (ops.f16_image)')
    h, w = srcs[0].shape[:2]
    n = len(srcs)
    out = torch.empty((h, w, 64), device=srcs[0].device, dtype=torch.float32)
    sp = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    sc = (ctypes.c_int * n)(*[s.shape[2] for s in srcs])
    wp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in packed_w])
    keep = [(_chk(t, 'arg') if t is not None else None) for t in (bias, gamma, par, residual)]
    w1 = ctypes.c_void_p(packed_w1x1.data_ptr()) if packed_w1x1 is not None else None
    _native.check(_native.lib().pnp_conv3x3_f16_ex(n, sp, sc, wp, _ptr(keep[0]), _ptr(keep[1]), w1, _ptr(keep[2]),
                                                   _ptr(keep[3]), act, _ptr(out), h, w, _ptr(trace), _stream()),
                  'pnp_conv3x3_f16')
    return out


@_on_device_of_first_tensor
def conv3x3_f16_maps(srcs, packed_w, bias=None, gamma=None, packed_w1x1=None, par=None, par_flags=None, residual=None, act=0,
                     out_f16=False, mirror=False, chain=False, trace=None):
    """The fp16-operand conv with explicit fp16 maps (include/pnpvcve_debug.h, pnp_conv3x3_f16_maps): a source of dtype
    float16 (h,w,64) is read as an fp16 map (the mirror its producer wrote), float32 sources are rounded on the fly.
    out_f16: the output is an fp16 map; mirror: additionally return the fp16 copy of the fp32 output written in the same
    pass -> (out, out16).  chain=True forces the chain of single-source launches for several fp32 sources."""
    n = len(srcs)
    mask = 0
    for i, t in enumerate(srcs):
        if not isinstance(t, torch.Tensor) or not t.is_cuda or not t.is_contiguous():
            raise RuntimeError('sources must be contiguous CUDA/HIP tensors')
        if t.dtype == torch.float16:
            mask |= 1 << i
        elif t.dtype != torch.float32:
            raise TypeError(f'source {i}: float32 or float16, got {t.dtype}')
    for p in list(packed_w) + ([packed_w1x1] if packed_w1x1 is not None else []):
        if not p.is_cuda or p.dtype != torch.float16 or not p.is_contiguous():
            raise TypeError('fp16 conv: weight images must be contiguous CUDA float16 tensors (ops.f16_image)')
    h, w = srcs[0].shape[:2]
    dev = srcs[0].device
    out = torch.empty((h, w, 64), device=dev, dtype=torch.float16 if out_f16 else torch.float32)
    out16 = torch.empty((h, w, 64), device=dev, dtype=torch.float16) if mirror else None
    sp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    sc = (ctypes.c_int * n)(*[t.shape[2] for t in srcs])
    wp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in packed_w])
    keep = [(_chk(t, 'arg') if t is not None else None) for t in (bias, gamma, par, residual)]
    if par_flags is not None and (par_flags.dtype != torch.int32 or not par_flags.is_cuda):
        raise TypeError('par_flags: int32 CUDA tensor (ops.par_tile_flags)')
    w1 = ctypes.c_void_p(packed_w1x1.data_ptr()) if packed_w1x1 is not None else None
    _native.check(_native.lib().pnp_conv3x3_f16_maps(n, sp, sc, mask, wp, _ptr(keep[0]), _ptr(keep[1]), w1, _ptr(keep[2]),
                                                     _ptr(par_flags), _ptr(keep[3]), act, _ptr(out), int(bool(out_f16)),
                                                     _ptr(out16), h, w, int(bool(chain)), _ptr(trace), _stream()),
                  'pnp_conv3x3_f16_maps')
    return (out, out16) if mirror else out


@_on_device_of_first_tensor
def mv_warp_nhwc_f16(feat, flow_x, flow_y):
    """mv_warp_nhwc writing an fp16 (h,w,c) map: saturating round-to-nearest-even of the fp32 result."""
    feat, flow_x, flow_y = _chk(feat, 'feat'), _chk(flow_x, 'flow_x'), _chk(flow_y, 'flow_y')
    h, w, c = feat.shape
    out = torch.empty((h, w, c), device=feat.device, dtype=torch.float16)
    _native.check(_native.lib().pnp_mv_warp_nhwc_f16out(_ptr(feat), _ptr(flow_x), _ptr(flow_y), _ptr(out), h, w, c, _stream()),
                  'pnp_mv_warp_nhwc_f16out')
    return out


@_on_device_of_first_tensor
def wino_image(packed, gamma=None):
    """Winograd F(2x2,3x3) image (65536 floats) of a packed 64 -> 64 direct-conv image (pack_conv3x3), times gamma[co] if given."""
    packed = _chk(packed, 'packed')
    if packed.numel() != 9 * 4096:
        raise ValueError('packed must be a 9-chunk 64-channel image')
    out = torch.empty(int(_native.lib().pnp_wino_image_floats()), device=packed.device, dtype=torch.float32)
    _native.check(_native.lib().pnp_wino_image_from_packed_f32(_ptr(packed), _ptr(_chk(gamma, 'gamma')) if gamma is not None else None,
                                                               _ptr(out), _stream()), 'pnp_wino_image_from_packed_f32')
    # which gain the image carries: conv3x3_wino scales only the BIAS by its `gamma` argument (the conv term's gain lives in the image), so
    # the two must be the same tensor -- checked there when the image still carries this tag
    out._pnp_gamma = None if gamma is None else gamma.detach().clone()
    return out


@_on_device_of_first_tensor
def wino_par_image(packed_w1x1):
    packed_w1x1 = _chk(packed_w1x1, 'packed_w1x1')
    out = torch.empty(int(_native.lib().pnp_wino_par_image_floats()), device=packed_w1x1.device, dtype=torch.float32)
    _native.check(_native.lib().pnp_wino_par_image_from_packed_f32(_ptr(packed_w1x1), _ptr(out), _stream()),
                  'pnp_wino_par_image_from_packed_f32')
    return out


@_on_device_of_first_tensor
def wino_rgb_image(packed_rgb):
    """Winograd image (4096 floats) of the frame's packed chunk (pack_conv3x3(w, cbase=0, csrc=3))."""
    packed_rgb = _chk(packed_rgb, 'packed_rgb')
    out = torch.empty(int(_native.lib().pnp_wino_rgb_image_floats()), device=packed_rgb.device, dtype=torch.float32)
    _native.check(_native.lib().pnp_wino_rgb_image_from_packed_f32(_ptr(packed_rgb), _ptr(out), _stream()),
                  'pnp_wino_rgb_image_from_packed_f32')
    return out


@_on_device_of_first_tensor
def conv3x3_wino_ms(srcs, wino_ws, bias=None, act=0, units=False):
    """The input conv on conv_wino.hip: srcs = [(h,w,4) frame, (h,w,64) maps ...], wino_ws = [wino_rgb_image(...), wino_image(...) ...];
    the 64-channel sources' images must sit within 4 GiB of each other (slices of ONE tensor do)."""
    srcs = [_chk(x, 'src') for x in srcs]
    wino_ws = [_chk(x, 'wino_w') for x in wino_ws]
    h, w = srcs[0].shape[:2]
    n = len(srcs)
    if n != len(wino_ws) or srcs[0].shape[2] != 4 or any(x.shape != (h, w, 64) for x in srcs[1:]):
        raise ValueError('srcs must be [(h,w,4), (h,w,64) ...] with one Winograd image each')
    out = torch.empty((h, w, 64), device=srcs[0].device, dtype=torch.float32)
    sp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in srcs])
    wp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in wino_ws])
    fn = _native.lib().pnp_conv3x3_wino_ms_units_f32 if units else _native.lib().pnp_conv3x3_wino_ms_f32
    _native.check(fn(n, sp, wp, _ptr(_chk(bias, 'bias')) if bias is not None else None, int(act), _ptr(out), h, w, _stream()),
                  'pnp_conv3x3_wino_ms_f32')
    return out


@_on_device_of_first_tensor
def conv3x3_wino(x, wino_w, bias=None, gamma=None, wino_w1x1=None, par=None, par_flags=None, residual=None, act=0, trace=None, units=False):
    """act(gamma * (conv3x3(x; W) + bias) + sum_j par_j * conv1x1_j(x)) + residual on conv_wino.hip; x (h,w,64) NHWC fp32,
    wino_w = wino_image(packed W, gamma) -- the SAME gamma -- and wino_w1x1 = wino_par_image(packed 1x1 images).  units=True: one block
    per 8x8 quadrant unit (the small-frame form, pnp_conv3x3_wino_units_f32; same values)."""
    x = _chk(x, 'x')
    h, w, c = x.shape
    if c != 64:
        raise ValueError('x must be (h, w, 64)')
    tag = getattr(wino_w, '_pnp_gamma', 'untagged')              # (set by wino_image; lost by views / copies: then nothing is checked)
    if not isinstance(tag, str) and ((tag is None) != (gamma is None) or (gamma is not None and not torch.equal(tag, gamma))):
        raise ValueError('conv3x3_wino: `gamma` scales only the bias -- the conv term carries the gain wino_image() folded into wino_w; '
                         'pass the SAME gamma tensor to both (or none to both)')
    out = torch.empty_like(x)
    opt = lambda t, n: _ptr(_chk(t, n)) if t is not None else None   # noqa: E731
    if par_flags is not None and (par_flags.dtype != torch.int32 or not par_flags.is_cuda):
        raise ValueError('par_flags must be a CUDA int32 tensor')
    args = (_ptr(x), _ptr(_chk(wino_w, 'wino_w')), opt(bias, 'bias'), opt(gamma, 'gamma'), opt(wino_w1x1, 'wino_w1x1'), opt(par, 'par'),
            ctypes.c_void_p(par_flags.data_ptr()) if par_flags is not None else None, opt(residual, 'residual'), int(act), _ptr(out), h, w)
    if units:
        _native.check(_native.lib().pnp_conv3x3_wino_units_f32(*args, _stream()), 'pnp_conv3x3_wino_units_f32')
    elif trace is None:
        _native.check(_native.lib().pnp_conv3x3_wino_f32(*args, _stream()), 'pnp_conv3x3_wino_f32')
    else:       # include/pnpvcve_debug.h: 16 uint64 per block
        _native.check(_native.lib().pnp_conv3x3_wino_f32_ex(*args, ctypes.c_void_p(trace.data_ptr()), _stream()), 'pnp_conv3x3_wino_f32_ex')
    return out


@_on_device_of_first_tensor
def frames_to_rgb8(frames):
    """(n,3,h,w) fp32 CUDA frames -> (n,h,w,3) uint8 RGB CUDA tensor with tensor2img's arithmetic."""
    frames = _chk(frames, 'frames')
    if frames.dim() != 4 or frames.shape[1] != 3:
        raise ValueError('frames must be (n,3,h,w)')
    n, _, h, w = frames.shape
    out = torch.empty((n, h, w, 3), device=frames.device, dtype=torch.uint8)
    _native.check(_native.lib().pnp_frames_to_rgb8(_ptr(frames), ctypes.c_void_p(out.data_ptr()), n, h, w, _stream()),
                  'pnp_frames_to_rgb8')
    return out


@_on_device_of_first_tensor
def psnr_frames(a, b, crop_border=0):
    """Per-frame PSNR with the reference's definition (uint8-rounded frames), computed on the GPU.
    a, b: (..., c, h, w) with any leading dims; returns a float64 CPU tensor of the leading shape."""
    a, b = _chk(a, 'a'), _chk(b, 'b')
    if a.shape != b.shape:
        raise AssertionError(f'Image shapes are different: {tuple(a.shape)}, {tuple(b.shape)}.')
    c, h, w = a.shape[-3:]
    frames = a.numel() // (c * h * w)
    sse = torch.empty(frames, dtype=torch.int64, device=a.device)
    _native.check(_native.lib().pnp_psnr_sse_f32(_ptr(a), _ptr(b), _ptr(sse), frames, c, h, w, int(crop_border),
                                                 _stream()), 'pnp_psnr_sse_f32')
    n = c * (h - 2 * crop_border) * (w - 2 * crop_border)
    mse = sse.cpu().double() / n
    out = 20.0 * torch.log10(255.0 / mse.sqrt())
    out[mse == 0] = float('inf')
    return out.reshape(a.shape[:-3])


@_on_device_of_first_tensor
def rasterise_side_info(records, rec_frame, slices, h, w):
    """Decoder MV records -> (mvs (T,4,h,w), partitions (T,3,h,w)) on the GPU
    (LoadImageFromFileList_ipb.__call__, loading_ipb.py:328-369, + RescaleToZeroOne + FramesToTensor).
    records (R,10) fp32 CUDA, rec_frame (R,) int32 CUDA, slices: sequence of 'I'/'P'/'B' (or ord values)."""
    records = _chk(records, 'records')
    if not rec_frame.is_cuda or rec_frame.dtype != torch.int32:
        raise TypeError('rec_frame must be a CUDA int32 tensor')
    rec_frame = rec_frame.contiguous()
    t = len(slices)
    sl = (ctypes.c_float * t)(*[float(ord(s) if isinstance(s, str) else s) for s in slices])
    dev = records.device
    mvs = torch.empty((t, 4, h, w), device=dev, dtype=torch.float32)
    par = torch.empty((t, 3, h, w), device=dev, dtype=torch.float32)
    scratch = torch.empty((t, 2, h, w), device=dev, dtype=torch.int32)
    _native.check(_native.lib().pnp_rasterise_side_info_f32(_ptr(records), _ptr(rec_frame), records.shape[0], sl, t, h, w,
                                                            _ptr(mvs), _ptr(par), _ptr(scratch), _stream()),
                  'pnp_rasterise_side_info_f32')
    return mvs, par


@_on_device_of_first_tensor
def modulated_deform_conv_nhwc(x, offset, mask_logits, weight, bias, flow=None, fp16=False):
    """mmcv.ops.modulated_deform_conv2d(x, offset, sigmoid(mask_logits), weight, bias, 1, 1, 1, 1, 16) for the
    hot path's shapes.  x (h,w,64) pixel-major; offset (288,h,w) and mask_logits (144,h,w) in mmcv's channel
    order; weight (64,64,3,3); optional flow (2,h,w) = (dx,dy) added to every offset.  Returns (h,w,64).
    fp16=True: fp16 MFMA operands (what PNP_PREC_F16 runs), fp32 accumulation."""
    x = _chk(x, 'x')
    h, w, _ = x.shape
    L = _native.lib()
    ref = torch.tensor([L.pnp_dcn_ref_channel(c) for c in range(448)], device=x.device)
    src = torch.cat([_chk(offset, 'offset'), _chk(mask_logits, 'mask')], 0)             # (432,h,w)
    om = torch.zeros((448, h, w), device=x.device, dtype=torch.float32)
    om[ref >= 0] = src[ref[ref >= 0]]
    om = om.permute(1, 2, 0).contiguous()
    wp = pack_conv3x3(weight)
    out = torch.empty_like(x)
    fx = _chk(flow[0], 'flow') if flow is not None else None
    fy = _chk(flow[1], 'flow') if flow is not None else None
    if fp16:
        w16 = torch.empty(9 * 4096, device=x.device, dtype=torch.float16)
        _native.check(L.pnp_dcn_f16_image_from_f32(_ptr(wp), ctypes.c_void_p(w16.data_ptr()), _stream()),
                      'pnp_dcn_f16_image_from_f32')
        _native.check(L.pnp_dcn_nhwc_f16(_ptr(x), _ptr(om), _ptr(fx), _ptr(fy), ctypes.c_void_p(w16.data_ptr()),
                                         _ptr(_chk(bias, 'bias')), _ptr(out), h, w, _stream()), 'pnp_dcn_nhwc_f16')
        return out
    _native.check(L.pnp_dcn_nhwc_f32(_ptr(x), _ptr(om), _ptr(fx), _ptr(fy), _ptr(wp), _ptr(_chk(bias, 'bias')), _ptr(out),
                                     h, w, _stream()), 'pnp_dcn_nhwc_f32')
    return out


@_on_device_of_first_tensor
def ssim_frames(a, b, crop_border=0):
    """Per-frame SSIM with the reference's definition (metrics.py:266-355), computed on the GPU in fp64.
    a, b: (..., c, h, w); returns a float64 CPU tensor of the leading shape."""
    a, b = _chk(a, 'a'), _chk(b, 'b')
    if a.shape != b.shape:
        raise AssertionError(f'Image shapes are different: {tuple(a.shape)}, {tuple(b.shape)}.')
    c, h, w = a.shape[-3:]
    frames = a.numel() // (c * h * w)
    L = _native.lib()
    nb = int(L.pnp_ssim_blocks(h, w, int(crop_border)))
    if nb < 1:
        raise ValueError('frames are smaller than the 11x11 SSIM window')
    part = torch.empty((frames * c, nb), dtype=torch.float64, device=a.device)
    _native.check(L.pnp_ssim_partials_f32(_ptr(a), _ptr(b), _ptr(part), frames, c, h, w, int(crop_border), _stream()),
                  'pnp_ssim_partials_f32')
    n = (h - 2 * crop_border - 10) * (w - 2 * crop_border - 10)
    per_plane = (part.sum(dim=1) / n).reshape(frames, c)
    if crop_border != 0 and c == 3:
        # reference quirk (metrics.py:343-350, see pnp_vcve_amd/metrics.py): with a crop only channel 0 of the BGR image,
        # i.e. the B plane of these RGB frames, enters the SSIM mean
        per_plane = per_plane[:, 2:3]
    return per_plane.mean(dim=1).cpu().reshape(a.shape[:-3])


@_on_device_of_first_tensor
def bae_block(x, w2_packed, b2, gamma, w1x1_packed, par, w1_packed, b1):
    """One BAE block (ResidualBlockNoBNDynamic_drt.forward, sr_backbone_utils.py:305-313,329), pnp_bae_block_f32.
    x (h,w,64) pixel-major; w2_packed = pack_conv3x3(experts, ew=attention); w1_packed = pack_conv3x3(conv1.weight);
    w1x1_packed = pack_conv1x1([...]) and par (3,h,w), or both None; gamma (64,) or None."""
    x = _chk(x, 'x')
    h, w, _ = x.shape
    keep = [(_chk(t, 'arg') if t is not None else None) for t in (w2_packed, b2, gamma, w1x1_packed, par, w1_packed, b1)]
    scratch = torch.empty_like(x)
    out = torch.empty_like(x)
    _native.check(_native.lib().pnp_bae_block_f32(_ptr(x), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]),
                                                  _ptr(keep[4]), _ptr(keep[5]), _ptr(keep[6]), _ptr(scratch), _ptr(out),
                                                  h, w, _stream()), 'pnp_bae_block_f32')
    return out


@_on_device_of_first_tensor
def pack_pixel_shuffle(weight, bias):
    """PixelShufflePack.upsample_conv (256,64,3,3) + bias (256) -> packed image for pixel_shuffle_conv."""
    weight, bias = _chk(weight, 'weight'), _chk(bias, 'bias')
    if tuple(weight.shape) != (256, 64, 3, 3) or tuple(bias.shape) != (256,):
        raise ValueError('pixel-shuffle conv: weight (256,64,3,3), bias (256,)')
    L = _native.lib()
    dst = torch.empty(int(L.pnp_packed_pixel_shuffle_floats()), device=weight.device, dtype=torch.float32)
    _native.check(L.pnp_pack_pixel_shuffle_f32(_ptr(weight), _ptr(bias), _ptr(dst), _stream()), 'pnp_pack_pixel_shuffle_f32')
    return dst


@_on_device_of_first_tensor
def pixel_shuffle_conv(x, packed, act=0):
    """PixelShufflePack.forward (upsample.py:40-51): x (h,w,64) -> (2h,2w,64); act 0 none / 1 relu / 2 leaky-relu(0.1)."""
    x, packed = _chk(x, 'x'), _chk(packed, 'packed')
    h, w, _ = x.shape
    out = torch.empty((2 * h, 2 * w, 64), device=x.device, dtype=torch.float32)
    _native.check(_native.lib().pnp_pixel_shuffle_conv_f32(_ptr(x), _ptr(packed), int(act), _ptr(out), h, w, _stream()),
                  'pnp_pixel_shuffle_conv_f32')
    return out

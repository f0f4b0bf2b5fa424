#!/usr/bin/env python
"""Standalone timing of every fp16-operand conv variant of a BAE block / input conv at a given frame size (HIP events, warm
caches defeated by rotating over several feature maps), plus the in-kernel timeline of the persistent kernel.

    python tools/bench_f16_block.py [H W] [--trace]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pnp_vcve_amd import ops  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
h, w = (int(args[0]), int(args[1])) if len(args) >= 2 else (720, 1280)
dev = torch.device('cuda:0')
torch.manual_seed(0)
NB = 4                                  # rotate over NB independent maps so that nothing stays in the 256 MB infinity cache by luck
xs = [torch.randn(h, w, 64, device=dev) for _ in range(NB)]
xs16 = [x.half() for x in xs]
rs = [torch.randn(h, w, 64, device=dev) for _ in range(NB)]
pw = ops.f16_image(ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05))
p1 = ops.f16_image(ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)]))
PW32 = ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
P132 = ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)])
cls = torch.randint(0, 3, ((h + 7) // 8, (w + 7) // 8), device=dev)
par = torch.stack([(cls == j).float() for j in range(3)]).repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :w].contiguous() / 255.0
flags = ops.par_tile_flags(par)
bias = torch.randn(64, device=dev) * 0.1
gam = torch.rand(64, device=dev) + 0.5
wg = torch.randn(64, 3 + 64 * 3, 3, 3, device=dev) * 0.03
lr4 = torch.rand(h, w, 4, device=dev)
lr4[..., 3] = 0
pk_lr = ops.f16_image(ops.pack_conv3x3(wg, 0, 3))
pk_w = [ops.f16_image(ops.pack_conv3x3(wg, 3 + 64 * j, 64)) for j in range(3)]


def timeit(name, fn, bytes_alg=None, reps=40):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    extra = f'  {bytes_alg * h * w / us / 1e6:6.2f} TB/s of {bytes_alg} B/px' if bytes_alg else ''
    print(f'{name:78s} {us:8.1f} us{extra}', flush=True)
    return us


F = ops.conv3x3_f16_maps
print(f'frame {h}x{w}, tiles {((h + 7) // 8) * ((w + 15) // 16)}; branches needed per tile: '
      f'{sum(((flags >> j) & 1).float().mean().item() for j in range(3)):.2f}')
kw = dict(bias=bias, gamma=gam, packed_w1x1=p1, par=par, act=1)
timeit('front  r02: fp32 x -> fp16 o, all branches', lambda i: F([xs[i % NB]], [pw], out_f16=True, **kw), 396)
timeit('front     : fp32 x -> fp16 o, branch skipping', lambda i: F([xs[i % NB]], [pw], out_f16=True, par_flags=flags, **kw), 396)
timeit('front     : fp16 x -> fp16 o, all branches', lambda i: F([xs16[i % NB]], [pw], out_f16=True, **kw), 268)
timeit('front     : fp16 x -> fp16 o, all branches, resident-weight kernel', lambda i: F([xs16[i % NB]], [pw], out_f16=True, **kw), 268)
timeit('front     : fp16 x -> fp16 o, branch skipping, resident-weight kernel', lambda i: F([xs16[i % NB]], [pw], out_f16=True, par_flags=flags, **kw), 268)
timeit('back   r02: fp16 o + fp32 residual -> fp32 x, resident-weight kernel', lambda i: F([xs16[i % NB]], [pw], bias=bias, residual=rs[i % NB]), 640)
timeit('back      : ... + fp16 mirror, resident-weight kernel', lambda i: F([xs16[i % NB]], [pw], bias=bias, residual=rs[i % NB], mirror=True), 768)
timeit('conv_hr r02: fp32 -> fp16', lambda i: F([xs[i % NB]], [pw], bias=bias, act=2, out_f16=True), 384)
timeit('conv_hr    : fp16 -> fp16, resident-weight kernel', lambda i: F([xs16[i % NB]], [pw], bias=bias, act=2, out_f16=True), 256)
# split fp16 (PNP_PREC_F16X3): fp32 maps both ways, three MFMAs per product
pw3 = ops.f16x3_image(ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05))
p13 = ops.f16x3_image(ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)]))
X3 = ops.conv3x3_f16x3
timeit('front  f16x3: fp32 x -> fp32 o, all branches', lambda i: X3([xs[i % NB]], [pw3], bias=bias, gamma=gam, packed_w1x1=p13, par=par, act=1), 524)
timeit('front  f16x3: fp32 x -> fp32 o, branch skipping', lambda i: X3([xs[i % NB]], [pw3], bias=bias, gamma=gam, packed_w1x1=p13, par=par, par_flags=flags, act=1), 524)
timeit('back   f16x3: fp32 o + fp32 residual -> fp32 x', lambda i: X3([xs[i % NB]], [pw3], bias=bias, residual=rs[i % NB]), 768)
timeit('conv_hr f16x3: fp32 -> fp32', lambda i: X3([xs[i % NB]], [pw3], bias=bias, act=2), 512)
timeit('front  fp32 (exact fp32 MFMA kernel), branch skipping', lambda i: ops.conv3x3([xs[i % NB]], [PW32], bias=bias, gamma=gam, packed_w1x1=P132, par=par, par_flags=flags, act=1), 524)
timeit('back   fp32 (exact fp32 MFMA kernel)', lambda i: ops.conv3x3([xs[i % NB]], [PW32], bias=bias, residual=rs[i % NB]), 768)
for nw in (1, 2, 3):
    s32 = [xs[(j + 1) % NB] for j in range(nw)]
    s16 = [xs16[(j + 1) % NB] for j in range(nw)]
    timeit(f'input conv r02: rgb + {nw} fp32 sources, launch chain -> fp32', lambda i: F([lr4] + s32, [pk_lr] + pk_w[:nw], bias=bias, act=2, chain=True))
    timeit(f'input conv r03: rgb + {nw} fp16 sources, ONE launch -> fp32 + fp16 mirror', lambda i: F([lr4] + s16, [pk_lr] + pk_w[:nw], bias=bias, act=2, mirror=True),
           16 + 128 * nw + 384)
fx = (torch.randint(-32, 33, ((h + 7) // 8, (w + 7) // 8), device=dev).float() / 4).repeat_interleave(8, 0).repeat_interleave(8, 1)[:h, :w].contiguous()
timeit('mv warp r02: fp32 out', lambda i: ops.mv_warp_nhwc(xs[i % NB], fx, fx), 520)
timeit('mv warp r03: fp16 out', lambda i: ops.mv_warp_nhwc_f16(xs[i % NB], fx, fx), 392)

if '--trace' in sys.argv:
    def trace(name, fn):
        dbg = torch.zeros(512 * 16, dtype=torch.int64, device=dev)
        for i in range(3):
            fn(None)
        fn(dbg)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(512, 16)
        d = d[d[:, 7] > 0]
        n = d[:, 7]
        tot = d[:, 3] - d[:, 0]
        print(f'--- timeline {name}: groups {len(d)}, tiles/group {n.min()}..{n.max()}, kernel span {d[:, 3].max() - d[:, 0].min()} cycles')
        for nm, v in (('matrix phase / tile', d[:, 1] / n), ('memory: epilogue / tile', d[:, 2] / n), ('memory: rest / tile', d[:, 6] / n),
                      ('  of which halo wait+cvt+LDS', d[:, 8] / n), ('barrier wait / tile', d[:, 9] / n), ('total / tile and group', tot / n)):
            print(f'    {nm:30s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}')
    if ((h + 7) // 8) * ((w + 15) // 16) >= 1024:
        trace('front r02 (fp32 x, all branches)', lambda t: F([xs[0]], [pw], out_f16=True, trace=t, **kw))
        trace('front (fp16 x, skipping), resident-weight kernel', lambda t: F([xs16[0]], [pw], out_f16=True, par_flags=flags, trace=t, **kw))
        trace('back (+ mirror), resident-weight kernel', lambda t: F([xs16[0]], [pw], bias=bias, residual=rs[0], mirror=True, trace=t))

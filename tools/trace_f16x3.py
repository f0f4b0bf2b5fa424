#!/usr/bin/env python
"""In-kernel timeline of the split-fp16 conv kernel (conv_f16x3.hip) at a frame size: per 8x16 tile the cycles of the prologue
(fp32 halo -> two fp16 tiles), the K loop and the epilogue, and how many blocks shared a CU while each one ran.

    python tools/trace_f16x3.py [H W]
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
x = torch.randn(h, w, 64, device=dev)
r = torch.randn(h, w, 64, device=dev)
pw3 = ops.f16x3_image(ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05))
p13 = ops.f16x3_image(ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)]))
cls = torch.randint(0, 3, ((h + 7) // 8, (w + 7) // 8), device=dev)
par = torch.stack([(cls == j).float() for j in range(3)]).repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :w].contiguous() / 255.0
flags = ops.par_tile_flags(par)
bias = torch.randn(64, device=dev) * 0.1
gam = torch.rand(64, device=dev) + 0.5
tiles = ((h + 7) // 8) * ((w + 15) // 16)


QUEUE = torch.zeros(16, dtype=torch.int32, device=dev) if '--queue' in sys.argv else None    # the generator's tile queue


def trace(name, fn0):
    fn = lambda t: fn0(t, QUEUE)
    dbg = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        fn(None)
    fn(dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(512, 8)
    d = d[d[:, 5] > 0]
    n = d[:, 5]
    tot = d[:, 3] - d[:, 0]
    print(f'--- {name}: {tiles} tiles on {len(d)} persistent blocks, tiles per block {n.min()}..{n.max()}')
    for nm, v in (('A tiles: wait + split + LDS', d[:, 1] / n), ('K loop', d[:, 2] / n), ('epilogue', d[:, 4] / n), ('total per tile', tot / n)):
        print(f'    {nm:28s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p50 {np.percentile(v, 50):8.0f}  p90 {np.percentile(v, 90):8.0f}')
    print(f'    in-kernel clock: {np.median(tot / (d[:, 6] / 100.0)) / 1e3:.3f} GHz (s_memtime cycles per 100 MHz s_memrealtime tick)')
    base = d[:, 7] & 0xff          # HW_REG_LDS_ALLOC's LDS_BASE: 0 = the block that reached its CU first (it wins the issue arbitration)
    for bb in np.unique(base):
        m = base == bb
        print(f'    blocks at LDS base {int(bb):3d}: {int(m.sum()):3d}, tiles each {n[m].min()}..{n[m].max()}, K loop {np.mean(d[m, 2] / n[m]):6.0f}, '
              f'per tile {np.mean(tot[m] / n[m]):6.0f}, lifetime {tot[m].mean():.0f}')
    print(f'    block lifetime: mean {tot.mean():.0f} cycles, max {tot.max()}; 12 MFMAs x 32 cycles per wave and chunk = 384, 18 chunks = 6912')


trace('conv_hr-like (no branches, no residual)', lambda t, q: ops.conv3x3_f16x3([x], [pw3], bias=bias, act=2, trace=t, tile_queue=q))
trace('back half (+ residual)', lambda t, q: ops.conv3x3_f16x3([x], [pw3], bias=bias, residual=r, trace=t, tile_queue=q))
trace('front half (branch skipping)', lambda t, q: ops.conv3x3_f16x3([x], [pw3], bias=bias, gamma=gam, packed_w1x1=p13, par=par, par_flags=flags, act=1, trace=t, tile_queue=q))
p1_f32 = ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)])
par255 = (par * 255.0).round() * (torch.ones((), device=dev) / 255.0)          # exactly 0 or float32(1) / float32(255)
flags255 = ops.par_tile_flags(par255)
print('tiles on the binary-map fast path:', int((((flags255 >> 3) & flags255 & 7) == (flags255 & 7)).sum()), 'of', flags255.numel())
trace('front half, binary-map fast path (masked operand, weights x 1/255)',
      lambda t, q: ops.conv3x3_f16x3([x], [pw3], bias=bias, gamma=gam, packed_w1x1=p1_f32, par=par255, par_flags=flags255, act=1, trace=t, tile_queue=q,
                                  scaled_w1x1=True))
trace('front half, same map, general path',
      lambda t, q: ops.conv3x3_f16x3([x], [pw3], bias=bias, gamma=gam, packed_w1x1=p1_f32, par=par255, par_flags=flags255, act=1, trace=t, tile_queue=q))

#!/usr/bin/env python
"""Diagnostic: per-block timeline of the tile-per-block fp16 conv kernel (conv3x3_f16_small_kernel): prologue (halo + first
weight chunks -> LDS), K loop, epilogue, in shader cycles; argv: h w mode(par|res|plain)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pnp_vcve_amd import ops  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (180, 320)
mode = sys.argv[3] if len(sys.argv) > 3 else 'res'
dev = torch.device('cuda:0')
x = torch.randn(h, w, 64, device=dev)
x2 = torch.randn(h, w, 64, device=dev)
pw = ops.f16_image(ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05))
p1 = ops.f16_image(ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)]))
par = (torch.rand(3, h, w, device=dev) > 0.66).float() / 255.0
bias = torch.randn(64, device=dev) * 0.1
ntiles = ((w + 15) // 16) * ((h + 7) // 8)


def run(trace=None):
    if mode == 'res':
        return ops.conv3x3([x], [pw], bias=bias, residual=x2, fp16=True, trace=trace)
    if mode == 'par':
        return ops.conv3x3([x], [pw], bias=bias, packed_w1x1=p1, par=par, act=1, fp16=True, trace=trace)
    return ops.conv3x3([x], [pw], bias=bias, act=2, fp16=True, trace=trace)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print(f'{h}x{w} {mode}: {e0.elapsed_time(e1) * 100:.1f} us per launch, {ntiles} tiles')
dbg = torch.zeros((ntiles + 1) * 16, dtype=torch.int64, device=dev)
run(dbg)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 7] > 0].astype(np.float64)
if len(d):
    for name, v in (('prologue', d[:, 1] - d[:, 0]), ('K loop', d[:, 2] - d[:, 1]), ('epilogue', d[:, 3] - d[:, 2]),
                    ('block total', d[:, 3] - d[:, 0])):
        print(f'  {name:12s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f} cycles')
    print('  kernel span', d[:, 3].max() - d[:, 0].min(), 'cycles; blocks', len(d))
else:
    print('  (no timeline: the launch went to the persistent kernel)')

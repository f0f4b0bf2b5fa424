#!/usr/bin/env python
"""Diagnostic: per-block phase sums of the fp16-operand persistent conv kernel (shader-clock stamps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pnp_vcve_amd import _native, ops  # noqa: E402

h, w = 720, 1280
mode = sys.argv[1] if len(sys.argv) > 1 else 'res'
dev = torch.device('cuda:0')
x = torch.randn(h, w, 64, device=dev)
x2 = torch.randn(h, w, 64, device=dev)
pw = ops.f16_image(ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05))
p1 = ops.f16_image(ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)]))
par = (torch.rand(3, h, w, device=dev) > 0.66).float() / 255.0
bias = torch.randn(64, device=dev) * 0.1


cls = torch.randint(0, 3, ((h + 7) // 8, (w + 7) // 8), device=dev)
parb = torch.stack([(cls == j).float() for j in range(3)]).repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :w].contiguous() / 255.0
flags = ops.par_tile_flags(parb)
gam = torch.rand(64, device=dev) + 0.5
x16 = x.half()


def run(trace=None):
    if mode in ('front', 'front16'):       # the BAE front half as the pipeline runs it: fp16 `o` map out, block-class partition map + flags
        return ops.conv3x3_f16_maps([x16 if mode == 'front16' else x], [pw], bias=bias, gamma=gam, packed_w1x1=p1, par=parb,
                                    par_flags=flags, act=1, out_f16=True, trace=trace)
    if mode == 'hr':                       # conv_hr: fp32 slot in, fp16 map out
        return ops.conv3x3_f16_maps([x], [pw], bias=bias, act=2, out_f16=True, trace=trace)
    if mode == 'res':
        return ops.conv3x3([x], [pw], bias=bias, residual=x2, fp16=True, trace=trace)
    if mode == 'par':
        return ops.conv3x3([x], [pw], bias=bias, packed_w1x1=p1, par=par, act=1, fp16=True, trace=trace)
    return ops.conv3x3([x], [pw], bias=bias, act=2, fp16=True, trace=trace)


for _ in range(3):
    run()
dbg = torch.zeros(512 * 16, dtype=torch.int64, device=dev)
run(dbg)       # include/pnpvcve_debug.h
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(512, 16)      # one row per 4-wave group (2 per block)
d = d[d[:, 7] > 0]
n = d[:, 7]
tot = d[:, 3] - d[:, 0]
print('mode', mode, 'groups', len(d), 'tiles/group min/max', n.min(), n.max(), 'phases', d[:, 5].min(), d[:, 5].max())
for name, v in (('group total', tot), ('matrix phase / tile', d[:, 1] / n), ('memory phase: epilogue / tile', d[:, 2] / n),
                ('memory phase: rest / tile', d[:, 6] / n), ('  of which halo wait+cvt+LDS', d[:, 8] / n),
                ('barrier wait / tile', d[:, 9] / n), ('total / tile', tot / n)):
    print(f'{name:32s} mean {v.mean():10.0f}  p10 {np.percentile(v, 10):10.0f}  p90 {np.percentile(v, 90):10.0f}  max {v.max():10.0f}')

#!/usr/bin/env python
"""Diagnostic: per-block phase sums of the persistent conv kernel (shader-clock stamps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pnp_vcve_amd import _native, ops  # noqa: E402

h, w = 720, 1280
dev = torch.device('cuda:0')
x = torch.randn(h, w, 64, device=dev)
x2 = torch.randn(h, w, 64, device=dev)
pw = ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
bias = torch.randn(64, device=dev) * 0.1
for _ in range(3):
    ops.conv3x3([x], [pw], bias=bias, residual=x2)
dbg = torch.zeros(512 * 16, dtype=torch.int64, device=dev)
ops.conv3x3([x], [pw], bias=bias, residual=x2, trace=dbg)      # include/pnpvcve_debug.h
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(512, 16)
n = d[:, 7]
tot = d[:, 3] - d[:, 0]
print('blocks', (n > 0).sum(), 'tiles/block min/max', n.min(), n.max())
for name, v in (('block total', tot), ('K loop / tile', d[:, 1] / n), ('epilogue / tile', d[:, 2] / n),
                ('hand-over / tile', d[:, 6] / np.maximum(n - 1, 1)), ('total / tile', tot / n)):
    print(f'{name:20s} mean {v.mean():10.0f}  p10 {np.percentile(v, 10):10.0f}  p90 {np.percentile(v, 90):10.0f}  max {v.max():10.0f}')
for i, nm in enumerate(['epi: act+transpose writes', 'hand: barrier 1', 'hand: halo regs -> LDS', 'hand: operand prefetch issue', 'hand: barrier 2']):
    v = d[:, 8 + i] / np.maximum(n - (0 if i == 0 else 1), 1)
    print(f'  {nm:28s} mean {v.mean():10.0f}  p10 {np.percentile(v, 10):10.0f}  p90 {np.percentile(v, 90):10.0f}')
print('kernel span (cycles)', d[:, 3].max() - d[:, 0].min())

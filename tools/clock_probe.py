#!/usr/bin/env python
"""In-kernel clock of the headline kernel under its own load (MI355X_MICROARCH.md, DVFS give-back item 6):
    clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz
stamped once around the whole strip of every block of conv3x3_persist_kernel, after >= 2 s of back-to-back un-stamped
launches of the same kernel on random data; median over the 512 blocks.  The stamps go to the debug buffer of
pnp_conv3x3_f32_ex only (include/pnpvcve_debug.h); no output value depends on them.

    python tools/clock_probe.py [--seconds 2.5] > profiles/rNN_clock.txt
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pnp_vcve_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=2.5)
ap.add_argument('--zeros', action='store_true',
                help='all-zero operands (same instruction stream, same cycles): if the kernel is POWER-limited the chip holds a higher '
                     'clock and the launch takes less wall time (MI355X_MICROARCH.md, DVFS give-back item 1)')
args = ap.parse_args()
h, w = 720, 1280
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(h, w, 64, device=dev)
res = torch.randn(h, w, 64, device=dev)
pw = ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
w1 = ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.05 for _ in range(3)])
par = (torch.rand(3, h, w, device=dev) / 255.0).contiguous()
bias = torch.randn(64, device=dev) * 0.1
gam = torch.rand(64, device=dev)
if args.zeros:
    for t_ in (x, res, pw, w1, bias):
        t_.zero_()
PEAK = 157.3


def run(kind, trace=None):
    if kind == 'back':       # block back half / conv_hr: K = 576, residual
        return ops.conv3x3([x], [pw], bias=bias, residual=res, trace=trace)
    return ops.conv3x3([x], [pw], bias=bias, gamma=gam, packed_w1x1=w1, par=par, act=1, trace=trace)     # front half, K = 768, dense par


print(f'device: {torch.cuda.get_device_name(0)}; kernel: conv3x3_persist_kernel (720x1280, 64->64, fp32 MFMA 32x32x2), '
      f'{"ALL-ZERO" if args.zeros else "random"} operands')
for kind, K in (('back', 576), ('front', 768)):
    for _ in range(3):
        run(kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:      # sustained load first: the chip settles on the clock it can hold
        for _ in range(50):
            run(kind)
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run(kind)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    dbg = torch.zeros(512 * 16, dtype=torch.int64, device=dev)
    for _ in range(20):
        run(kind)
    run(kind, trace=dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(512, 16)
    live = d[:, 7] > 0
    cyc = (d[live, 3] - d[live, 0]).astype(np.float64)
    ref = (d[live, 14] - d[live, 13]).astype(np.float64)
    ghz = cyc / ref * 0.1
    fl = 2.0 * K * 64 * h * w
    tf = fl / (us * 1e-6) / 1e12
    print(f'{kind:5s} half (K={K}): {n} warm launches over {args.seconds:.1f} s; un-stamped launch {us:7.1f} us = {tf:6.1f} TFLOP/s '
          f'= {tf / PEAK:.3f} of the 157.3 TFLOP/s peak (2.4 GHz x 256 CU x 256 FLOP/clk)')
    print(f'      in-kernel clock over the strip, {int(live.sum())} blocks: median {np.median(ghz):.3f} GHz  '
          f'(p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); 100 MHz ticks per strip median {np.median(ref):.0f}')
    ceil = PEAK * np.median(ghz) / 2.4
    print(f'      matrix-pipe ceiling at that clock: {ceil:6.1f} TFLOP/s -> the kernel runs at {tf / ceil:.3f} of it')

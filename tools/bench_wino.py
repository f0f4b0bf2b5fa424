"""Kernel-level A/B of the Winograd conv (csrc/conv_wino.hip) against the direct persistent kernel at a frame size (default 720p):
back half (plain conv + residual), conv_hr (leaky-relu), front half (partition branches on a synthetic one-hot/255 map, with and
without tile flags).  Same session, alternating; HIP-event time over N launches each.

    python tools/bench_wino.py [--h 720 --w 1280 --iters 50]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnp_vcve_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--h', type=int, default=720)
    ap.add_argument('--w', type=int, default=1280)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--units', action='store_true', help='also time the quadrant-unit kernel (small frames)')
    args = ap.parse_args()
    h, w = args.h, args.w
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(h, w, 64, device=dev, generator=g)
    res = torch.randn(h, w, 64, device=dev, generator=g)
    wt = torch.randn(64, 64, 3, 3, device=dev, generator=g) * 0.05
    b = torch.randn(64, device=dev, generator=g) * 0.1
    gamma = torch.rand(64, device=dev, generator=g)
    w1 = [torch.randn(64, 64, 1, 1, device=dev, generator=g) * 0.1 for _ in range(3)]
    rng = np.random.RandomState(3)
    cls = np.repeat(np.repeat(rng.randint(0, 3, (h // 8 + 1, w // 8 + 1)), 8, 0), 8, 1)[:h, :w]
    par = torch.from_numpy(np.stack([(cls == j).astype(np.float32) / np.float32(255.0) for j in range(3)])).to(dev)
    pw, p1 = ops.pack_conv3x3(wt), ops.pack_conv1x1(w1)
    u, ug, up = ops.wino_image(pw), ops.wino_image(pw, gamma), ops.wino_par_image(p1)
    flags = ops.par_tile_flags(par)
    flop_back, flop_front = 2.0 * 576 * 64 * h * w, 2.0 * (576 + 192) * 64 * h * w
    tiles = ((h + 15) // 16) * ((w + 15) // 16)
    cases = [
        ('back half   direct', flop_back, lambda: ops.conv3x3([x], [pw], bias=b, residual=res)),
        ('back half   winograd', flop_back, lambda: ops.conv3x3_wino(x, u, bias=b, residual=res)),
        ('conv_hr     direct', flop_back, lambda: ops.conv3x3([x], [pw], bias=b, act=2)),
        ('conv_hr     winograd', flop_back, lambda: ops.conv3x3_wino(x, u, bias=b, act=2)),
        ('front half  direct   (flags)', flop_front, lambda: ops.conv3x3([x], [pw], bias=b, gamma=gamma, packed_w1x1=p1, par=par, act=1,
                                                                          par_flags=flags)),
        ('front half  winograd (flags)', flop_front, lambda: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par,
                                                                               par_flags=flags, act=1)),
        ('front half  winograd (dense)', flop_front, lambda: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, act=1)),
    ]
    if args.units:      # the small-frame form: one block per 8x8 quadrant unit
        cases += [
            ('back half   winograd units', flop_back, lambda: ops.conv3x3_wino(x, u, bias=b, residual=res, units=True)),
            ('conv_hr     winograd units', flop_back, lambda: ops.conv3x3_wino(x, u, bias=b, act=2, units=True)),
            ('front half  winograd units (flags)', flop_front, lambda: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par,
                                                                                        par_flags=flags, act=1, units=True)),
        ]
    print(f'{h}x{w}: {tiles} 16x16 tiles; algorithmic GFLOP back / front = {flop_back / 1e9:.2f} / {flop_front / 1e9:.2f}')
    for r in range(args.rounds):
        for name, flop, fn in cases:
            us = timed(fn, args.iters)
            print(f'round {r}  {name:32s} {us:8.1f} us   {flop / us / 1e6:7.1f} TFLOP/s (algorithmic)', flush=True)


if __name__ == '__main__':
    main()

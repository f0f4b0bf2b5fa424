#!/usr/bin/env python
"""Per-variant timing of the fused conv kernel and the MV warp at a given frame size
(HIP events on the current stream, interleaved rounds, median)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pnp_vcve_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--h', type=int, default=720)
    ap.add_argument('--w', type=int, default=1280)
    ap.add_argument('--rounds', type=int, default=10)
    a = ap.parse_args()
    h, w = a.h, a.w
    dev = torch.device('cuda:0')
    x = torch.randn(h, w, 64, device=dev)
    x2 = torch.randn(h, w, 64, device=dev)
    x3 = torch.randn(h, w, 64, device=dev)
    lr4 = torch.rand(h, w, 4, device=dev)
    wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    win = torch.randn(64, 195, 3, 3, device=dev) * 0.05
    pw = ops.pack_conv3x3(wt)
    p1 = ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev) * 0.1 for _ in range(3)])
    pin = [ops.pack_conv3x3(win, 0, 3)] + [ops.pack_conv3x3(win, 3 + 64 * j, 64) for j in range(3)]
    bias = torch.randn(64, device=dev) * 0.1
    gamma = torch.rand(64, device=dev) * 2
    par = (torch.rand(3, h, w, device=dev) > 0.66).float() / 255.0
    fx = (torch.randint(-32, 33, (h // 8, w // 8), device=dev).float() / 4).repeat_interleave(8, 0).repeat_interleave(8, 1).contiguous()
    fy = (torch.randint(-32, 33, (h // 8, w // 8), device=dev).float() / 4).repeat_interleave(8, 0).repeat_interleave(8, 1).contiguous()

    off = torch.randn(288, h, w, device=dev) * 1.5
    ml = torch.randn(144, h, w, device=dev)
    variants = {
        'conv K=576 plain': (lambda: ops.conv3x3([x], [pw], bias=bias, act=2), 2 * 576 * 64),
        'conv K=576 +residual (block back)': (lambda: ops.conv3x3([x], [pw], bias=bias, residual=x2), 2 * 576 * 64),
        'conv K=768 +gamma+par+relu (block front)': (lambda: ops.conv3x3([x], [pw], bias=bias, gamma=gamma, packed_w1x1=p1, par=par, act=1), 2 * 768 * 64),
        'input conv K=1755 (lr+3x64)': (lambda: ops.conv3x3([lr4, x, x2, x3], pin, bias=bias, act=2), 2 * 195 * 9 * 64),
        'mv_warp': (lambda: ops.mv_warp_nhwc(x, fx, fy), None),
        'dcn (deform_groups 16, flow-guided)': (lambda: ops.modulated_deform_conv_nhwc(x, off, ml, wt, bias, flow=torch.stack([fx, fy])), 'dcn'),
    }
    times = {k: [] for k in variants}
    for k, (fn, _) in variants.items():
        fn()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for k, (fn, _) in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e3)
    for k, (fn, fl) in variants.items():
        ts = sorted(times[k])
        med = ts[len(ts) // 2]
        if fl == 'dcn':
            print(f'{k:44s} median {med:9.1f} us  min {ts[0]:9.1f} us  (incl. host-side om permutation in this wrapper)')
        elif fl:
            print(f'{k:44s} median {med:9.1f} us  min {ts[0]:9.1f} us  {fl * h * w / med / 1e6:7.1f} TFLOP/s')
        else:
            print(f'{k:44s} median {med:9.1f} us  min {ts[0]:9.1f} us  {520.0 * h * w / med / 1e3:7.1f} GB/s')


if __name__ == '__main__':
    main()

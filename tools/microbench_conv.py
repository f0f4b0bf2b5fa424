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
    ap.add_argument('--reps', type=int, default=8, help='back-to-back launches per timed interval')
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
    hw_ = ops.f16_image(pw)
    h1 = ops.f16_image(p1)
    hin = [ops.f16_image(p) for p in pin]
    import ctypes
    from pnp_vcve_amd import _native
    L = _native.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    keep = []

    def prepared(fn_name, srcs, wimgs, bias_=None, gamma_=None, w1=None, par_=None, res=None, act=0):
        """pre-marshalled direct call (a 100-us kernel is shorter than ops.conv3x3's python overhead)"""
        n = len(srcs)
        out = torch.empty((h, w, 64), device=dev)
        sp = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in srcs])
        sc = (ctypes.c_int * n)(*[s_.shape[2] for s_ in srcs])
        wp = (ctypes.c_void_p * n)(*[p_.data_ptr() for p_ in wimgs])
        keep.extend([out, sp, sc, wp])
        P = lambda t_: ctypes.c_void_p(t_.data_ptr()) if t_ is not None else None
        args = (n, sp, sc, wp, P(bias_), P(gamma_), P(w1), P(par_), P(res), act, P(out), h, w, st)
        f = getattr(L, fn_name)
        return lambda: f(*args)

    variants = {
        'conv K=576 plain': (prepared('pnp_conv3x3_f32', [x], [pw], bias, act=2), 2 * 576 * 64),
        'conv K=576 +residual (block back)': (lambda: ops.conv3x3([x], [pw], bias=bias, residual=x2), 2 * 576 * 64),
        'conv K=768 +gamma+par+relu (block front)': (lambda: ops.conv3x3([x], [pw], bias=bias, gamma=gamma, packed_w1x1=p1, par=par, act=1), 2 * 768 * 64),
        'input conv K=1755 (lr+3x64)': (lambda: ops.conv3x3([lr4, x, x2, x3], pin, bias=bias, act=2), 2 * 195 * 9 * 64),
        'fp16 conv K=576 plain': (prepared('pnp_conv3x3_f16', [x], [hw_], bias, act=2), 2 * 576 * 64),
        'fp16 conv K=576 +residual (block back)': (prepared('pnp_conv3x3_f16', [x], [hw_], bias, res=x2), 2 * 576 * 64),
        'fp16 conv K=768 +gamma+par+relu (front)': (prepared('pnp_conv3x3_f16', [x], [hw_], bias, gamma, h1, par, act=1), 2 * 768 * 64),
        'fp16 input conv K=1755 (3 launches)': (prepared('pnp_conv3x3_f16', [lr4, x, x2, x3], hin, bias, act=2), 2 * 195 * 9 * 64),
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
            for _r in range(a.reps):
                fn()
            e1.record()
            e1.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    for k, (fn, fl) in variants.items():
        ts = sorted(times[k])
        med = ts[len(ts) // 2]
        if fl == 'dcn':
            print(f'{k:44s} median {med:9.1f} us  min {ts[0]:9.1f} us  (incl. host-side om permutation in this wrapper)')
        elif fl:
            hbm = (3 if 'residual' in k else 2) * 256.0 * h * w / med / 1e3
            print(f'{k:44s} median {med:9.1f} us  min {ts[0]:9.1f} us  {fl * h * w / med / 1e6:7.1f} TFLOP/s  '
                  f'{hbm:7.0f} GB/s (in+out{"+res" if "residual" in k else ""})')
        else:
            print(f'{k:44s} median {med:9.1f} us  min {ts[0]:9.1f} us  {520.0 * h * w / med / 1e3:7.1f} GB/s')


if __name__ == '__main__':
    main()

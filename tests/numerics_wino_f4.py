#!/usr/bin/env python
"""fp32 error of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct form -- one 64 -> 64 conv and the whole recurrent clip (33
3x3 convs per frame, two sweeps) -- measured on the CPU with the oracle as the frame (the numbers that decide
whether a PNP_OPT_WINOGRAD = 3 kernel is worth building).  Not collected by pytest (no test_ prefix); lives under tests/ because it
imports oracle/ (test infrastructure).

The Winograd convs are emulated in fp32: transforms as fp32 adds / multiplies in the order a kernel would do them (rows, then columns),
products summed over the input channels by an fp32 matmul per transform position (a different order than an MFMA chain, the same
magnitude of rounding).  Every 3x3 conv whose weight is (64, C, 3, 3) with C a multiple of 64 (block halves incl. the expert-mixed
ones, conv_hr) and the wide sources of the input convs go through the emulation; the RGB slice of an input conv, conv_last and the
1x1 branches stay direct -- as in csrc/conv_wino.hip.

    python tests/numerics_wino_f4.py [--h 128 --w 128 --t 7]      # ~1 min at 128x128
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_ref  # noqa: E402
from pnp_vcve_amd import synthetic as syn  # noqa: E402

# Lavin & Gray 2015: F(2x2,3x3) and F(4x4,3x3) (interpolation points 0, +-1, +-2, inf)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
              np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)

ORIG_CONV2D = F.conv2d


def wino_conv(x, w, bias, m):
    """conv3x3(pad 1) of x (1,C,H,W) with w (O,C,3,3) as F(m x m, 3x3) in fp32"""
    BT, G, AT = (BT2, G2, AT2) if m == 2 else (BT4, G4, AT4)
    a = m + 2
    n, c, h, wd = x.shape
    o = w.shape[0]
    th, tw = -(-h // m), -(-wd // m)
    xp = F.pad(x, (1, tw * m - wd + 1, 1, th * m - h + 1))
    # patches (c, th, tw, a, a)
    p = xp[0].unfold(1, a, m).unfold(2, a, m)
    bt = torch.from_numpy(BT).float()
    # V = B^T d B in fp32, rows then columns (weights image U = G g G^T computed in fp64 and rounded once, like wino_image_kernel)
    v = torch.einsum('ij,cyxjk->cyxik', bt, p)
    v = torch.einsum('cyxik,lk->cyxil', v, bt)
    u = torch.einsum('ij,ocjk,lk->ocil', torch.from_numpy(G), w.double(), torch.from_numpy(G)).float()
    # per position: (o, c) @ (c, tiles) in fp32
    mm = torch.einsum('ocil,cyxil->oyxil', u, v)
    at = torch.from_numpy(AT).float()
    y = torch.einsum('ij,oyxjk->oyxik', at, mm)
    y = torch.einsum('oyxik,lk->oyxil', y, at)                  # (o, th, tw, m, m)
    y = y.permute(0, 1, 3, 2, 4).reshape(o, th * m, tw * m)[:, :h, :wd]
    if bias is not None:
        y = y + bias.view(-1, 1, 1)
    return y.unsqueeze(0)


MODE = 0        # 0 direct, 2 F(2x2), 4 F(4x4)
FEATS = {}      # MODE -> the inputs of conv_last (the last 64-channel map of every frame: what 33 convs per frame and the recurrence made)


def conv2d_patched(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if w.dim() == 4 and w.shape[0] == 3 and w.shape[1] == 64:
        FEATS.setdefault(MODE, []).append(x.clone())
    if MODE and w.dim() == 4 and w.shape[2:] == (3, 3) and padding == 1 and w.shape[0] == 64 and x.shape[0] == 1 and groups == 1:
        cin = w.shape[1]
        if cin % 64 == 0:
            return wino_conv(x, w, bias, MODE)
        if cin % 64 == 3:                                       # input conv over [frame, wide sources]: the frame's slice stays direct
            y = ORIG_CONV2D(x[:, :3], w[:, :3], bias, padding=1)
            return y + wino_conv(x[:, 3:], w[:, 3:], None, MODE)
    return ORIG_CONV2D(x, w, bias, stride, padding, dilation, groups)


def main():
    global MODE
    ap = argparse.ArgumentParser()
    ap.add_argument('--h', type=int, default=128)
    ap.add_argument('--w', type=int, default=128)
    ap.add_argument('--t', type=int, default=7)
    args = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    # ---- one conv on a unit-scale map against fp64
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 64, 96, 96, generator=g) * 2 - 1
    w = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) * 0.06
    ref = ORIG_CONV2D(x.double(), w.double(), padding=1)
    print('one 64->64 conv, unit-scale map, max|. - fp64|:  direct fp32 %.2e   F(2x2) %.2e   F(4x4) %.2e'
          % (float((ORIG_CONV2D(x, w, padding=1).double() - ref).abs().max()), float((wino_conv(x, w, None, 2).double() - ref).abs().max()),
             float((wino_conv(x, w, None, 4).double() - ref).abs().max())))
    # ---- the whole clip through the oracle with every Winograd-eligible conv replaced
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd = cpu_ref.to_torch_state(syn.make_state_dict(cfg, seed=2025))
    clip = syn.make_clip(seed=1000, n=1, t=args.t, h=args.h, w=args.w, slices='IBBBP', qp_mode='qp', crf=25, block=8, par_classes=3)
    a = {k: torch.from_numpy(v) for k, v in clip.items()}
    outs = {}
    cpu_ref.F.conv2d = conv2d_patched
    try:
        for MODE in (0, 2, 4):
            with torch.no_grad():
                outs[MODE] = cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])
    finally:
        cpu_ref.F.conv2d = ORIG_CONV2D
    for m in (2, 4):
        d = (outs[m] - outs[0]).abs()
        fe = max(float((x - y).abs().max()) for x, y in zip(FEATS[m], FEATS[0]))
        fs = max(float(y.abs().max()) for y in FEATS[0])
        print(f'{args.t}x3x{args.h}x{args.w} clip, F({m}x{m},3x3) in every eligible conv vs the oracle: output max {float(d.max()):.2e}  mean {float(d.mean()):.2e}'
              f'   (gates: north_star 1e-3, tests 2e-5);  last 64-channel map of a frame: max |diff| {fe:.2e} on values up to {fs:.2f}')


if __name__ == '__main__':
    main()

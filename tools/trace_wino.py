"""Per-tile cycle split of the Winograd conv (csrc/conv_wino.hip) from its timeline buffer: K loop vs epilogue, in-kernel clock,
block lifetimes.    python tools/trace_wino.py [--h 720 --w 1280] [--kind back|hr|front|front_dense]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnp_vcve_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--h', type=int, default=720)
    ap.add_argument('--w', type=int, default=1280)
    ap.add_argument('--kind', default='all')
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
    from pnp_vcve_amd import _native
    words = {v: torch.tensor([v], dtype=torch.int32, device=dev) for v in (0, 8)}

    def gated(word, fn):
        _native.lib().pnp_debug_wino_gate_word(ctypes.c_void_p(words[word].data_ptr()))
        try:
            return fn()
        finally:
            _native.lib().pnp_debug_wino_gate_word(None)
    kinds = {
        'hr': lambda tr: ops.conv3x3_wino(x, u, bias=b, act=2, trace=tr),
        'back': lambda tr: ops.conv3x3_wino(x, u, bias=b, residual=res, trace=tr),
        'front': lambda tr: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, par_flags=flags, act=1, trace=tr),
        # the generator's one gated launch, fold-only body (every 8x8 block of `par` is one constant plane)
        'front_fold': lambda tr: gated(8, lambda: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, par_flags=flags, act=1, trace=tr)),
        'front_gated_branch': lambda tr: gated(0, lambda: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, par_flags=flags, act=1, trace=tr)),
        'front_dense': lambda tr: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, act=1, trace=tr),
        # no partition record anywhere (an I frame): the branch chunks hold no MFMA -- what the PAR structure costs by itself
        'front_zero': lambda tr: ops.conv3x3_wino(x, ug, bias=b, gamma=gamma, wino_w1x1=up, par=par * 0, par_flags=flags, act=1, trace=tr),
    }
    if args.kind == 'all':
        tr0 = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
        print('front_fold == front (branch kernel), bit for bit:', bool(torch.equal(kinds['front_fold'](tr0), kinds['front'](tr0))), flush=True)
    for name, fn in kinds.items():
        if args.kind not in ('all', name):
            continue
        tr = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
        for _ in range(20):        # the clock settles under load
            fn(tr)
        torch.cuda.synchronize()
        d = tr.cpu().numpy().reshape(256, 16)
        d = d[d[:, 7] > 0]
        n = d[:, 7].astype(np.float64)
        life = (d[:, 3] - d[:, 0]).astype(np.float64)
        clk = life / ((d[:, 14] - d[:, 13]).astype(np.float64) * 10e-9) / 1e9
        print(f'{name:12s} blocks {len(d)}  tiles/block {n.min():.0f}..{n.max():.0f}  K loop {np.median(d[:, 1] / n):8.0f} cycles/tile  '
              f'epilogue {np.median(d[:, 2] / n):7.0f}  (branch chunks {np.median(d[:, 8] / n):6.0f})  other {np.median((life - d[:, 1] - d[:, 2]) / n):6.0f}  '
              f'lifetime median {np.median(life):9.0f} max {life.max():9.0f} cycles  clock {np.median(clk):.2f} GHz  '
              f'launch {(d[:, 14].max() - d[:, 13].min()) * 10e-3:.1f} us', flush=True)
        q = d[d[:, 4] > 0]
        if len(q):      # blocks with a quadrant unit behind their last whole tile: wait for loads + barrier | four steps | output
            print(f'{"":12s} quadrant units {len(q)}: start -> first step {np.median(q[:, 5] - q[:, 4]):6.0f} cycles, steps {np.median(q[:, 6] - q[:, 5]):6.0f}, '
                  f'output {np.median(q[:, 3] - q[:, 6]):6.0f}; lifetime of these blocks {np.median(q[:, 3] - q[:, 0]):9.0f} vs the others {np.median((d[d[:, 4] == 0])[:, 3] - (d[d[:, 4] == 0])[:, 0]):9.0f}', flush=True)


if __name__ == '__main__':
    main()

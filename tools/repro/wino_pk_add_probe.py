"""Where does the PAR + RES Winograd tile kernel go wrong when hipcc may emit v_pk_add_f32 (conv_wino.hip built WITHOUT
-target-feature -packed-fp32-ops; round 6: packed ADDS alone reproduce round 5's signature -- no packed multiply or FMA in the code
object)?  Run once per build; prints the wrong output positions as (tile, wave quadrant, Winograd tile, pixel in tile, channel)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import golden_util as gu
from pnp_vcve_amd import ops
dev = torch.device('cuda:0')
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
nhwc = lambda x: ops.nchw_to_nhwc(G(x))[0]
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (56, 72)
x = gu.syn.uniform(10, 'x', (1, 64, h, w), -1, 1)
res = gu.syn.uniform(10, 'r', (1, 64, h, w), -2, 2)
wt = gu.syn.uniform(10, 'w', (64, 64, 3, 3), -0.06, 0.06)
b = gu.syn.uniform(10, 'b', (64,), -0.1, 0.1)
w1 = [gu.syn.uniform(10, f'w1{j}', (64, 64, 1, 1), -3.0, 3.0) for j in range(3)]
rng = np.random.RandomState(12)
cls = np.repeat(np.repeat(rng.randint(0, 3, ((h + 7) // 8, (w + 7) // 8)), 8, 0), 8, 1)[:h, :w]
par = np.stack([(cls == j).astype(np.float32) / np.float32(255.0) for j in range(3)])
u = ops.wino_image(ops.pack_conv3x3(G(wt)))
up = ops.wino_par_image(ops.pack_conv1x1([G(v) for v in w1]))
xd = torch.from_numpy(x).double()
ref = F.conv2d(xd, torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1)
for j in range(3):
    ref = ref + torch.from_numpy(par[j]).double()[None, None] * F.conv2d(xd, torch.from_numpy(w1[j]).double())
ref = ref + torch.from_numpy(res).double()
for name, kw in (('branches + residual (PAR+RES)', dict(wino_w1x1=up, par=G(par), residual=nhwc(res))),
                 ('residual only (RES)', dict(residual=nhwc(res))), ('branches only (PAR)', dict(wino_w1x1=up, par=G(par)))):
    out = ops.conv3x3_wino(nhwc(x), u, bias=G(b), **kw)          # (h, w, 64)
    r = ref if 'residual' in kw and 'wino_w1x1' in kw else None
    if r is None:
        r = F.conv2d(xd, torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1)
        if 'wino_w1x1' in kw:
            for j in range(3):
                r = r + torch.from_numpy(par[j]).double()[None, None] * F.conv2d(xd, torch.from_numpy(w1[j]).double())
        if 'residual' in kw:
            r = r + torch.from_numpy(res).double()
    e = (out.permute(2, 0, 1).double().cpu() - r[0]).abs()       # (64, h, w)
    bad = (e > 1e-4).nonzero()
    print(f'{name}: max err {float(e.max()):.3e}, wrong values {len(bad)} of {e.numel()}')
    seen = {}
    for ch, y, xx in bad.tolist():
        tile = (y // 16, xx // 16)
        wave = 2 * ((y % 16) // 8) + ((xx % 16) // 8)
        wt_ = (((y % 8) // 2), ((xx % 8) // 2))
        key = (wave, wt_, (y % 2, xx % 2), ch % 16, ch // 16)
        seen[key] = seen.get(key, 0) + 1
    for k, v in sorted(seen.items())[:24]:
        print('   wave %d  wino tile %s  pixel %s  channel lane m=%d  N tile %d : %d tiles' % (*k, v))
    if bad.numel():
        o = out.permute(2, 0, 1).double().cpu()
        rr = torch.from_numpy(res).double()[0]
        print('   tile (0,0), per pixel the wrong channels:')
        for y in range(16):
            for xx in range(16):
                chs = [c for c in range(64) if e[c, y, xx] > 1e-4]
                if chs:
                    print('     y %2d x %2d: %s' % (y, xx, chs))
        print('   first wrong values: got, want, want - residual, residual, got - (want - residual)')
        for ch, y, xx in bad.tolist()[:12]:
            g, wv, rv = float(o[ch, y, xx]), float(r[0, ch, y, xx]), float(rr[ch, y, xx])
            d = g - (wv - rv)
            # is the residual that was added the residual of some other position of the same pixel block?
            cand = [(c2, y2, x2) for c2 in range(64) for y2 in range(max(0, y - 3), min(h, y + 4)) for x2 in range(max(0, xx - 3), min(w, xx + 4))
                    if abs(float(rr[c2, y2, x2]) - d) < 2e-6]
            print('     ch %2d y %2d x %2d: %9.5f %9.5f %9.5f %9.5f %9.5f  residual-like source %s' % (ch, y, xx, g, wv, wv - rv, rv, d, cand[:3]))
        tiles = sorted({(y // 16, xx // 16) for _, y, xx in bad.tolist()})
        print('   tiles (row, col) with wrong values:', tiles, ' frame tiles:', ((h + 15) // 16, (w + 15) // 16))

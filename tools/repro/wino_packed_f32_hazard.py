"""Error pattern of the Winograd conv on an identity weight: run once per build of conv_wino.hip (with / without packed fp32 ops;
profiles/r05_wino_packed_f32_hazard.txt)."""
import os
import sys

import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pnp_vcve_amd import ops
dev = torch.device('cuda:0')
h, w = 32, 32
x = torch.randn(h, w, 64, device=dev)
wt = torch.zeros(64, 64, 3, 3, device=dev)
wt[torch.arange(64), torch.arange(64), 1, 1] = 1.0
u = ops.wino_image(ops.pack_conv3x3(wt))
y = ops.conv3x3_wino(x, u)
e = (y - x).abs()
print('max err', float(e.max()))
bad = (e > 1e-5)
print('bad fraction', float(bad.float().mean()))
print('bad by channel group of 16:', [float(bad[:, :, 16*i:16*i+16].float().mean()) for i in range(4)])
print('bad by pixel row:', [round(float(bad[r].float().mean()), 2) for r in range(h)])
print('bad by pixel col:', [round(float(bad[:, c].float().mean()), 2) for c in range(w)])
# where does a wrong value come from?
yy = y.cpu().numpy(); xx = x.cpu().numpy()
idx = np.argwhere(bad.cpu().numpy())[:8]
for (r, c, ch) in idx:
    v = yy[r, c, ch]
    m = np.argwhere(np.abs(xx - v) < 1e-6)
    print((r, c, ch), 'got', v, 'want', xx[r, c, ch], 'matches x at', m[:3].tolist())

#!/usr/bin/env python
"""Does interleaving TWO 720p clips on two streams fill the tail of each persistent conv launch (7200 tiles on 512 strips = 14.06
rounds: the 15th runs on 32 blocks, ~3 % of a launch)?  frames/s for n = 1, n = 2 one after another, n = 2 concurrently.

    python tools/tail_fill_probe.py [fp32|f16x3|fp16]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from pnp_vcve_amd import synthetic as syn  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda:0')
cfg = dict(syn.DEFAULT_GENERATOR_CFG)
sd = syn.make_state_dict(cfg, seed=2025)
m = bench.build_model(cfg, sd, dev, prec)


def rate(n, concurrent, steps=4):
    type(m).LARGE_FRAME_CONTEXTS = 2 if concurrent else 1
    m._workspace.clear()
    _, a = bench.make_inputs(1000, 7, 720, 1280, dev, n)
    f = lambda: m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])      # noqa: E731
    with torch.no_grad():
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = f()
        torch.cuda.synchronize()
    return steps * 7 * n / (time.perf_counter() - t0), out


r1, o1 = rate(1, False)
r2s, o2s = rate(2, False)
r2c, o2c = rate(2, True)
print(f'{prec}: 1 clip {r1:.2f} frames/s; 2 clips one after another {r2s:.2f}; 2 clips on two streams {r2c:.2f} ({100 * (r2c / r2s - 1):+.1f} %); '
      f'bit-identical: {torch.equal(o2s, o2c)}')

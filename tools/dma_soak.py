#!/usr/bin/env python
"""Soak of the LDS-DMA pipeline of the Winograd tile kernels (round 6): many forwards of the same clip must be bit-identical to the first
-- a load that lands in LDS after the barrier that publishes it would show up as a run-to-run difference -- at the headline size, at a
ragged size, at 180x320, with and without unrelated memory traffic on a second stream (which stretches the latencies the counted
vmcnt waits rely on being ordered, not short), and with two clips in flight.

    python tools/dma_soak.py [--n 40]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pnp_vcve_amd import synthetic as syn  # noqa: E402
from pnp_vcve_amd.registry import build_backbone  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=40)
args = ap.parse_args()
dev = torch.device('cuda:0')


def model(**kw):
    cfg = dict(syn.DEFAULT_GENERATOR_CFG, **kw)
    sd = syn.make_state_dict(cfg, seed=2025)
    m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m.to(dev).eval()


def fwd(m, a):
    with torch.no_grad():
        return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])


def clip(h, w, n=1, t=7, block=8, seed=1000):
    c = syn.make_clip(seed=seed, n=n, t=t, h=h, w=w, slices='IBBBP', qp_mode='qp', crf=25, block=block, par_classes=3)
    return {k: torch.from_numpy(v).to(dev) for k, v in c.items()}


noise_src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
noise_dst = torch.empty_like(noise_src)
side = torch.cuda.Stream()
bad = 0
for name, h, w, n, block, kw, reps in (('720x1280 IBBBP', 720, 1280, 1, 8, {}, args.n), ('720x1280, two clips in flight', 720, 1280, 2, 8, {}, args.n // 2),
                                       ('716x1268 (ragged tiles and quadrants)', 716, 1268, 1, 8, {}, args.n // 2),
                                       ('720x1280, 4x4 blocks (branch bodies)', 720, 1280, 1, 4, {}, args.n // 2),
                                       ('720x1280 channel-last blocks', 720, 1280, 1, 8, {'channel_first': False}, args.n // 2),
                                       ('180x320', 180, 320, 1, 8, {}, 4 * args.n)):
    m = model(**kw)
    a = clip(h, w, n=n, block=block, t=7 if h > 200 else 7)
    ref = fwd(m, a)
    torch.cuda.synchronize()
    diffs = 0
    for i in range(reps):
        if i % 2:                                  # every other forward beside 256-MB copies on another stream
            with torch.cuda.stream(side):
                for _ in range(8):
                    noise_dst.copy_(noise_src, non_blocking=True)
        out = fwd(m, a)
        if not torch.equal(out, ref):
            diffs += 1
            print('  DIFFERENCE at forward', i, float((out - ref).abs().max()), flush=True)
        torch.cuda.synchronize()
    bad += diffs
    print(f'{name}: {reps} forwards, {diffs} differed from the first; finite: {bool(torch.isfinite(ref).all())}', flush=True)
    del m, a, ref
    torch.cuda.empty_cache()
print('SOAK', 'FAILED' if bad else 'OK')
sys.exit(1 if bad else 0)

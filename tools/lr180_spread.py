#!/usr/bin/env python
"""Why did the 7x3x180x320 fp32 Winograd entry read 480 ... 702 frames/s across runs of round 5 while the direct path beside it read
560 every time?  One fresh process: per-forward wall times (synchronised) of `--n` consecutive forwards for
PNP_OPT_WINOGRAD = 1 and 0, under perturbations: right after model construction, after an idle gap, right behind a 720p load
(what bench.py's secondary list does: the lr180 entries run behind 720p entries in the same process), and in the bench's own form
(2 warm-ups + 5 steps, one synchronise around the 5).

    python tools/lr180_spread.py [--n 30] [--tag A]      # run it in several fresh processes; the lines carry the tag
"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pnp_vcve_amd import _native, synthetic as syn  # noqa: E402
from pnp_vcve_amd.registry import build_backbone  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=30)
ap.add_argument('--tag', default='')
ap.add_argument('--h', type=int, default=180)
ap.add_argument('--w', type=int, default=320)
ap.add_argument('--skip-720p', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
T = 7


def model(wino):
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd = syn.make_state_dict(cfg, seed=2025)
    m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.to(dev).eval()
    m.set_option(_native.OPT_WINOGRAD, wino)
    return m


def inputs(h, w):
    clip = syn.make_clip(seed=1000, n=1, t=T, h=h, w=w, slices='IBBBP', qp_mode='qp', crf=25, block=8 if h % 8 == 0 else 4, par_classes=3)
    return {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}


def fwd(m, a):
    with torch.no_grad():
        return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])


def series(m, a, n):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fwd(m, a)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return out


def bench_form(m, a, warm=2, steps=5):
    for _ in range(warm):
        fwd(m, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fwd(m, a)
    torch.cuda.synchronize()
    return T * steps / (time.perf_counter() - t0)


def show(label, ms):
    fps = [T / (x * 1e-3) for x in ms]
    print(f'{args.tag} {label:58s} first3 ms {ms[0]:7.2f} {ms[1]:7.2f} {ms[2]:7.2f} | rest: min {min(ms[3:]):6.2f} med {statistics.median(ms[3:]):6.2f} '
          f'max {max(ms[3:]):6.2f} ms  -> {min(fps[3:]):6.0f} / {statistics.median(fps[3:]):6.0f} / {max(fps[3:]):6.0f} frames/s', flush=True)


a = inputs(args.h, args.w)
mw, md = model(1), model(0)
for name, m in (('winograd=1', mw), ('winograd=0', md)):
    show(f'{name} cold (first forwards of the process)', series(m, a, args.n))
for name, m in (('winograd=1', mw), ('winograd=0', md)):
    show(f'{name} warm, back to back', series(m, a, args.n))
for name, m in (('winograd=1', mw), ('winograd=0', md)):
    time.sleep(2.0)
    show(f'{name} after 2 s idle', series(m, a, args.n))
print(args.tag, 'bench form (2 warm-ups + 5 steps, one sync): winograd=1 %.0f %.0f %.0f   winograd=0 %.0f %.0f %.0f frames/s'
      % (*[bench_form(mw, a) for _ in range(3)], *[bench_form(md, a) for _ in range(3)]), flush=True)
if not args.skip_720p:
    big = inputs(720, 1280)
    for name, m in (('winograd=1', mw), ('winograd=0', md)):
        for _ in range(3):
            fwd(mw, big)                         # ~0.3 s of the headline's load right in front
        torch.cuda.synchronize()
        show(f'{name} right behind 3 x 720p forwards', series(m, a, args.n))
        for _ in range(3):
            fwd(mw, big)
        torch.cuda.synchronize()
        print(args.tag, f'{name} bench form right behind 720p: %.0f frames/s' % bench_form(m, a), flush=True)
    # the same model object switched between frame sizes (workspace re-use: does the first small forward behind a big one pay?)
    del big
    torch.cuda.empty_cache()
    for name, m in (('winograd=1', mw), ('winograd=0', md)):
        show(f'{name} after empty_cache()', series(m, a, args.n))

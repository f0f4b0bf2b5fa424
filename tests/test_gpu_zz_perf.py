"""Perf guard, collected LAST (tests/conftest.py orders parity files first): catches a build that is right but catastrophically slow --
the accumulators of the Winograd K loop in scratch memory gave correct results ten times slower; nothing here
compares one rate with another, and the margin is wide: the headline shape runs at ~80 frames/s, the direct kernels of rounds 1-4 at 46."""
import statistics
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_headline_shape_is_not_catastrophically_slow():
    from pnp_vcve_amd import synthetic as syn
    from pnp_vcve_amd.registry import build_backbone
    dev = torch.device('cuda:0')
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd = syn.make_state_dict(cfg, seed=2025)
    m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.to(dev).eval()
    clip = syn.make_clip(seed=1000, n=1, t=7, h=720, w=1280, slices='IBBBP', qp_mode='qp', crf=25, block=8, par_classes=3)
    a = {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}

    def fwd():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])

    for _ in range(2):
        fwd()
    rates = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        rates.append(3 * 7 / (time.perf_counter() - t0))
    med = statistics.median(rates)
    print('7x3x720x1280 fp32, median of 3 x 3 forwards:', med, 'frames/s', rates)
    assert med > 30.0, rates

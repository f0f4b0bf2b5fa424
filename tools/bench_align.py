#!/usr/bin/env python
"""Per-kind device time of one clip for each aligner (deform = vos | basic | fvc) at a given frame size."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pnp_vcve_amd import synthetic as syn  # noqa: E402
from pnp_vcve_amd.registry import build_backbone  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--h', type=int, default=720)
    ap.add_argument('--w', type=int, default=1280)
    ap.add_argument('--t', type=int, default=4)
    ap.add_argument('--fp16', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for deform in ('vos', 'basic', 'fvc'):
        cfg = dict(syn.DEFAULT_GENERATOR_CFG, deform=deform)
        sd = syn.make_state_dict(cfg, seed=3)
        m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        m = m.to(dev).eval()
        m.fp16_enabled = a.fp16
        clip = syn.make_clip(seed=4, n=1, t=a.t, h=a.h, w=a.w, slices='IBBBP')
        x = {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}
        with torch.no_grad():
            m(x['lq'], x['QPs'], x['slices'], x['mvs'], x['base_QPs'], x['partitions'])
            m.profile(True)
            m(x['lq'], x['QPs'], x['slices'], x['mvs'], x['base_QPs'], x['partitions'])
            torch.cuda.synchronize()
        p = m.profile_read()
        m.profile(False)
        line = f'deform={deform:6s}'
        for k, v in p.items():
            if v['launches']:
                unit = v['work'] / (v['ms'] * 1e-3) / (1e9 if k in ('mv_warp', 'dcn') else 1e12)
                line += f'  {k}: {v["launches"]}x {1e3 * v["ms"] / v["launches"]:.1f} us ({unit:.0f} {"GB/s" if k in ("mv_warp", "dcn") else "TF"})'
        print(line)


if __name__ == '__main__':
    main()

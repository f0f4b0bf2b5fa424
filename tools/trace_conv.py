#!/usr/bin/env python
"""Diagnostic: per-block timeline of one conv launch (shader-clock stamps written by the kernel)."""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pnp_vcve_amd import _native, ops  # noqa: E402


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 720
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    dev = torch.device('cuda:0')
    x = torch.randn(h, w, 64, device=dev)
    x2 = torch.randn(h, w, 64, device=dev)
    wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    pw = ops.pack_conv3x3(wt)
    bias = torch.randn(64, device=dev) * 0.1
    ntiles = ((w + 15) // 16) * ((h + 7) // 8)
    for _ in range(3):
        ops.conv3x3([x], [pw], bias=bias, residual=x2, variant=_native.CONV_TILE_BIG)
    dbg = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)
    # include/pnpvcve_debug.h: the tile-per-block kernel with 8x16 tiles and a timeline buffer
    ops.conv3x3([x], [pw], bias=bias, residual=x2, variant=_native.CONV_TILE_BIG, trace=dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(ntiles, 16).astype(np.int64)
    t0, t1, t2, t3 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    hw, xcc, rt = d[:, 4], d[:, 5], d[:, 6]
    base = t0.min()
    print('blocks', ntiles, 'span (cycles)', (t3.max() - base), 'realtime span us', (rt.max() - rt.min()) / 100.0)

    def st(name, v):
        print(f'{name:28s} mean {v.mean():10.0f}  p10 {np.percentile(v, 10):10.0f}  p50 {np.percentile(v, 50):10.0f}  p90 {np.percentile(v, 90):10.0f}  max {v.max():10.0f}')
    st('stage A (t1-t0)', t1 - t0)
    st('K loop (t2-t1)', t2 - t1)
    st('epilogue (t3-t2)', t3 - t2)
    st(' issue stores (e1-t2)', d[:, 11] - t2)
    st(' drain (t3-e1)', t3 - d[:, 11])
    st('block total', t3 - t0)
    clk = (t3 - t0) / np.maximum(d[:, 6] - d[:, 12], 1) * 100.0
    print(f'shader clock from s_memtime/s_memrealtime: median {np.median(clk):.0f} MHz  p10 {np.percentile(clk,10):.0f}  p90 {np.percentile(clk,90):.0f}')
    # co-residency: group by (xcc, se/sh/cu bits of HW_ID)
    cu = ((xcc & 0xF) << 16) | (hw & 0xFF00)      # drop wave/simd/pipe bits
    groups = defaultdict(list)
    for i in range(ntiles):
        groups[int(cu[i])].append(i)
    print('distinct CUs', len(groups), 'blocks/CU min/max', min(len(v) for v in groups.values()), max(len(v) for v in groups.values()))
    # concurrency profile on one CU
    k = sorted(groups)[0]
    idx = sorted(groups[k], key=lambda i: t0[i])
    print('one CU timeline (start, stageA, kloop, epi) in kcycles:')
    for i in idx[:12]:
        print(f'  start {(t0[i]-base)/1e3:8.1f}  stage {(t1[i]-t0[i])/1e3:6.1f}  k {(t2[i]-t1[i])/1e3:6.1f}  epi {(t3[i]-t2[i])/1e3:6.1f}  end {(t3[i]-base)/1e3:8.1f}  tile {d[i,7]}')
    # fraction of CU time with 0/1/2 blocks inside their K loop
    tot = defaultdict(float)
    for k, ids in groups.items():
        ev = []
        for i in ids:
            ev.append((t1[i], 1))
            ev.append((t2[i], -1))
        ev.sort()
        cur, last = 0, ev[0][0]
        s0 = min(t0[i] for i in ids)
        e0 = max(t3[i] for i in ids)
        tot[0] += ev[0][0] - s0
        for tt, dlt in ev:
            tot[cur] += tt - last
            cur += dlt
            last = tt
        tot[0] += e0 - last
    allt = sum(tot.values())
    print('CU time share with N blocks in K loop:', {k: round(v / allt, 3) for k, v in sorted(tot.items())})


if __name__ == '__main__':
    main()

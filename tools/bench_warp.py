#!/usr/bin/env python
"""Device time of the MV alignment kernel alone (pnp_mv_warp_nhwc_f32 on a 64-channel map, quarter-pel block vectors as SURVEY 8(d)):
    python tools/bench_warp.py [H W]        ->  us per launch, GB/s of its 520 B per pixel"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pnp_vcve_amd import ops  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (720, 1280)
dev = torch.device('cuda:0')
torch.manual_seed(0)
feat = torch.randn(h, w, 64, device=dev)
blk = torch.randint(-64, 65, (2, (h + 7) // 8, (w + 7) // 8), device=dev).float() / 4.0
fx, fy = (blk[i].repeat_interleave(8, 0).repeat_interleave(8, 1)[:h, :w].contiguous() for i in range(2))
for mode in ('bilinear', 'nearest'):
    for _ in range(5):
        ops.mv_warp_nhwc(feat, fx, fy, interpolation=mode)
    reps = 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.mv_warp_nhwc(feat, fx, fy, interpolation=mode)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    alg = h * w * ((520 if mode == 'bilinear' else 520) )
    print(f'{h}x{w} {mode:8s} {us:7.1f} us per launch  {alg / us / 1e3:7.0f} GB/s of 520 B per pixel')

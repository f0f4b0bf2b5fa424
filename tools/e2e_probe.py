#!/usr/bin/env python
"""The end-to-end loop of bench.py (pipeline_e2e) by itself, a few repetitions: where does the forward inside the loop lose against the
same two-clip forward in the bench?   [GPU_MAX_HW_QUEUES=8] python tools/e2e_probe.py [--reps 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--workers', type=int, default=16)
ap.add_argument('--prewarm', action='store_true', help='fill the caching allocators (device + pinned host) before the first pass')
args = ap.parse_args()
dev = torch.device('cuda:0')
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'), ' prewarm =', args.prewarm, flush=True)
if args.prewarm:
    x = [torch.empty(512 << 20, dtype=torch.uint8, device=dev) for _ in range(8)]
    y = [torch.empty(64 << 20, dtype=torch.uint8).pin_memory() for _ in range(12)]
    torch.cuda.synchronize()
    del x, y
for r in range(args.reps):
    e = bench.pipeline_e2e(dev, 7, workers=args.workers)
    print('rep %d: %.1f frames/s  total %.3f s  forward %.3f s (%.1f frames/s inside the forward)  loader waits %.3f  metrics+submit %.3f  drain %.3f'
          % (r, e['value'], e['seconds_total'], e['seconds_generator_forward'], e['frames'] / e['seconds_generator_forward'],
             e['seconds_main_thread_waiting_for_loader_h2d'], e['seconds_metrics_and_uint8_d2h_and_submit'], e['seconds_png_drain_after_last_clip']), flush=True)

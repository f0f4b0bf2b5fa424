#!/usr/bin/env python
"""The end-to-end loop of bench.py's secondary entry (disk -> loader -> GPU rasteriser -> generator -> metrics -> async PNG) in the
three precisions.   python tools/e2e_precisions.py [fp32|f16x3|fp16 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda:0')
for prec in (sys.argv[1:] or ['fp32', 'f16x3', 'fp16']):
    e = bench.pipeline_e2e(dev, 7, clips=8, workers=16, precision=prec)
    print(prec, round(e['value'], 1), 'frames/s; before the final drain', round(e['frames_per_s_before_the_final_png_drain'], 1),
          '| forward', round(e['seconds_generator_forward'], 3), 'loader wait', round(e['seconds_main_thread_waiting_for_loader_h2d'], 3),
          'drain', round(e['seconds_png_drain_after_last_clip'], 3), 'other', round(e['seconds_metrics_and_uint8_d2h_and_submit'], 3), flush=True)

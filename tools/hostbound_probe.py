#!/usr/bin/env python
"""Is a workload bound by the host's launch rate?  Time until the forward call returns (all launches issued) against time
until the GPU is done, for the small-frame workloads.   python tools/hostbound_probe.py

r03, one MI355X: 180x320 fp16 x3 clips: host 2.7 ms (798 launches, 3.4 us each) / GPU 7.1 ms; 128x128 fp32 x8: 7.0 / 24.7 ms;
720p fp16: 1.1 / 31.8 ms -- none of them is launch-rate-bound on the host."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from pnp_vcve_amd import synthetic as syn
dev = torch.device('cuda:0')
cfg = dict(syn.DEFAULT_GENERATOR_CFG)
sd = syn.make_state_dict(cfg, seed=2025)
for (wl, prec, clips) in (('lr180', 'fp16', 3), ('lr180', 'fp16', 8), ('128', 'fp32', 8), ('128', 'fp32', 1), ('720p', 'fp16', 1)):
    h, w = bench.WORKLOADS[wl]
    m = bench.build_model(cfg, sd, dev, prec)
    _, a = bench.make_inputs(1, 7, h, w, dev, n=clips)
    def fwd():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    host, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fwd(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print(f'{wl} {prec} clips {clips}: host returns after {np.median(host)*1e3:7.2f} ms, GPU done after {np.median(tot)*1e3:7.2f} ms  ({clips*7/np.median(tot):.0f} frames/s)')
    del m

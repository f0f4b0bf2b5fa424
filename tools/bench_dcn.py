#!/usr/bin/env python
"""Microbenchmark of the modulated-deformable-conv kernel (dcn.hip) at 720p on realistic operands: block-constant
quarter-pel flow, small learned offsets; argv: h w offset-spread."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pnp_vcve_amd import _native, ops, synthetic as syn  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (720, 1280)
spread = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5       # learned offsets ~ U(-spread, spread) px
dev = torch.device('cuda:0')
L = _native.lib()
x = torch.randn(h, w, 64, device=dev)
om = (torch.rand(h, w, 448, device=dev) * 2 - 1) * spread
blk = torch.from_numpy(syn.randint(3, 'mv', (2, h // 8, w // 8), -32, 32).astype(np.float32) / 4.0)
flow = blk.repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous().to(dev)
wp = ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
bias = torch.zeros(64, device=dev)
out = torch.empty_like(x)
P = lambda t: ctypes.c_void_p(t.data_ptr())     # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for variant in (0,):
    for use_flow in (True, False):
        fx, fy = (P(flow[0]), P(flow[1])) if use_flow else (ctypes.c_void_p(0), ctypes.c_void_p(0))
        for _ in range(3):
            L.pnp_dcn_nhwc_f32(P(x), P(om), fx, fy, P(wp), P(bias), P(out), h, w, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.pnp_dcn_nhwc_f32(P(x), P(om), fx, fy, P(wp), P(bias), P(out), h, w, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f'offsets U(+-{spread}) px, flow={use_flow}: '
              f'{us:8.1f} us  {2240 * h * w / us / 1e3:7.0f} GB/s of 2240 B/px')

# fp16 MFMA operands (PNP_PREC_F16)
w16 = torch.empty(9 * 4096, device=dev, dtype=torch.float16)
L.pnp_dcn_f16_image_from_f32(P(wp), P(w16), st)
for _ in range(3):
    L.pnp_dcn_nhwc_f16(P(x), P(om), P(flow[0]), P(flow[1]), P(w16), P(bias), P(out), h, w, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.pnp_dcn_nhwc_f16(P(x), P(om), P(flow[0]), P(flow[1]), P(w16), P(bias), P(out), h, w, st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f'fp16 MFMA operands, flow=True: {us:8.1f} us  {2240 * h * w / us / 1e3:7.0f} GB/s of 2240 B/px')

# per-phase shader-clock sums (include/pnpvcve_debug.h)
nq = L.pnp_dcn_trace_u64s()                # 8 u64 per wave x 8 waves per block x one block per CU of THIS device
dbg = torch.zeros(nq, dtype=torch.int64, device=dev)
L.pnp_dcn_nhwc_f32_ex(P(x), P(om), P(flow[0]), P(flow[1]), P(wp), P(bias), P(out), h, w, P(dbg), st)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nq // 64, 8, 8).astype(np.float64)
n = np.maximum(d[..., 7], 1)
for grp, nm in ((slice(0, 4), 'group A (gather, then MFMA)'), (slice(4, 8), 'group B (MFMA, then gather)')):
    print(nm, 'tiles/wave', n[:, grp].mean())
    for i, name in enumerate(['window fill', 'prologue', 'gather (8 taps)', 'MFMA (9 taps)', 'barriers + weights', 'epilogue', 'total']):
        print(f'   {name:20s} {np.mean(d[:, grp, i] / n[:, grp]):9.0f} cycles / tile')

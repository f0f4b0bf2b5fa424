#!/usr/bin/env python
"""hipMemsetAsync as a node of a captured graph: pnp_generator_forward used to zero the split-fp16 tile queue (64 bytes at the end of
its workspace) with hipMemsetAsync at the start of every clip.  Under generator.use_graphs (torch.cuda.graph capture of the whole
clip, ~300 nodes) the first replay was right, and from the second replay on those 64 bytes held pointer-like garbage when the
kernels behind the memset node read them (ROCm 7.2.0, MI355X) -- blocks then drew garbage tickets.  A fill KERNEL in its place is
replayed correctly (csrc/generator.hip).  This script is the small form: memset node + one kernel node, replayed four times.

    python tools/repro/graph_memset_node.py          (on an MI355X; prints the buffer after every replay, expected all 1)
"""
import ctypes

import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device('cuda:0')
big = torch.empty(117294080, dtype=torch.uint8, device=dev)           # the size of the workspace in the failing case
buf = big[-256:].view(torch.int32)[:16]
buf.fill_(7)
side = torch.cuda.Stream()
with torch.cuda.stream(side):                                         # warm-up outside the capture
    hip.hipMemsetAsync(buf.data_ptr(), 0, 64, torch.cuda.current_stream().cuda_stream)
    buf.add_(1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    rc = hip.hipMemsetAsync(buf.data_ptr(), 0, 64, torch.cuda.current_stream().cuda_stream)
    buf.add_(1)
print('capture rc', rc)
for i in range(4):
    torch.empty(1 << 20, device=dev).normal_()                        # allocator / stream traffic between replays
    g.replay()
    torch.cuda.synchronize()
    print('replay', i, buf.cpu().tolist())

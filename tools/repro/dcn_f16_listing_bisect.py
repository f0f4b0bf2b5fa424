#!/usr/bin/env python
"""Which instruction class is involved in the fp16 DCN kernel's wrong, run-to-run varying samples (r03: csrc/dcn.hip built with hipcc's
SLP vectoriser, run-time gather / MFMA order; tools/repro/dcn_f16_hazard.py)?

Round 6 found the Winograd kernel's "packed fp32" failure by looking at the listing (pnp_vcve_amd/isa_hazards.py); the DCN listing
holds no such store.  This script asks the listing directly: ONE bad source, ONE compile to a listing, then one library per
instruction class with `s_nop 1` put behind (or in front of) every instruction of that class inside dcn_window_kernel<true> -- the same
instructions otherwise -- and 12 runs of each on the same input.  A class whose padding makes the kernel deterministic and right holds
the late reader.

    python tools/repro/dcn_f16_listing_bisect.py [--reps 12] [--only CLASS,...] [--subset lo:hi]      (on an MI355X)
"""
import argparse
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dcn_f16_hazard import CSRC, ENTRY, GOOD  # noqa: E402
from pnp_vcve_amd import isa_hazards  # noqa: E402

HIPCC = '/opt/rocm/bin/hipcc'
LLVM = isa_hazards.LLVM
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-w', '-I', CSRC]
KERNEL = '_ZN12_GLOBAL__N_117dcn_window_kernelILb1EEEv7DcnArgs'

CLASSES = {                      # name -> (regex on the mnemonic, 'after' | 'before')
    'none': (r'^$', 'after'),
    'all': (r'^[vsdbg]', 'after'),
    'v_pk_any': (r'^v_pk_', 'after'),
    'v_pk_f32': (r'^v_pk_(add|mul|fma|mov)_(f32|b32)', 'after'),
    'before_v_pk_f32': (r'^v_pk_(add|mul|fma|mov)_(f32|b32)', 'before'),
    'v_pk_f16': (r'^v_pk_.*(f16|i16|u16)', 'after'),
    'v_mfma': (r'^v_mfma', 'after'),
    'before_v_mfma': (r'^v_mfma', 'before'),
    'ds_write': (r'^ds_write', 'after'),
    'ds_read': (r'^ds_read', 'after'),
    'vmem_load': (r'^(buffer|global|flat|scratch)_load', 'after'),
    'vmem_store': (r'^(buffer|global|flat|scratch)_store', 'after'),
    'v_cvt': (r'^v_cvt', 'after'),
    'trans': (r'^v_(exp|rcp|rsq|sqrt|log|sin|cos)_', 'after'),
    'v_mov64': (r'^v_(mov_b64|lshlrev_b64|lshl_add_u64|mad_[iu]64)', 'after'),
    'exec_writes': (r'^s_(and|or|andn2|xor|mov)_(saveexec_)?b64', 'after'),
    'branches': (r'^s_cbranch', 'after'),
    'waitcnt': (r'^s_waitcnt', 'after'),
    'v_cndmask': (r'^v_cndmask', 'after'),
    'accvgpr': (r'^v_accvgpr', 'after'),
    'readlane': (r'^v_(readlane|readfirstlane|writelane)|^ds_bpermute|^ds_swizzle|_dpp', 'after'),
}


def patch(text, cls, subset=None):
    """s_nop 1 beside every instruction of the class inside the fp16 kernel; subset = (lo, hi): only instances lo <= i < hi"""
    rx, where = CLASSES[cls]
    rx = re.compile(rx)
    lines = text.split('\n')
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ':'))
    b = next(i for i in range(a, len(lines)) if 's_endpgm' in lines[i])
    out, n = lines[:a], 0
    for l in lines[a:b + 1]:
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        hit = bool(m) and bool(rx.search(m.group(1))) and not m.group(1).startswith(('s_nop', 's_endpgm'))
        if hit:
            inside = subset is None or subset[0] <= n < subset[1]
            n += 1
            if inside and where == 'before':
                out.append('\ts_nop 1')
            out.append(l)
            if inside and where == 'after' and not m.group(1).startswith(('s_cbranch', 's_branch')):
                out.append('\ts_nop 1')
            elif inside and where == 'after':
                out[-1:] = ['\ts_nop 1', l]                      # (a branch: the pad goes in front of it)
        else:
            out.append(l)
    return '\n'.join(out + lines[b + 1:]), n


def build(listing, d, name):
    """patched listing -> shared library with the extern "C" entry (build_native.compile_unit's steps)"""
    lst, dev_o, dev_co, fb = (os.path.join(d, f'{name}.{e}') for e in ('s', 'o', 'co', 'hipfb'))
    with open(lst, 'w') as f:
        f.write(listing)
    subprocess.check_call([f'{LLVM}/clang', '-cc1as', '-triple', 'amdgcn-amd-amdhsa', '-filetype', 'obj', '-target-cpu', 'gfx950',
                           '-mrelocation-model', 'pic', '-o', dev_o, lst])
    subprocess.check_call([f'{LLVM}/lld', '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', dev_co, dev_o])
    subprocess.check_call([f'{LLVM}/clang-offload-bundler', '-type=o', '-bundle-align=4096',
                           '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950', '-input=/dev/null', f'-input={dev_co}', f'-output={fb}'])
    so = os.path.join(d, f'lib_{name}.so')
    subprocess.check_call([HIPCC] + FLAGS + ['-Wno-unused-command-line-argument', '--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', fb,
                                             '-shared', os.path.join(d, 'dcn_bad.hip'), '-o', so])
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=12)
    ap.add_argument('--only', default='')
    ap.add_argument('--subset', default='')
    ap.add_argument('--h', type=int, default=72)
    ap.add_argument('--w', type=int, default=80)
    args = ap.parse_args()
    import torch
    from pnp_vcve_amd import _native, ops
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    src = open(os.path.join(CSRC, 'dcn.hip')).read()
    assert src.count(GOOD) == 1
    d = tempfile.mkdtemp(prefix='dcnbis_')
    with open(os.path.join(d, 'dcn_bad.hip'), 'w') as f:
        f.write(src.replace(GOOD, 'if (grp == 0) {') + ENTRY)
    raw = os.path.join(d, 'raw.s')
    subprocess.check_call([HIPCC] + FLAGS + ['-Wno-unused-command-line-argument', '--cuda-device-only', '-S', '-o', raw, os.path.join(d, 'dcn_bad.hip')])
    text, padded = isa_hazards.patch_listing(open(raw).read())
    print(f'listing: {len(text.splitlines())} lines, {padded} wide-store hazard sites padded by the build as usual', flush=True)
    h, w = args.h, args.w
    x = torch.randn(h, w, 64, device=dev)
    off = torch.randn(288, h, w, device=dev) * 1.5
    mk = torch.randn(144, h, w, device=dev)
    blk = (torch.randint(-16, 17, (2, (h + 7) // 8, (w + 7) // 8), device=dev).float() / 4)
    flow = blk.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :w].contiguous()
    wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    bias = torch.randn(64, device=dev) * 0.1
    ref16 = ops.modulated_deform_conv_nhwc(x, off, mk, wt, bias, flow=flow, fp16=True)
    L = _native.lib()
    refc = torch.tensor([L.pnp_dcn_ref_channel(c) for c in range(448)], device=dev)
    srcm = torch.cat([off, mk], 0)
    om = torch.zeros((448, h, w), device=dev)
    om[refc >= 0] = srcm[refc[refc >= 0]]
    om = om.permute(1, 2, 0).contiguous()
    wp = ops.pack_conv3x3(wt)
    w16 = torch.empty(9 * 4096, device=dev, dtype=torch.float16)
    _native.check(L.pnp_dcn_f16_image_from_f32(ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(w16.data_ptr()), None), 'img')
    fx, fy = flow[0].contiguous(), flow[1].contiguous()
    torch.cuda.synchronize()
    subset = tuple(int(v) for v in args.subset.split(':')) if args.subset else None
    for cls in ([c for c in args.only.split(',') if c] or list(CLASSES)):
        listing, n = patch(text, cls, subset)
        so = build(listing, d, cls + (f'_{subset[0]}_{subset[1]}' if subset else ''))
        fn = ctypes.CDLL(so).repro_dcn_f16
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        outs = []
        for _ in range(args.reps):
            o = torch.empty_like(x)
            rc = fn(x.data_ptr(), om.data_ptr(), fx.data_ptr(), fy.data_ptr(), w16.data_ptr(), bias.data_ptr(), o.data_ptr(), h, w, None)
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs.append(o)
        distinct = 1 + sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
        d16 = max(float((o - ref16).abs().max()) for o in outs)
        bad = max(int(((o - ref16).abs() > 1e-3).sum()) for o in outs)
        verdict = 'OK' if distinct == 1 and d16 == 0.0 else 'WRONG'
        print(f'{cls:18s} {n:5d} instances padded{"" if subset is None else " (subset %d:%d)" % subset}:  {verdict:6s} runs differing from run 0 {distinct - 1:2d}/{args.reps - 1}  '
              f'max|d| vs the shipped fp16 kernel {d16:.3e}  wrong elements (worst run) {bad}', flush=True)


if __name__ == '__main__':
    main()

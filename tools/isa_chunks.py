#!/usr/bin/env python
"""Per-chunk instruction statistics of one conv3x3_wino_kernel instantiation from hipcc's -S output (chunks = code between
s_barriers): MFMAs, scratch traffic, AGPR <-> VGPR moves, vector-ALU adds, LDS / buffer traffic, vmcnt(0) waits.

    hipcc <FLAGS + EXTRA_FLAGS of build_native.py> -S --cuda-device-only -o wino.s pnp_vcve_amd/csrc/conv_wino.hip
    python tools/isa_chunks.py wino.s 0000        # PAR RES MS FO as four 0/1 digits; `all` = one summary line per instantiation
"""
import re
import sys


def bodies(path):
    s = open(path).read().split('\n')
    out = {}
    for k, l in enumerate(s):
        m = re.match(r'^(_ZN12_GLOBAL__N_1\d+conv3x3_wino\w*kernel\w+):', l)
        if m:
            j = k
            while not s[j].startswith('\t.end_amdhsa_kernel'):
                j += 1
            out[m.group(1)] = s[k:j]
    return out


def stats(c):
    cnt = lambda pat: sum(1 for l in c if re.search(pat, l))      # noqa: E731
    return dict(lines=len(c), mfma=cnt('v_mfma'), sld=cnt('scratch_load'), sst=cnt('scratch_store'), ard=cnt('v_accvgpr_read'),
                awr=cnt('v_accvgpr_write'), vadd=cnt(r'v_(add|sub|subrev)_f32'), vfma=cnt(r'v_(fma|fmac|mul)_f32'), dsr=cnt('ds_read'), dsw=cnt('ds_write'),
                bld=cnt('buffer_load'), bst=cnt('buffer_store'), vm0=cnt(r'vmcnt\(0\)'), nop=cnt('s_nop'), vmov=cnt(r'v_mov_b32'))


def main():
    path, which = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'all'
    b = bodies(path)
    for name, body in b.items():
        m = re.search(r'kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)E', name)
        tag = ''.join(m.groups()) if m else name[16:60]
        if which == 'all':
            print(tag, ' '.join(f'{k}={v}' for k, v in stats(body).items()))
            continue
        if tag != which:
            continue
        seg, cur = [], []
        for l in body:
            cur.append(l)
            if 's_barrier' in l:
                seg.append(cur)
                cur = []
        seg.append(cur)
        for k, c in enumerate(seg):
            print(f'{k:3d}', ' '.join(f'{a}={v}' for a, v in stats(c).items()))


main()

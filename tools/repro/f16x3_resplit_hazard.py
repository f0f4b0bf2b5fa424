#!/usr/bin/env python
"""Repro harness for DESIGN.md 3.6 finding 5: the split-fp16 conv kernel's GENERAL partition re-split (par_j(pixel) * x, re-split
into hi / lo fp16 on the vector ALUs inside the dealt MFMA block of a branch chunk) written as float-VECTOR code gave wrong,
run-to-run varying results in an intermediate r04 build; as scalar code built with -fno-slp-vectorize it is bit-stable.

Builds four variants of csrc/conv_f16x3.hip -- {scalar, vector} re-split x {with, without} -fno-slp-vectorize -- links each with
the other objects of the in-tree library, swaps it in, and runs a probe in a child process: the front half on general partition
maps, 8 launches under unrelated traffic, compared bit for bit with the first launch and with the fp64 contraction.  Also counts
the v_pk_*_f32 instructions of the PAR instantiation.  The in-tree library is restored at the end.

    python tools/repro/f16x3_resplit_hazard.py --out gpurun_out/f16x3_hazard          (on an MI355X)
"""
import argparse
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'pnp_vcve_amd', 'csrc')
LIB = os.path.join(ROOT, 'pnp_vcve_amd', 'lib', 'libpnpvcve_hip.so')
OBJ = os.path.join(ROOT, 'pnp_vcve_amd', 'lib', 'obj')

SCALAR_HEAD = '                        h8 nh, nl;\n'
SCALAR_TAIL = '                        xa[2 + r] = nl;\n'
VECTOR = '''                        typedef float f32x8 __attribute__((ext_vector_type(8)));
                        f32x8 xf = (__builtin_convertvector(fa[r], f32x8) + __builtin_convertvector(fa[2 + r], f32x8) * X3_INV) * pj;
                        xf = __builtin_elementwise_min(__builtin_elementwise_max(xf, (f32x8)(-65504.f)), (f32x8)(65504.f));
                        const h8 nh = __builtin_convertvector(xf, h8);
                        xa[r] = nh;
                        xa[2 + r] = __builtin_convertvector((xf - __builtin_convertvector(nh, f32x8)) * X3_SCALE, h8);
'''

PROBE = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests')
import numpy as np, torch, torch.nn.functional as F
import golden_util as gu
from pnp_vcve_amd import ops
h, w = 264, 272
def U(tag, shape, lo, hi): return gu.syn.uniform(38, tag, shape, lo, hi)
x, wt, b, gam = U('x', (1, 64, h, w), -1, 1), U('w', (64, 64, 3, 3), -0.06, 0.06), U('b', (64,), -0.1, 0.1), U('g', (64,), 0.5, 1.5)
w1 = [U('w1_%%d' %% j, (64, 64, 1, 1), -0.1, 0.1) * np.float32(40) for j in range(3)]
par = U('par', (3, h, w), 0.0, 1.0 / 255.0)                   # general (non-binary) values: every tile takes the re-split form
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
xs, pg = ops.nchw_to_nhwc(G(x))[0], G(par)
kw = dict(bias=G(b), gamma=G(gam), packed_w1x1=ops.pack_conv1x1([G(v) for v in w1]), par=pg, par_flags=ops.par_tile_flags(pg), act=1)
pw = [ops.pack_conv3x3(G(wt))]
ref = F.conv2d(D(x), D(wt), D(b), padding=1) * D(gam).view(1, 64, 1, 1)
for j in range(3):
    ref = ref + D(par[j]).view(1, 1, h, w) * F.conv2d(D(x), D(w1[j]))
ref = F.relu(ref)
first = ops.conv3x3_f16x3([xs], pw, **kw).clone()
worst, nbad = 0.0, 0
for rep in range(8):
    torch.randn(1 << 22, device='cuda').sin_()
    o = ops.conv3x3_f16x3([xs], pw, **kw)
    d = (o - first).abs()
    worst, nbad = max(worst, float(d.max())), max(nbad, int((d > 0).sum()))
err = float((ops.nhwc_to_nchw(first.unsqueeze(0)).cpu().double() - ref).abs().max())
print('RESULT run_to_run_max %%.3e differing_values %%d of %%d  max_abs_err_vs_fp64 %%.3e' %% (worst, nbad, first.numel(), err))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/f16x3_hazard')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    from pnp_vcve_amd import build_native
    build_native.build(verbose=False)
    src = open(os.path.join(CSRC, 'conv_f16x3.hip')).read()
    i, j = src.index(SCALAR_HEAD), src.index(SCALAR_TAIL) + len(SCALAR_TAIL)
    forms = {'scalar': src, 'vector': src[:i] + VECTOR + src[j:]}
    others = [os.path.join(OBJ, f) for f in sorted(os.listdir(OBJ)) if f.endswith('.o') and f != 'conv_f16x3.o']
    backup = os.path.join(a.out, 'lib_in_tree.so')
    shutil.copy(LIB, backup)
    probe = os.path.join(a.out, 'probe.py')
    open(probe, 'w').write(PROBE % dict(root=ROOT))
    report = []
    try:
        for form, text in forms.items():
            for flag in ('-fno-slp-vectorize', ''):
                tag = f'{form}{"_noslp" if flag else "_slp"}'
                cu = os.path.join(CSRC, f'_repro_{tag}.hip')            # beside the real file: its #includes are relative
                open(cu, 'w').write(text)
                obj = os.path.join(a.out, tag + '.o')
                base = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + ([flag] if flag else [])
                try:
                    subprocess.check_call(base + ['-c', cu, '-o', obj], stderr=subprocess.DEVNULL)
                    asm = os.path.join(a.out, tag + '.s')
                    subprocess.check_call(base + ['-S', '--cuda-device-only', cu, '-o', asm], stderr=subprocess.DEVNULL)
                finally:
                    os.remove(cu)
                body = open(asm).read()
                k = body.index('conv3x3_f16x3_kernelILb1ELb0ELb0ELb0EEE')           # PAR, no trace, 8x16 tiles
                fn = body[k:body.index('.end_amdhsa_kernel', k) if '.end_amdhsa_kernel' in body[k:] else None]
                fn = fn[:fn.index('s_endpgm')] if 's_endpgm' in fn else fn
                npk = len(re.findall(r'v_pk_(?:mul|add|fma|max|min)_f32', fn))
                subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + others + [obj])
                out = subprocess.run([sys.executable, probe], capture_output=True, text=True)
                res = [ln for ln in out.stdout.splitlines() if ln.startswith('RESULT')]
                line = f'{tag:14s} v_pk_*_f32 in the PAR kernel: {npk:4d}   ' + (res[0][7:] if res else 'PROBE FAILED: ' + out.stderr[-300:])
                print(line, flush=True)
                report.append(line)
    finally:
        shutil.copy(backup, LIB)
    open(os.path.join(a.out, 'report.txt'), 'w').write('\n'.join(report) + '\n')


if __name__ == '__main__':
    main()

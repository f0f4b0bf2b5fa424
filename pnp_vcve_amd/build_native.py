"""Builds libpnpvcve_hip.so (gfx950) in-tree with hipcc.  No torch involved."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libpnpvcve_hip.so')
SOURCES = ['conv_mfma.hip', 'conv_persist.hip', 'conv_wino.hip', 'conv_wino_ms.hip', 'conv_f16.hip', 'conv_f16x3.hip', 'conv_last.hip', 'warp.hip', 'prep.hip', 'metrics.hip', 'raster.hip', 'dcn.hip', 'generator.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# per-source extras.  dcn.hip: hipcc's SLP vectoriser packs the scalar coordinate / weight arithmetic of the deformable gather into
# v_pk_*_f32 pairs; every build with that packing gave wrong, run-to-run varying samples in the fp16 instantiation under some
# timing (the round-2 code shape rarely, others always), every build without it is correct under every perturbation tried
# (tools/repro/dcn_f16_hazard.py, profiles/r03_dcn_hazard_report.txt).
EXTRA_FLAGS = {'dcn.hip': ['-fno-slp-vectorize'],
               # conv_f16x3.hip (r04): an intermediate build of the general partition re-split (float-vector code inside a cut-up chunk)
               # showed the same signature; the committed structure is bit-stable with or without the flag
               # (tools/repro/f16x3_resplit_hazard.py) -- kept as the conservative form, scalar code + this flag
               'conv_f16x3.hip': ['-fno-slp-vectorize'],
               # conv_wino.hip (r05): the same signature a third time -- with v_pk_{add,mul}_f32 in the epilogue / the input transform, a few
               # fixed (lane, register) slots of the output came out wrong, deterministically, and moved when unrelated code moved
               # (tools/repro/wino_packed_f32_hazard.py; profiles/r05_wino_packed_f32_hazard.txt).  The kernel is built without packed fp32 VALU ops at all
               #   -pragma-unroll-threshold: the K loop of a tile is ~1200 MFMAs unrolled from `#pragma unroll` loops; past LLVM's default
               #   limit of 16384 (estimated) instructions a loop silently stays rolled, the accumulator array gets indexed dynamically
               #   and lands in scratch MEMORY (private_segment 1616 B, results right, ten times slower).  That is what "source orders
               #   hipcc does not like" were in DESIGN.md 8; with the limit out of the way partial effects go too (22 -> 13, 65 -> 46
               #   spill slots in the residual kernels)
               'conv_wino.hip': None, 'conv_wino_ms.hip': None}
_WINO = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops', '-mllvm', '-pragma-unroll-threshold=1000000']
# conv_wino.hip (r06): the tile kernels keep all 64 accumulator quads in AGPRs only if the allocator does not park some of them in arch
# VGPRs between uses: with LLVM's default local assignment order a tile's K loop holds ~100 v_accvgpr_read + ~70 v_accvgpr_write (each
# one a vector-ALU instruction in an MFMA gap, the reads behind an s_nop); -greedy-reverse-local-assignment leaves ~30 reads and no
# write (tools/isa_chunks.py; block convs 333 -> 329 us).  The multi-source instantiation (conv_wino_ms.hip = the same source,
# WINO_MS_TU; a run-time loop over the sources with the accumulators carried around it) is its own translation unit because it LOST
# with that flag while its ring / halo still went through registers (316 B of scratch; 79.1 -> 77.3 frames/s when everything was built
# with it).  Since the LDS-DMA ring (28 staging registers gone) the flag is right for it too: 228 + 228 accumulator moves per source
# segment -> 0, 20 B of scratch.  The unit stays separate: its flags have diverged once.
EXTRA_FLAGS['conv_wino.hip'] = _WINO + ['-mllvm', '-greedy-reverse-local-assignment']
EXTRA_FLAGS['conv_wino_ms.hip'] = _WINO + ['-mllvm', '-greedy-reverse-local-assignment']
# sources a translation unit #includes (beyond the headers)
INCLUDES = {'conv_wino_ms.hip': ['conv_wino.hip']}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.getmtime(d) > mt for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(HERE, 'lib', 'obj')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'pnpvcve.h'))
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s, os.path.abspath(__file__)] + headers + [os.path.join(CSRC, i) for i in INCLUDES.get(src, [])]):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

"""Builds libpnpvcve_hip.so (gfx950) in-tree with hipcc.  No torch involved.

Every translation unit goes device code -> assembly listing -> isa_hazards.patch_listing -> assembler -> code object -> bundle, and the
host pass embeds that bundle (the steps `hipcc -c` runs itself, with one edit of the listing in between): gfx950 needs a wait state
between a 16-byte store and a vector-ALU write of its data registers that LLVM does not insert for the store form the Winograd
epilogue uses, and a source-level fence costs that kernel its register allocation (isa_hazards.py)."""
import os
import subprocess
import sys

from . import isa_hazards

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libpnpvcve_hip.so')
SOURCES = ['conv_mfma.hip', 'conv_persist.hip', 'conv_wino.hip', 'conv_wino_ms.hip', 'conv_f16.hip', 'conv_f16x3.hip', 'conv_last.hip', 'warp.hip', 'prep.hip', 'metrics.hip', 'raster.hip', 'dcn.hip', 'generator.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# per-source extras.  Rounds 3-5 met a signature three times -- wrong values whenever hipcc packed fp32 arithmetic into v_pk_*_f32
# (dcn.hip r03, conv_f16x3.hip r04, conv_wino.hip r05; tools/repro/*_hazard.py, profiles/r0[345]_*hazard*) -- and fenced it off per
# file with the flags below.  Round 6 found the cause of the Winograd case: a 16-byte store followed at once by a vector-ALU write of
# its data registers (isa_hazards.py; a packed op writes two of them per instruction and hits the window ten times as often).  The
# build pads that in every unit's listing, and the all-packed Winograd build passes every test -- but runs 20 % slower (the allocator
# spills in the tile loop), so its flag stays as a measured-good code shape.  dcn.hip and conv_f16x3.hip have no such site in their
# listings with or without SLP packing: their failures are NOT explained by it and -fno-slp-vectorize stays a fence there.
EXTRA_FLAGS = {'dcn.hip': ['-fno-slp-vectorize'],
               'conv_f16x3.hip': ['-fno-slp-vectorize'],
               # conv_wino.hip: built without packed fp32 VALU ops (profiles/r06_wino_ab.txt: 89.3 frames/s; all packed 71.7; packed sums
               # only: +0.6 % with every product written element by element -- not taken)
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


def _run(cmd, verbose):
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)


def compile_unit(src, obj, flags, verbose=False, hipcc=None):
    """src (.hip) -> obj: the device listing is padded against the store hazard before it is assembled; -> number of stores padded.
    If a tool of the split pipeline is missing or fails (another ROCm layout), the unit is compiled by `hipcc -c` in one go and the
    object is linted instead: a unit with an unpadded site is an error, never a silent build."""
    hipcc = hipcc or os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    try:
        return _compile_unit_padded(src, obj, flags, verbose, hipcc)
    except (subprocess.CalledProcessError, OSError) as e:
        print(f'build_native: split pipeline failed for {os.path.basename(src)} ({e}); compiling in one step and linting the object', flush=True)
        _run([hipcc] + flags + ['-c', src, '-o', obj], verbose)
        try:
            sites, _ = isa_hazards.lint_library(obj)
        except (subprocess.CalledProcessError, OSError) as e2:
            print(f'build_native: WARNING: {os.path.basename(src)} could not be disassembled either ({e2}): built UNCHECKED for the wide-store hazard '
                  f'(python tools/lint_store_hazard.py on a machine with the ROCm LLVM tools)', flush=True)
            return 0
        if sites:
            raise RuntimeError(f'{os.path.basename(src)}: {len(sites)} wide store(s) followed by a write of their data registers and no way to pad them '
                               f'here (isa_hazards.py): {sites[:2]}')
        return 0


def _compile_unit_padded(src, obj, flags, verbose, hipcc):
    llvm = isa_hazards.LLVM
    stem = obj[:-2] if obj.endswith('.o') else obj
    lst, dev_o, dev_co, fb = stem + '.gfx950.s', stem + '.gfx950.o', stem + '.gfx950.co', stem + '.hipfb'
    _run([hipcc] + flags + ['-Wno-unused-command-line-argument', '--cuda-device-only', '-S', '-o', lst, src], verbose)
    with open(lst) as f:
        text, padded = isa_hazards.patch_listing(f.read())
    with open(lst, 'w') as f:
        f.write(text)
    if verbose and padded:
        print(f'   {os.path.basename(src)}: s_nop behind {padded} store(s) whose data registers the next instruction writes', flush=True)
    _run([f'{llvm}/clang', '-cc1as', '-triple', 'amdgcn-amd-amdhsa', '-filetype', 'obj', '-target-cpu', 'gfx950', '-mrelocation-model', 'pic',
          '-o', dev_o, lst], verbose)
    _run([f'{llvm}/lld', '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', dev_co, dev_o], verbose)
    _run([f'{llvm}/clang-offload-bundler', '-type=o', '-bundle-align=4096', '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950',
          '-input=/dev/null', f'-input={dev_co}', f'-output={fb}'], verbose)
    host, skip = [], 0
    for i, f in enumerate(flags):                                  # (device-only target features mean nothing to the host pass)
        if skip:
            skip -= 1
        elif f == '-Xclang' and flags[i + 1:i + 2] == ['-target-feature']:
            skip = 3
        else:
            host.append(f)
    _run([hipcc] + host + ['-Wno-unused-command-line-argument', '--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', fb, '-c', src, '-o', obj], verbose)
    for tmp in (lst, dev_o, dev_co, fb):
        os.remove(tmp)
    return padded


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
        deps = [s, os.path.abspath(__file__), os.path.join(HERE, 'isa_hazards.py')] + headers + [os.path.join(CSRC, i) for i in INCLUDES.get(src, [])]
        if force or _stale(o, deps):
            compile_unit(s, o, FLAGS + EXTRA_FLAGS.get(src, []), verbose, hipcc)
    if force or _stale(LIB, objs):
        _run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, verbose)
    return LIB


def build_variant(name, extra, verbose=False):
    """lib/ab/lib_<name>.so: the current library with conv_wino.hip and conv_wino_ms.hip recompiled with `extra` flags added
    (tools/build_wino_variants.sh; run in turn with tools/try_libs.sh)"""
    build(verbose=False)
    ab = os.path.join(HERE, 'lib', 'ab')
    os.makedirs(ab, exist_ok=True)
    objdir = os.path.join(HERE, 'lib', 'obj')
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in SOURCES if not s.startswith('conv_wino')]
    for src in ('conv_wino.hip', 'conv_wino_ms.hip'):
        o = os.path.join(ab, src.replace('.hip', f'_{name}.o'))
        compile_unit(os.path.join(CSRC, src), o, [f for f in FLAGS if f != '-Wall'] + ['-w'] + EXTRA_FLAGS[src] + extra, verbose)
        objs.append(o)
    out = os.path.join(ab, f'lib_{name}.so')
    _run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, verbose)
    return out


if __name__ == '__main__':
    if '--variant' in sys.argv:
        i = sys.argv.index('--variant')
        print('built', build_variant(sys.argv[i + 1], sys.argv[i + 2].split(), verbose='-v' in sys.argv))
    else:
        print(build(force='--force' in sys.argv, verbose=True))

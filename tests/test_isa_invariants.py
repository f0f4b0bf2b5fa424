"""ISA-level invariants of the split-fp16 conv kernel that its correctness rests on and that no arithmetic test can see
(csrc/conv_f16x3.hip; DESIGN.md 3.6 finding 5).  hipcc cross-compiles gfx950 here: no GPU needed."""
import os
import re
import subprocess

import pytest

from pnp_vcve_amd import build_native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


def _counted_barrier_slack(src, tmp_path, name_re, min_sites):
    """per kernel whose mangled name matches name_re: [LDS instructions behind the last ds_write_b128 - K] for every
    `s_waitcnt lgkmcnt(K); s_barrier` with K in (4, 6)"""
    asm = tmp_path / (src + '.s')
    flags = build_native.FLAGS + build_native.EXTRA_FLAGS.get(src, [])
    subprocess.check_call([HIPCC] + [f for f in flags if f != '-Wall'] + ['-S', '--cuda-device-only', '-o', str(asm),
                          os.path.join(ROOT, 'pnp_vcve_amd', 'csrc', src)], stderr=subprocess.DEVNULL)
    out = {}
    for fn in re.split(r'\n(?=_Z\w+:\s)', asm.read_text()):
        if not re.match(name_re, fn):
            continue
        lines = fn.split('\n')
        sites = [i for i, l in enumerate(lines) if re.search(r's_waitcnt lgkmcnt\((4|6)\)', l) and 's_barrier' in lines[i + 1]]
        if not sites:
            continue
        assert len(sites) >= min_sites, lines[0]
        behind = []
        for i in sites:
            k = int(re.search(r'lgkmcnt\((\d)\)', lines[i]).group(1))
            n, j = 0, i - 1
            while j >= 0 and 'ds_write_b128' not in lines[j]:
                n += bool(re.search(r'\bds_(read|write)', lines[j]))
                j -= 1
            assert j >= 0 and n >= k, (lines[0], i, k, n)
            behind.append(n - k)
        out[lines[0]] = behind
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_f16_small_kernel_counted_chunk_barrier_still_covers_the_ring_write(tmp_path):
    """the same invariant in conv3x3_f16_small_kernel (csrc/conv_f16.hip: ring write in k-step 1 of a chunk, lgkmcnt(6) = the
    fragment reads of k-steps 2 and 3)"""
    slack = _counted_barrier_slack('conv_f16.hip', tmp_path, r'_Z\w*conv3x3_f16_small_kernel\w*:', 8)
    assert len(slack) >= 2 and all(min(v) == 0 for v in slack.values())


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_f16x3_counted_chunk_barrier_still_covers_the_ring_write(tmp_path):
    """The per-chunk barrier of the K loop is `s_waitcnt lgkmcnt(K); s_barrier` with K = 6 (8x16 tiles) / 4 (4x16 tiles): LDS
    operations complete in order, so "at most K outstanding" means the ring write (ds_write_b128) of the chunk has landed ONLY IF at
    least K LDS operations were issued behind it -- the fragment reads of the following quarters, placed there by the chunk's
    sched_group_barrier sequence.  A scheduler that moved one of those reads in front of the write would let the barrier release
    waves that then read a half-written weight chunk: wrong, run-to-run varying results and no compile error.  Checked on the
    compiler's own output for every instantiation: behind the last ds_write_b128 in front of each such barrier there are >= K LDS
    instructions, and somewhere exactly K (the count is not slack)."""
    slack = _counted_barrier_slack('conv_f16x3.hip', tmp_path, r'_Z\w*conv3x3_f16x3_kernel\w*:', 17)      # one site per chunk but the last
    assert len(slack) >= 8                                     # {PAR, plain, MS} x {8x16, 4x16} x {trace}
    assert all(min(v) == 0 for v in slack.values())


def test_winograd_kernel_is_built_without_packed_fp32_valu_ops(tmp_path):
    """csrc/conv_wino.hip with v_pk_{add,mul}_f32 in its code object gave wrong values in fixed (lane, register) slots that moved with
    unrelated code motion (profiles/r05_wino_packed_f32_hazard.txt) -- the third appearance of that signature (DCN r03, split-fp16
    r04).  The file is compiled with the packed-fp32 target feature off; this fails if the flag is dropped or stops working."""
    assert '-packed-fp32-ops' in build_native.EXTRA_FLAGS['conv_wino.hip']
    asm = tmp_path / 'conv_wino.s'
    flags = build_native.FLAGS + build_native.EXTRA_FLAGS['conv_wino.hip']
    subprocess.check_call([HIPCC] + [f for f in flags if f != '-Wall'] + ['-S', '--cuda-device-only', '-o', str(asm),
                          os.path.join(ROOT, 'pnp_vcve_amd', 'csrc', 'conv_wino.hip')], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    kernels = [fn for fn in re.split(r'\n(?=_Z\w+:\s)', text) if 'conv3x3_wino_kernel' in fn.split('\n')[0]]
    assert len(kernels) == 7                                   # {plain, residual, branches, branches + residual, multi-source, fold-only (+ residual)}
    # the accumulators and the transformed patch live in REGISTERS: a source order hipcc does not like once put both arrays into scratch
    # memory (private_seg_size 1616: correct results, ten times slower) -- a few spill slots are tolerated
    sizes = [int(v) for v in re.findall(r'conv3x3_wino_kernel\w+\.private_seg_size, (\d+)', text)]
    assert len(sizes) == 7 and max(sizes) <= 512, sizes
    # the quadrant-unit kernels of small frames (five instantiations): straight-line code, no spill slot at all, no packed fp32 either
    units = [fn for fn in re.split(r'\n(?=_Z\w+:\s)', text) if 'conv3x3_wino_quad' in fn.split('\n')[0]]
    usizes = [int(v) for v in re.findall(r'conv3x3_wino_quad\w+\.private_seg_size, (\d+)', text)]
    assert len(units) == 5 and len(usizes) == 5 and max(usizes) == 0, usizes
    for fn in units:
        assert fn.count('v_mfma_f32_16x16x4_f32') >= 256 and not re.search(r'\bv_pk_(add|mul|fma)_f32\b', fn)
    for fn in kernels:
        assert fn.count('v_mfma_f32_16x16x4_f32') >= 1024                           # the K loop is there ...
        assert not re.search(r'\bv_pk_(add|mul|fma)_f32\b', fn), fn.split('\n')[0]    # ... and no packed fp32 arithmetic beside it

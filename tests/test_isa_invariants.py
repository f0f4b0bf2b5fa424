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


def _wino_asm(tmp_path, src):
    asm = tmp_path / (src + '.s')
    flags = build_native.FLAGS + build_native.EXTRA_FLAGS[src]
    subprocess.check_call([HIPCC] + [f for f in flags if f != '-Wall'] + ['-S', '--cuda-device-only', '-o', str(asm),
                          os.path.join(ROOT, 'pnp_vcve_amd', 'csrc', src)], stderr=subprocess.DEVNULL)
    return asm.read_text()


def _tile_loop(fn, tail_mfmas):
    """the lines of one tile's K loop: first MFMA .. last MFMA in front of the quadrant-unit tail (its `tail_mfmas` last ones)"""
    lines = fn.split('\n')
    mf = [i for i, l in enumerate(lines) if 'v_mfma_f32_16x16x4_f32' in l]
    return lines[mf[0]:mf[len(mf) - tail_mfmas - 1]]


def test_winograd_kernel_is_built_without_packed_fp32_valu_ops(tmp_path):
    """Both Winograd translation units are compiled with the packed-fp32 target feature off.  Through round 5 that was a fence: builds
    with v_pk_{add,mul}_f32 gave wrong values in fixed (lane, register) slots (profiles/r05_wino_packed_f32_hazard.txt).  Round 6 found
    the cause -- a 16-byte store followed at once by a write of its data registers, which a packed op
    hits ten times as often (pnp_vcve_amd/isa_hazards.py; padded in the build, checked below) -- and with it padded the all-packed
    build is correct but 20 % slower (the allocator spills in the tile loop: 71.7 vs 89.3 frames/s, profiles/r06_wino_ab.txt), so the
    flag stays, as a performance choice.  Also bounded here, on the compiler's own output: the spill slots of every instantiation and
    the scratch traffic INSIDE the K loop of the default path's kernels (a spill reload is an s_waitcnt vmcnt(0): it drains the weight /
    halo requests in flight)."""
    for src in ('conv_wino.hip', 'conv_wino_ms.hip'):
        assert '-packed-fp32-ops' in build_native.EXTRA_FLAGS[src] and '-pragma-unroll-threshold=1000000' in build_native.EXTRA_FLAGS[src]
    text = _wino_asm(tmp_path, 'conv_wino.hip')
    kernels = {re.search(r'kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)E', fn.split('\n')[0]).groups(): fn
               for fn in re.split(r'\n(?=_Z\w+:\s)', text) if 'conv3x3_wino_kernel' in fn.split('\n')[0]}
    # {plain, residual, branches, branches + residual}; the fold-only bodies live in the two gated kernels (fold-only | branch body behind
    # the frame's partition word), the multi-source one in conv_wino_ms.hip
    assert sorted(kernels) == sorted([('0', '0', '0', '0'), ('0', '1', '0', '0'), ('1', '0', '0', '0'), ('1', '1', '0', '0')])
    gated = [fn for fn in re.split(r'\n(?=_Z\w+:\s)', text) if 'conv3x3_wino_gated_kernel' in fn.split('\n')[0]]
    assert len(gated) == 2
    for fn in gated:
        assert fn.count('v_mfma_f32_16x16x4_f32') >= 1280 + 2240 and not re.search(r'\bv_pk_(add|mul|fma)_f32\b', fn)
        # the fold-only body comes first: its K loop (the first 1024 MFMAs) stays clear of scratch traffic like the standalone kernels'
        lines = fn.split('\n')
        mf = [i for i, l in enumerate(lines) if 'v_mfma_f32_16x16x4_f32' in l]
        assert sum(1 for l in lines[mf[0]:mf[1023]] if 'scratch_' in l) <= 4
    gsizes = [int(v) for v in re.findall(r'conv3x3_wino_gated_kernel\w+\.private_seg_size, (\d+)', text)]
    assert len(gsizes) == 2 and max(gsizes) <= 160, gsizes
    # the accumulators and the transformed patch live in REGISTERS: a source order hipcc does not like once put both arrays into scratch
    # memory (private_seg_size 1616: correct results, ten times slower).  Round 6 (patch rows read where they are
    # transformed): 40-52 B for the kernels of the default path (r05: 84-176), <= 128 B for the branch kernels (208-236)
    sizes = {tuple(k): int(v) for *k, v in re.findall(r'conv3x3_wino_kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)E\w+\.private_seg_size, (\d+)', text)}
    assert len(sizes) == 4
    for k, v in sizes.items():
        assert v <= (64 if k[0] == '0' else 160), (k, v)
    for k, fn in kernels.items():
        assert fn.count('v_mfma_f32_16x16x4_f32') >= 1024                           # the K loop is there ...
        assert not re.search(r'\bv_pk_(add|mul|fma)_f32\b', fn), k                  # ... and no packed fp32 arithmetic beside it
        if k[0] == '0':
            # plain / residual / fold-only: at most a handful of scratch instructions between a tile's first and last MFMA (r05: up to 18)
            loop = _tile_loop(fn, 256)
            n = sum(1 for l in loop if 'scratch_' in l)
            assert n <= 4, (k, n)
            # ... and few accumulator quads parked in arch VGPRs (r05: ~100 v_accvgpr_read + ~70 v_accvgpr_write per tile)
            assert sum(1 for l in loop if 'v_accvgpr_write' in l) <= 16 and sum(1 for l in loop if 'v_accvgpr_read' in l) <= 48, k
    # the quadrant-unit kernels of small frames: straight-line code, no spill slot at all, no packed fp32 either
    units = [fn for fn in re.split(r'\n(?=_Z\w+:\s)', text) if 'conv3x3_wino_quad' in fn.split('\n')[0]]
    usizes = [int(v) for v in re.findall(r'conv3x3_wino_quad\w+\.private_seg_size, (\d+)', text)]
    assert len(units) == 6 and len(usizes) == 6 and max(usizes) == 0, usizes      # {plain, RES, PAR, PAR+RES} + the two gated ones (fold-only | branch body)
    for fn in units:
        assert fn.count('v_mfma_f32_16x16x4_f32') >= 256 and not re.search(r'\bv_pk_(add|mul|fma)_f32\b', fn)


def test_winograd_multi_source_unit_is_its_own_translation_unit(tmp_path):
    """conv_wino_ms.hip = conv_wino.hip with WINO_MS_TU: exactly the multi-source tile kernel and its quadrant-unit twin, its own
    translation unit because its flags have differed (build_native.py), without packed fp32; its segment loop holds no scratch
    instruction (r05: 31 reloads + 32 stores per segment) and -- since the ring is LDS-DMA and the allocation flag applies here too --
    no accumulator move between AGPRs and arch VGPRs (before: 228 + 228 per segment)"""
    assert '-greedy-reverse-local-assignment' in build_native.EXTRA_FLAGS['conv_wino.hip']
    assert '-greedy-reverse-local-assignment' in build_native.EXTRA_FLAGS['conv_wino_ms.hip']
    text = _wino_asm(tmp_path, 'conv_wino_ms.hip')
    fns = [fn for fn in re.split(r'\n(?=_Z\w+:\s)', text) if 'conv3x3_wino' in fn.split('\n')[0]]
    names = sorted(fn.split('\n')[0] for fn in fns)
    assert len(fns) == 2 and 'quad_ms_kernel' in names[1] and 'kernelILb0ELb0ELb1ELb0E' in names[0], names
    sizes = dict(re.findall(r'(conv3x3_wino\w+)\.private_seg_size, (\d+)', text))
    assert all(int(v) <= (96 if 'quad' not in k else 0) for k, v in sizes.items()), sizes
    ms = [fn for fn in fns if 'quad' not in fn.split('\n')[0]][0]
    assert ms.count('v_mfma_f32_16x16x4_f32') >= 1024 + 64 and not re.search(r'\bv_pk_(add|mul|fma)_f32\b', text)
    lines = ms.split('\n')
    mf = [i for i, l in enumerate(lines) if 'v_mfma_f32_16x16x4_f32' in l]
    seg = lines[mf[64]:mf[-1]]                      # behind the 64 MFMAs of the frame's RGB chunks: one source's 16 chunks
    assert sum(1 for l in seg if 'scratch_' in l) <= 2, sum(1 for l in seg if 'scratch_' in l)
    assert sum(1 for l in seg if 'v_accvgpr_' in l) <= 32, sum(1 for l in seg if 'v_accvgpr_' in l)


def test_winograd_lds_dma_loads_have_landed_before_the_barrier_that_publishes_them(tmp_path):
    """The tile kernels' weight ring and halo slabs arrive as LDS-DMA loads (`buffer_load_dwordx4 ... lds`: no destination register, so no
    compiler-made wait ever covers them) and are published to the other waves by the chunk barriers.  The pipeline's contract
    (conv_wino.hip, the counted `s_waitcnt vmcnt(N)` at the end of every chunk): a DMA load issued in chunk c has landed before the
    barrier at the top of chunk c + 2 -- the ring chunk requested in chunk c is first read behind that barrier.  vmcnt retires in order,
    so `vmcnt(N)` completes everything but the last N vector-memory instructions issued.  Checked on the compiler's own output by
    replaying the instruction stream of one tile's K loop for the plain, the residual and the multi-source kernel and for the fold-only
    body inside the gated kernel: at every barrier, every LDS-DMA load issued before the PREVIOUS barrier must be complete."""
    vm = re.compile(r'^\s+(buffer_(load|store|atomic)|global_(load|store)|scratch_(load|store))')
    checked = 0
    for src, pat in (('conv_wino.hip', r'conv3x3_wino_kernelILb0ELb0ELb0ELb0E'), ('conv_wino.hip', r'conv3x3_wino_kernelILb0ELb1ELb0ELb0E'),
                     ('conv_wino.hip', r'conv3x3_wino_gated_kernelILb0E'), ('conv_wino_ms.hip', r'conv3x3_wino_kernelILb0ELb0ELb1ELb0E')):
        text = _wino_asm(tmp_path, src)
        fn = [f for f in re.split(r'\n(?=_Z\w+:\s)', text) if re.search(pat, f.split('\n')[0])][0]
        lines = fn.split('\n')
        mf = [i for i, l in enumerate(lines) if 'v_mfma_f32_16x16x4_f32' in l]
        first = 64 if 'Lb1ELb0EEEv' in lines[0] and 'kernelILb0ELb0ELb1' in lines[0] else 0       # (multi-source: behind the RGB chunks)
        loop = lines[mf[first]:mf[first + 1023]]
        issued = 0                       # vector-memory instructions issued so far
        done = 0                         # ... of which known complete (in-order retirement)
        dma = []                         # issue indices of LDS-DMA loads
        barriers = []                    # value of `issued` at each barrier
        ndma = 0
        for l in loop:
            if vm.match(l):
                issued += 1
                if l.rstrip().endswith(' lds'):
                    dma.append(issued)
                    ndma += 1
            m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
            if m:
                done = max(done, issued - int(m.group(1)))
            if 's_barrier' in l:
                if len(barriers) >= 1:
                    must = [d for d in dma if d <= barriers[-1]]             # issued before the previous barrier
                    assert all(d <= done for d in must), (lines[0][:80], len(barriers), done, must[-1])
                barriers.append(issued)
        assert len(barriers) >= 15 and ndma >= 16 * 4, (lines[0][:80], len(barriers), ndma)    # 16 chunks, 4 ring pieces each (+ halo pieces)
        checked += 1
    assert checked == 4


def test_no_wide_store_is_followed_by_a_write_of_its_data_registers(tmp_path):
    """gfx950: a vector-memory store of more than 64 bits still reads its data registers when the next instruction issues; a vector-ALU
    write of one of them then reaches memory in lanes 12-15 of each row of 16 (tools/repro/store_x4_then_wide_valu.hip,
    profiles/r06_store_x4_hazard_probe.txt).  LLVM pads that except for MUBUF stores with an SGPR soffset -- the Winograd epilogue's
    form -- so build_native.py pads the device listing itself.  Checked here: the padding pass on a listing with the pattern in all its
    forms, and the LINKED library, disassembled -- not one site may be left in any of its code objects."""
    from pnp_vcve_amd import isa_hazards
    lst = """
_Zkernel:
	buffer_store_dwordx4 v[34:37], v55, s[24:27], s8 offen
	v_pk_add_f32 v[34:35], v[230:231], v[198:199]
	buffer_store_dwordx4 v[18:21], v60, s[24:27], s8 offen
	v_add_f32_e32 v22, v1, v2
	v_add_f32_e32 v21, v1, v2
	buffer_store_dwordx4 v[10:13], v60, s[24:27], 0 offen
	s_nop 0
	v_accvgpr_read_b32 v13, a3
	buffer_store_dwordx4 v[10:13], v60, s[24:27], 0 offen
	s_nop 1
	v_accvgpr_read_b32 v13, a3
	buffer_store_dwordx2 v[10:11], v60, s[24:27], s8 offen
	v_mov_b32_e32 v10, v1
	global_store_dwordx4 v[2:3], v[6:9], off
	v_mov_b32_e32 v2, 0
	v_mov_b32_e32 v9, 0
	buffer_store_dwordx4 v[40:43], v60, s[24:27], s8 offen
.LBB0_1:
	v_cmp_gt_i32_e32 vcc, v40, v1
	buffer_store_dwordx4 v[40:43], v60, s[24:27], s8 offen
	ds_read_b128 v[40:43], v3
	s_endpgm
"""
    found = isa_hazards.lint_listing(lst.split('\n'))
    assert [(f[2].split()[1], f[3].split()[0]) for f in found] == [('v[34:37],', 'v_pk_add_f32'), ('v[10:13],', 'v_accvgpr_read_b32'),
                                                                    ('v[2:3],', 'v_mov_b32_e32')], found
    patched, n = isa_hazards.patch_listing(lst)
    assert n == 3 and not isa_hazards.lint_listing(patched.split('\n'))
    assert patched.count('s_nop') == 2 + 3 and 'offen\n\ts_nop 0 ' in patched          # one wait state behind the SGPR-soffset form
    lib = build_native.build()
    sites, objects = isa_hazards.lint_library(lib)
    assert objects == len(build_native.SOURCES)
    assert not sites, sites[:4]
    # ... and it is the padding that makes it so: the Winograd unit's own listing has such sites (when this stops being true the pass
    # has nothing left to do -- fine -- but then say so here)
    raw = _wino_asm(tmp_path, 'conv_wino.hip')
    assert len(isa_hazards.lint_listing(raw.split('\n'))) >= 1

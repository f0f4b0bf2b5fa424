"""ISA-level invariants of the split-fp16 conv kernel that its correctness rests on and that no arithmetic test can see
(csrc/conv_f16x3.hip; DESIGN.md 3.6 finding 5).  hipcc cross-compiles gfx950 here: no GPU needed."""
import os
import re
import subprocess

import pytest

from pnp_vcve_amd import build_native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_f16x3_counted_chunk_barrier_still_covers_the_ring_write(tmp_path):
    """The per-chunk barrier of the K loop is `s_waitcnt lgkmcnt(K); s_barrier` with K = 6 (8x16 tiles) / 4 (4x16 tiles): LDS
    operations complete in order, so "at most K outstanding" means the ring write (ds_write_b128) of the chunk has landed ONLY IF at
    least K LDS operations were issued behind it -- the fragment reads of the following quarters, placed there by the chunk's
    sched_group_barrier sequence.  A scheduler that moved one of those reads in front of the write would let the barrier release
    waves that then read a half-written weight chunk: wrong, run-to-run varying results and no compile error.  Checked on the
    compiler's own output for every instantiation: behind the last ds_write_b128 in front of each such barrier there are >= K LDS
    instructions, and somewhere exactly K (the count is not slack)."""
    asm = tmp_path / 'conv_f16x3.s'
    flags = build_native.FLAGS + build_native.EXTRA_FLAGS.get('conv_f16x3.hip', [])
    subprocess.check_call([HIPCC] + [f for f in flags if f != '-Wall'] + ['-S', '--cuda-device-only', '-o', str(asm),
                          os.path.join(ROOT, 'pnp_vcve_amd', 'csrc', 'conv_f16x3.hip')], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    kernels = [f for f in re.split(r'\n(?=_Z\w+:\s)', text) if re.match(r'_Z\w*conv3x3_f16x3_kernel\w*:', f)]
    assert len(kernels) >= 8                                   # {PAR, plain, MS} x {8x16, 4x16} x {trace}
    for fn in kernels:
        lines = fn.split('\n')
        sites = [i for i, l in enumerate(lines) if re.search(r's_waitcnt lgkmcnt\((4|6)\)', l) and 's_barrier' in lines[i + 1]]
        assert len(sites) >= 17, lines[0]                       # one per chunk but the last
        behind = []
        for i in sites:
            k = int(re.search(r'lgkmcnt\((\d)\)', lines[i]).group(1))
            n, j = 0, i - 1
            while j >= 0 and 'ds_write_b128' not in lines[j]:
                n += bool(re.search(r'\bds_(read|write)', lines[j]))
                j -= 1
            assert j >= 0 and n >= k, (lines[0], i, k, n)
            behind.append(n - k)
        assert min(behind) == 0, lines[0]

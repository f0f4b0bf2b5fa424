"""A gfx950 hazard the compiler does not pad, found in round 6 -- and what the build does about it.

    buffer_store_dwordx4 v[34:37], v55, s[24:27], s8 offen          a vector-memory store of more than 64 bits ...
    v_add_f32 v35, v1, v2                                           ... and the NEXT instruction is a vector-ALU write of one of its
                                                                        data registers

On MI355X the store then writes the NEW value in lanes 12-15 of every row of 16 (tools/repro/store_x4_then_wide_valu.hip,
profiles/r06_store_x4_hazard_probe.txt: 2 % of the stores behind a 32-bit VALU op, 23 % behind a packed one; buffer, global and
scratch stores alike; stores of 64 bits or less and LDS stores are not affected).  LLVM's hazard recogniser knows the rule -- 2 wait states on gfx940+ -- but exempts MUBUF stores whose soffset operand is
an SGPR (GCNHazardRecognizer::createsVALUHazard), the form the Winograd tile kernels' epilogue uses (tile offset in the scalar
operand, so that a lane's offset can be the out-of-range marker).  One wait state is enough for that form.  This was the root cause
of the "packed fp32" wrong values of round 5's Winograd kernel (profiles/r06_wino_pk_add_probe.txt): a packed op writes two data
registers at once and hits the window ten times as often, but a scalar build is exposed just the same whenever the scheduler puts the
producer of the next N tile's element right behind the store -- round 6's verified build had 12 such sites (all tests green, 420 soak
forwards bit-identical: with one wave per SIMD the timing happened to be benign).  NOT explained by it: the similar-looking failures
of dcn.hip's fp16 instantiation (r03) and of an intermediate conv_f16x3.hip (r04) with SLP-packed arithmetic -- their listings hold no
such site with or without the packing, so their -fno-slp-vectorize stays a fence.

A source-level fence (an s_nop statement, or a store form LLVM does pad) moves the register allocation of the 1200-MFMA tile loop off
its optimum (36 -> 400 B of scratch, 88 -> 73 frames/s), so the fix is applied where it costs nothing: build_native.py compiles
every translation unit's device code to a listing, `patch_listing` puts an `s_nop` behind each such store, and the patched listing
is assembled.  `lint_library` disassembles the code objects of the linked library and finds the pattern again (none must be left):
tests/test_isa_invariants.py runs it in the CPU suite, tools/lint_store_hazard.py from the command line.
"""
import os
import re
import subprocess
import tempfile

LLVM = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
STORE = re.compile(r'^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34]|'
                   r'buffer_store_format_xyzw?|tbuffer_store_format_xyzw?)\w*\s+(.*)$')
REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')
INSN = re.compile(r'^\s+([a-z_0-9]+)\s*(.*?)\s*(//.*|;.*)?$')
NOT_A_VGPR_WRITE = ('v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane', 'v_nop')


def _regs(operand):
    m = REG.search(operand)
    if not m:
        return None
    if m.group(1):
        return m.group(1), int(m.group(2)), int(m.group(3))
    return m.group(4), int(m.group(5)), int(m.group(5))


def _store_data(mn, ops):
    parts = [p.strip() for p in ops.split(',')]
    if mn.startswith(('global_store', 'flat_store', 'scratch_store')):
        return _regs(parts[1]) if len(parts) > 1 else None        # global_store vaddr, vdata, saddr
    return _regs(parts[0])                                        # buffer_store vdata, vaddr, srsrc, soffset


def wait_states_needed(mn, ops):
    """1 behind a MUBUF store with its soffset in an SGPR (measured), 2 behind every other form (LLVM's rule for gfx940+, and measured)"""
    if mn.startswith(('buffer_store', 'tbuffer_store')):
        parts = [p.strip() for p in ops.split(',')]
        if len(parts) > 3 and re.match(r'^(s\d+|ttmp\d+|m0)\b', parts[3]):
            return 1
    return 2


def lint_listing(lines, where=''):
    """-> [(kernel, line number of the store (1-based), store, offending instruction, where, wait states missing)]"""
    found, kernel, pending = [], '?', []
    for no, raw in enumerate(lines, 1):
        line = raw.rstrip('\n')
        lab = re.match(r'^(?:[0-9a-f]+ <)?([A-Za-z_.$][\w.$]*)>?:', line)
        if lab:
            if not lab.group(1).startswith(('.L', 'BB')):
                kernel = lab.group(1)
            continue                                               # (a label between the two does not change what executes next)
        m = INSN.match(line)
        if not m or m.group(1).startswith(('.', '/')):
            continue
        mn, ops = m.group(1), m.group(2)
        still = []
        for data, sline, sno, left in pending:                     # stores whose window is still open at this instruction
            if mn.startswith('v_') and not mn.startswith(NOT_A_VGPR_WRITE):
                dst = _regs(ops.split(',')[0])
                if dst and dst[0] == data[0] and dst[1] <= data[2] and dst[2] >= data[1]:
                    found.append((kernel, sno, sline.strip().split('//')[0].strip(), line.strip().split('//')[0].strip(), where, left))
                    continue
            left -= (int(ops.split()[0], 0) + 1) if mn == 's_nop' else 1
            if left > 0:
                still.append((data, sline, sno, left))
        pending = still
        s = STORE.match(line)
        if s:
            data = _store_data(s.group(1), s.group(2))
            if data and data[2] - data[1] >= 2:
                pending.append((data, line, no, wait_states_needed(s.group(1), s.group(2))))
    return found


def patch_listing(text):
    """-> (patched assembly text, number of stores padded): an `s_nop` right behind every store lint_listing reports"""
    lines = text.split('\n')
    sites = {}
    for _, sno, _, _, _, missing in lint_listing(lines):
        sites[sno] = max(sites.get(sno, 0), missing)
    for sno in sorted(sites, reverse=True):
        lines.insert(sno, '\ts_nop %d                                  ; (isa_hazards.py: the store above still reads the register written next)' % (sites[sno] - 1))
    out = '\n'.join(lines)
    left = lint_listing(out.split('\n'))
    assert not left, left[:3]
    return out, len(sites)


def device_listings(lib):
    """disassembly of every gfx950 code object bundled in a host library / object"""
    out = []
    with tempfile.TemporaryDirectory() as td:
        cp = os.path.join(td, 'lib.so')
        with open(lib, 'rb') as f, open(cp, 'wb') as g:
            g.write(f.read())
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', cp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
        for name in sorted(os.listdir(td)):
            if 'amdgcn' in name:
                dis = subprocess.run([f'{LLVM}/llvm-objdump', '-d', os.path.join(td, name)], check=True, capture_output=True, text=True).stdout
                out.append((name.split('lib.so.')[-1], dis.splitlines()))
    return out


def lint_library(path):
    """-> (sites, number of code objects) of a linked library / object, or of an assembly listing (*.s)"""
    if path.endswith('.s'):
        with open(path) as f:
            return lint_listing(f.readlines(), path), 1
    found, n = [], 0
    for name, lines in device_listings(path):
        found += lint_listing(lines, name)
        n += 1
    return found, n

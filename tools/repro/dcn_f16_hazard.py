#!/usr/bin/env python
"""Repro / root-cause harness for the fp16 DCN kernel hazard noted in csrc/dcn.hip (round 2: with the run-time two-armed
gather/MFMA order the fp16 instantiation gave corrupted, non-deterministic sums on gfx950 / ROCm 7.2).

Builds VARIANTS of csrc/dcn.hip (text substitutions of the one file, each compiled to its own tiny .so with an extern "C" entry),
runs each several times on the same input and reports (a) run-to-run determinism, (b) max |d| against the shipped fp16 kernel
and against the shipped fp32 kernel, and (c) the instruction neighbourhood of every v_mfma_f32_32x32x16_f16 of the variant
(from -save-temps) so that the good and the bad code shapes can be diffed.

    python tools/repro/dcn_f16_hazard.py --out gpurun_out/dcn_hazard          (on an MI355X)
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
CSRC = os.path.join(ROOT, 'pnp_vcve_amd', 'csrc')

ENTRY = r'''
extern "C" int repro_dcn_f16(const float* x, const float* om, const float* fx, const float* fy, const void* w16, const float* bias,
                             float* out, int h, int w, void* st) {
    DcnArgs d;
    d.dbg = nullptr; d.x = x; d.om = om; d.fx = fx; d.fy = fy; d.w = nullptr; d.w16 = w16; d.bias = bias; d.out = out; d.H = h; d.W = w;
    return launch_dcn(d, (hipStream_t)st);
}
'''
GOOD = 'if (F16 || grp == 0) {'
NOPS = 'asm volatile("s_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15" ::: "memory");'
MFMA_F16_HEAD = '        if (F16) {\n'
MFMA_F16_TAIL = '            return;\n        }\n        const f32x4* bb = sB + lane + grp * (8 * 64);'


def variants(src):
    assert src.count(GOOD) == 1 and src.count(MFMA_F16_HEAD) == 1 and src.count(MFMA_F16_TAIL) == 1
    bad = src.replace(GOOD, 'if (grp == 0) {')
    yield 'good_compile_time_order', src, []
    yield 'bad_run_time_order', bad, []
    yield 'bad_nops_after_mfma', bad.replace(MFMA_F16_TAIL, '            ' + NOPS + '\n' + MFMA_F16_TAIL), []
    yield 'bad_nops_before_mfma', bad.replace(MFMA_F16_HEAD, MFMA_F16_HEAD + '            ' + NOPS + '\n'), []
    yield 'bad_nops_both', bad.replace(MFMA_F16_TAIL, '            ' + NOPS + '\n' + MFMA_F16_TAIL).replace(
        MFMA_F16_HEAD, MFMA_F16_HEAD + '            ' + NOPS + '\n'), []
    # the same bad source with the machine scheduler's MFMA-aware pass / hazard padding knobs
    yield 'bad_mfma_padding_100', bad, ['-mllvm', '-amdgpu-mfma-padding-ratio=100']
    yield 'bad_O1', bad, ['-O1']
    yield 'bad_O2', bad, ['-O2']
    # which compiler stage?  (all on the bad source)
    yield 'bad_waitcnt_forcezero', bad, ['-mllvm', '-amdgpu-waitcnt-forcezero=1']          # s_waitcnt 0 behind every instruction
    yield 'bad_waitcnt_force_vm', bad, ['-mllvm', '-amdgpu-waitcnt-forcevm=1']
    yield 'bad_waitcnt_force_lgkm', bad, ['-mllvm', '-amdgpu-waitcnt-forcelgkm=1']
    yield 'bad_no_slp', bad, ['-fno-slp-vectorize']
    yield 'bad_no_postRA_sched', bad, ['-mllvm', '-enable-post-misched=0']
    yield 'bad_no_misched', bad, ['-mllvm', '-enable-misched=0']
    yield 'bad_no_machine_sink', bad, ['-mllvm', '-disable-machine-sink']
    yield 'bad_no_exec_mask_opt', bad, ['-mllvm', '-amdgpu-enable-pre-ra-optimizations=0']
    yield 'good_waitcnt_forcezero', src, ['-mllvm', '-amdgpu-waitcnt-forcezero=1']
    # candidate fix: no SLP vectorisation (no v_pk_*_f32 in the gather) -- under every timing perturbation that broke the others
    yield 'good_no_slp', src, ['-fno-slp-vectorize']
    yield 'good_no_slp_waitcnt_forcezero', src, ['-fno-slp-vectorize', '-mllvm', '-amdgpu-waitcnt-forcezero=1']
    yield 'bad_no_slp_waitcnt_forcezero', bad, ['-fno-slp-vectorize', '-mllvm', '-amdgpu-waitcnt-forcezero=1']
    yield 'bad_no_slp_nops_both', bad.replace(MFMA_F16_TAIL, '            ' + NOPS + '\n' + MFMA_F16_TAIL).replace(
        MFMA_F16_HEAD, MFMA_F16_HEAD + '            ' + NOPS + '\n'), ['-fno-slp-vectorize']
    # both arms gather first, but still selected at run time (two copies of the same arm): accumulator copies without reordering
    same = src.replace(GOOD, 'if (grp == 0) {').replace(
        '        } else {\n            mfma_tap(av);\n            stamp(d_m);\n            gather(ky, kx, oo, om, nav);\n            stamp(d_g);\n        }',
        '        } else {\n            gather(ky, kx, oo, om, nav);\n            stamp(d_g);\n            mfma_tap(av);\n            stamp(d_m);\n        }')
    yield 'run_time_branch_same_order_in_both_arms', same, []
    yield 'same_order_no_slp', same, ['-fno-slp-vectorize']
    # is the rare global-memory fallback of gather() (a divergent region under `if (__any(outside))`) involved?
    yield 'bad_without_global_fallback', bad.replace('if (__any(outside)) {', 'if (false && __any(outside)) {'), []
    yield 'bad_fallback_always_taken', bad.replace('if (__any(outside)) {', 'if (true || __any(outside)) {'), []
    # the accumulators pinned in one register range by an empty asm after each arm (no per-arm allocation, no copies)
    pin = 'asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));'
    yield 'bad_acc_pinned_after_arms', bad.replace('        if (++kx > 1) {', '        ' + pin + '\n        if (++kx > 1) {'), []
    yield 'bad_sched_barriers_around_mfma', bad.replace(MFMA_F16_HEAD, MFMA_F16_HEAD + '            __builtin_amdgcn_sched_barrier(0);\n').replace(
        MFMA_F16_TAIL, '            __builtin_amdgcn_sched_barrier(0);\n            ' + NOPS + '\n            __builtin_amdgcn_sched_barrier(0);\n' + MFMA_F16_TAIL), []


def mfma_neighbourhoods(asm_path):
    s = open(asm_path).read()
    name = '_ZN12_GLOBAL__N_117dcn_window_kernelILb1EEEv7DcnArgs'
    i = s.index(name + ':')
    body = s[i:s.index('s_endpgm', i)].splitlines()
    ins = [ln.strip() for ln in body if ln.strip() and not ln.strip().startswith((';', '.')) and not ln.strip().endswith(':')]
    out, last = [], -100
    for k, ln in enumerate(ins):
        if ln.startswith('v_mfma') and k - last > 14:
            out.append(f'--- instruction {k} of {len(ins)}')
            out.extend('    ' + x[:120] for x in ins[max(0, k - 4):k + 22])
            last = k
    nmov = sum(1 for x in ins if x.startswith(('v_mov_b64', 'v_mov_b32')) )
    nacc = sum(1 for x in ins if x.startswith('v_accvgpr'))
    return out, dict(instructions=len(ins), v_mov=nmov, v_accvgpr=nacc, mfma=sum(1 for x in ins if x.startswith('v_mfma')),
                     s_nop=sum(1 for x in ins if x.startswith('s_nop')))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'dcn_hazard'))
    ap.add_argument('--reps', type=int, default=12)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import torch
    from pnp_vcve_amd import _native, ops
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    src = open(os.path.join(CSRC, 'dcn.hip')).read()
    report = []
    for h, w in ((72, 80), (720, 1280)):
        x = torch.randn(h, w, 64, device=dev)
        off = torch.randn(288, h, w, device=dev) * 1.5
        mk = torch.randn(144, h, w, device=dev)
        blk = (torch.randint(-16, 17, (2, (h + 7) // 8, (w + 7) // 8), device=dev).float() / 4)
        flow = blk.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :w].contiguous()
        wt = torch.randn(64, 64, 3, 3, device=dev) * 0.05
        bias = torch.randn(64, device=dev) * 0.1
        ref32 = ops.modulated_deform_conv_nhwc(x, off, mk, wt, bias, flow=flow, fp16=False)
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
        torch.cuda.synchronize()
        report.append(f'=== frame {h}x{w}: shipped fp16 vs shipped fp32 kernel max|d| = {float((ref16 - ref32).abs().max()):.3e}')
        for name, text, flags in variants(src):
            d = tempfile.mkdtemp(prefix='dcnrep_')
            cpp = os.path.join(d, 'dcn_variant.hip')
            open(cpp, 'w').write(text + ENTRY)
            so = os.path.join(d, 'libdcnrep.so')
            cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I', CSRC, '-save-temps=obj',
                   cpp, '-o', so] + flags
            b = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
            if b.returncode != 0:
                report.append(f'{name:45s} BUILD FAILED: {b.stderr[-300:]}')
                continue
            lib = ctypes.CDLL(so)
            fn = lib.repro_dcn_f16
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            outs = []
            for _ in range(args.reps):
                o = torch.empty_like(x)
                rc = fn(x.data_ptr(), om.data_ptr(), flow[0].contiguous().data_ptr(), flow[1].contiguous().data_ptr(), w16.data_ptr(),
                        bias.data_ptr(), o.data_ptr(), h, w, None)
                assert rc == 0, rc
                torch.cuda.synchronize()
                outs.append(o)
            distinct = 1 + sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
            d16 = max(float((o - ref16).abs().max()) for o in outs)
            d32 = max(float((o - ref32).abs().max()) for o in outs)
            bad_elems = int(((outs[-1] - ref16).abs() > 1e-3).sum())
            asm = [f for f in os.listdir(d) if f.endswith('gfx950.s')]
            nb, stats = mfma_neighbourhoods(os.path.join(d, asm[0]))
            if (h, w) == (72, 80):
                with open(os.path.join(args.out, f'isa_{name}.txt'), 'w') as f:
                    f.write(f'# {name}: {stats}\n' + '\n'.join(nb) + '\n')
            if d16 > 0 and (h, w) == (72, 80):          # where are the wrong elements?
                bad = ((outs[-1] - ref16).abs() > 1e-3)
                ys, xs, cs = bad.nonzero(as_tuple=True)
                def hist(v, n):
                    return torch.bincount(v, minlength=n).tolist()
                report.append(f'    wrong elements by channel%32: {hist(cs % 32, 32)}')
                report.append(f'    by channel//32 (N tile): {hist(cs // 32, 2)}   by row%8 (2*wave + (m>>4)): {hist(ys % 8, 8)}   by col%16: {hist(xs % 16, 16)}')
                report.append(f'    distinct 8x16 tiles touched: {len(set(((ys // 8) * 100 + xs // 16).tolist()))} of {((h + 7) // 8) * ((w + 15) // 16)}')
                # a wrong A operand (a sample) spoils all 64 channels of its pixel (both N tiles share it); a wrong accumulator
                # half spoils the 32 channels of ONE N tile
                n0_, n1_ = bad[..., :32].sum(-1), bad[..., 32:].sum(-1)
                pix = (n0_ + n1_) > 0
                report.append(f'    wrong pixels {int(pix.sum())}: both N tiles wrong {int(((n0_ > 0) & (n1_ > 0)).sum())}, only channels 0..31 {int(((n0_ > 0) & (n1_ == 0)).sum())}, '
                              f'only channels 32..63 {int(((n0_ == 0) & (n1_ > 0)).sum())}; wrong channels per wrong pixel: mean {float((n0_ + n1_)[pix].float().mean()):.1f}')
                # are whole 16-pixel rows of a wave wrong together?
                rows = bad.any(-1)
                rr = rows.view(h // 8 if h % 8 == 0 else -1, 8, -1) if h % 8 == 0 else None
                if rr is not None and w % 16 == 0:
                    seg = rows.view(h, w // 16, 16).sum(-1)
                    report.append(f'    wrong pixels per (image row, 16-pixel segment) where any is wrong: histogram {torch.bincount(seg[seg > 0].flatten(), minlength=17).tolist()}')
            if d16 > 0 and (h, w) == (72, 80) and name == 'bad_run_time_order':
                # what do the wrong values look like?  candidates: the sum over taps >= k0 only (the accumulator lost what it held
                # before tap k0: SrcC read wrong) or over taps < k0 only (updates from tap k0 on were lost)
                o = outs[-1]
                badm = (o - ref16).abs() > 1e-3
                for k0 in range(1, 9):
                    wz = wt.clone()
                    wz.view(64, 64, 9)[:, :, :k0] = 0                      # taps 0 .. k0-1 removed
                    tail = ops.modulated_deform_conv_nhwc(x, off, mk, wz, bias, flow=flow, fp16=True)       # bias + taps >= k0
                    head = ref16 - tail + bias.view(1, 1, 64)                                             # bias + taps < k0
                    m_tail = int(((o - tail).abs() < 2e-3)[badm].sum())
                    m_head = int(((o - head).abs() < 2e-3)[badm].sum())
                    report.append(f'    wrong elements that equal bias + taps >= {k0} only: {m_tail:6d}   bias + taps < {k0} only: {m_head:6d}   (of {int(badm.sum())})')
            verdict = 'OK ' if distinct == 1 and d16 == 0.0 else ('DETERMINISTIC but differs' if distinct == 1 else 'NON-DETERMINISTIC')
            report.append(f'{name:45s} {verdict:28s} runs differing from run 0: {distinct - 1:2d}/{args.reps - 1}  max|d| vs shipped fp16 {d16:.3e}  '
                          f'vs fp32 {d32:.3e}  elements off by > 1e-3: {bad_elems}  {stats}')
    txt = '\n'.join(report)
    print(txt)
    open(os.path.join(args.out, 'report.txt'), 'w').write(txt + '\n')


if __name__ == '__main__':
    main()

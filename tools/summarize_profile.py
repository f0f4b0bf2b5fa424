"""Condense rocprofv3 csv output (kernel trace + PMC passes) into a per-kernel summary."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ('conv3x3_wino_quad_ms_kernel', 'conv3x3_wino_quad_kernel', 'conv3x3_wino_gated_kernel', 'conv3x3_wino_kernel', 'conv3x3_persist_kernel', 'conv3x3_mfma_kernel', 'conv3x3_f16x3_kernel', 'conv3x3_f16_small_kernel', 'conv3x3_f16_multi_kernel', 'conv3x3_f16_kernel', 'par_tile_flags_kernel', 'conv_last_valu_kernel', 'dcn_window_kernel',
              'mv_warp_nhwc64_kernel', 'mv_warp_nhwc_kernel', 'psnr_sse_kernel', 'flow_warp_nchw_kernel', 'pack_weights_kernel',
              'pack_lr_kernel', 'caa_predict_kernel', 'mix_bias_kernel'):
        if k in name:
            if k == 'conv3x3_mfma_kernel':
                import re
                mm = re.search(r'conv3x3_mfma_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', name)
                if not mm:
                    mm = re.search(r'conv3x3_mfma_kernel<(\d+), (\d+), (\d+), (\d+)>', name)
                return k + ('<%s,%s,%s,%s>' % mm.groups() if mm else '')
            if k in ('conv3x3_wino_quad_kernel', 'conv3x3_wino_gated_kernel', 'conv3x3_wino_kernel', 'conv3x3_f16x3_kernel', 'conv3x3_f16_kernel', 'conv3x3_f16_small_kernel', 'conv3x3_f16_multi_kernel', 'conv3x3_persist_kernel', 'dcn_window_kernel', 'mv_warp_nhwc_kernel', 'mv_warp_nhwc64_kernel'):     # keep the template arguments
                import re
                mm = re.search(k + r'<([^>]*)>', name)
                if mm:
                    return k + '<' + mm.group(1).replace(' ', '') + '>'
                mm = re.search(k + r'I((?:Lb[01]E)+)(?:Li\d+E)*E', name)                          # mangled bools
                if mm:
                    return k + '<' + ','.join('true' if b == '1' else 'false' for b in re.findall(r'Lb([01])E', mm.group(1))) + '>'
            return k
    return name[:60]


def main(root):
    tr = glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True)
    if tr:
        agg = defaultdict(lambda: [0, 0.0])
        for f in tr:
            for r in csv.DictReader(open(f)):
                d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
                a = agg[short(r['Kernel_Name'])]
                a[0] += 1
                a[1] += d
        tot = sum(v[1] for v in agg.values())
        print('== kernel trace (all dispatches of the profiled command)')
        print(f'{"kernel":70s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"%":>6s}')
        for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f'{k:70s} {n:7d} {us:12.1f} {us / n:10.2f} {100 * us / tot:6.2f}')
    for pas in ('pmc_sq', 'pmc_fetch', 'pmc_write', 'pmc_lds'):
        fs = glob.glob(os.path.join(root, pas, '**', '*counter_collection.csv'), recursive=True)
        if not fs:
            continue
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for f in fs:
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                agg[k][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[(k, r['Counter_Name'])] += 1
        print(f'== {pas}: per-kernel counter AVERAGE per dispatch')
        for k in sorted(agg):
            parts = [f'{c}={v / cnt[(k, c)]:.4g}' for c, v in sorted(agg[k].items())]
            n = max(cnt[(k, c)] for c in agg[k])
            print(f'{k:70s} n={n:<6d} ' + ' '.join(parts))


def pmc_json(root, out):
    """per-kernel HBM traffic per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE reads exactly half
    of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section); separate passes per counter."""
    import json
    res = {}
    for pas, cname in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        for f in glob.glob(os.path.join(root, pas, '**', '*counter_collection.csv'), recursive=True):
            acc = defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] != cname:
                    continue
                a = acc[short(r['Kernel_Name'])]
                a[0] += float(r['Counter_Value'])
                a[1] += 1
            for k, (v, n) in acc.items():
                res.setdefault(k, {})[cname + '_KiB_per_launch'] = v / n
                res[k]['launches'] = n
    for k, d in res.items():
        if 'FETCH_SIZE_KiB_per_launch' in d and 'WRITE_SIZE_KiB_per_launch' in d:
            d['hbm_bytes_per_launch'] = (2 * d['FETCH_SIZE_KiB_per_launch'] + d['WRITE_SIZE_KiB_per_launch']) * 1024
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv[1])
    if len(sys.argv) > 2:
        pmc_json(sys.argv[1], sys.argv[2])

"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

Generates tests/golden/*.npz by running the *imported reference* (through
oracle/ref_shim.py) on seeded synthetic inputs/weights defined in
tests/golden_util.py.  Only outputs are stored; inputs are regenerated from
seeds on whichever machine runs the tests.

    python oracle/gen_golden.py                    # (re)write every fixture + manifest
    python oracle/gen_golden.py --only a,b         # only the named generator / flow_warp / metrics cases, merged into the manifest
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np          # noqa: E402
import torch                # noqa: E402

import golden_util as gu    # noqa: E402
from oracle import cpu_ref, ref_shim   # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    assert ref_shim.available(), 'reference tree not found'
    torch.set_num_threads(8)
    os.makedirs(gu.GOLDEN_DIR, exist_ok=True)
    manifest = dict(reference='ZeldaM1/PnP-VCVE @ /root/reference (imported on CPU via oracle/ref_shim.py)',
                    torch=torch.__version__, cases={})
    only = None
    if '--only' in sys.argv:
        only = set(sys.argv[sys.argv.index('--only') + 1].split(','))
        with open(os.path.join(gu.GOLDEN_DIR, 'manifest.json')) as f:
            manifest = json.load(f)
    Ref = ref_shim.reference_generator_class()
    sbu, da, bvn = ref_shim.reference_modules()
    ref_flow_warp = ref_shim.reference_flow_warp()

    # ---- full generator -------------------------------------------------
    for case in gu.GEN_CASES:
        if only is not None and case['name'] not in only:
            continue
        cfg, sd_np, clip = gu.gen_case_inputs(case)
        m = Ref(**cfg).eval()
        sd = cpu_ref.to_torch_state(sd_np)
        m.load_state_dict(sd, strict=True)
        a = {k: T(v) for k, v in clip.items()}
        with torch.no_grad():
            ref = m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])
            mine = cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'],
                                             a['partitions'])
            # how much each ingredient matters in this case (so a parity gate of 1e-3 can see it)
            sens = {}
            sens['par->0'] = float((cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'],
                                                              a['base_QPs'], a['partitions'] * 0) - ref).abs().max())
            sens['mvs->0'] = float((cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'] * 0,
                                                              a['base_QPs'], a['partitions']) - ref).abs().max())
            sens['QPs*2'] = float((cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'] * 2, a['slices'], a['mvs'],
                                                             a['base_QPs'], a['partitions']) - ref).abs().max())
            sens['base*2'] = float((cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'],
                                                              a['base_QPs'] * 2, a['partitions']) - ref).abs().max())
            sens['allkey'] = float((cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'] * 0 + 80,
                                                              a['mvs'], a['base_QPs'], a['partitions']) - ref).abs().max())
        d = float((ref - mine).abs().max())
        np.savez(os.path.join(gu.GOLDEN_DIR, case['name'] + '.npz'), out=ref.numpy())
        manifest['cases'][case['name']] = dict(kind='generator', shape=list(ref.shape),
                                               oracle_vs_reference_maxabs=d, sensitivity=sens,
                                               out_minus_lq_maxabs=float((ref - (a['lq'] if not cfg['vsr'] else 0)).abs().max()))
        print(case['name'], list(ref.shape), 'oracle-vs-ref', d, sens, flush=True)
        assert d < 1e-5

    # ---- flow_warp --------------------------------------------------------
    for case in gu.WARP_CASES:
        if only is not None and case['name'] not in only:
            continue
        x, flow = gu.warp_case_inputs(case)
        mode = case.get('mode', 'bilinear')
        with torch.no_grad():
            ref = ref_flow_warp(T(x), T(flow), interpolation=mode)
            mine = cpu_ref.flow_warp(T(x), T(flow), mode)
        d = float((ref - mine).abs().max())
        np.savez(os.path.join(gu.GOLDEN_DIR, case['name'] + '.npz'), out=ref.numpy())
        manifest['cases'][case['name']] = dict(kind='flow_warp', shape=list(ref.shape), oracle_vs_reference_maxabs=d)
        print(case['name'], d, flush=True)
        assert d < 1e-5

    # ---- PSNR / tensor2img / BasicVSR.evaluate (row 8f-2): the reference's own functions -------------
    ref_psnr, ref_tensor2img, RefBasicVSR = ref_shim.reference_metrics()

    class _Cfg(dict):                      # mmcv.Config stand-in: attribute access + .get (names only)
        __getattr__ = dict.__getitem__

    for case in gu.METRIC_CASES:
        if only is not None and case['name'] not in only:
            continue
        out_np, gt_np = gu.metric_case_inputs(case)
        out_t, gt_t = T(out_np), T(gt_np)
        nt = out_t.shape[1]
        img_out = np.stack([ref_tensor2img(out_t[:, i]) for i in range(nt)])          # (T, h, w, 3) uint8 BGR
        img_gt = np.stack([ref_tensor2img(gt_t[:, i]) for i in range(nt)])
        store = dict(img_out=img_out, img_gt=img_gt)
        worst = 0.0
        for crop in (0, 3):
            vals = np.array([ref_psnr(img_out[i], img_gt[i], crop) for i in range(nt)], np.float64)
            store[f'psnr_crop{crop}'] = vals
            model = RefBasicVSR.__new__(RefBasicVSR)                                   # evaluate() reads test_cfg only
            model.test_cfg = _Cfg(metrics=['PSNR'], crop_border=crop)
            finite = [i for i in range(nt) if np.isfinite(vals[i])]
            store[f'evaluate_all_crop{crop}'] = np.float64(RefBasicVSR.evaluate(model, out_t, gt_t)['PSNR'])
            store[f'evaluate_finite_crop{crop}'] = np.float64(
                RefBasicVSR.evaluate(model, out_t[:, finite], gt_t[:, finite])['PSNR'])
            mine = [cpu_ref.psnr_uint8(cpu_ref.tensor2img_uint8(out_t[0, i]), cpu_ref.tensor2img_uint8(gt_t[0, i]), crop)
                    for i in range(nt)]
            for i in range(nt):
                assert np.array_equal(cpu_ref.tensor2img_uint8(out_t[0, i]), img_out[i])
                if np.isfinite(vals[i]):
                    worst = max(worst, abs(mine[i] - vals[i]))
                else:
                    assert mine[i] == vals[i]
            store[f'finite_frames_crop{crop}'] = np.array(finite, np.int64)
        np.savez(os.path.join(gu.GOLDEN_DIR, case['name'] + '.npz'), **store)
        manifest['cases'][case['name']] = dict(kind='metrics', oracle_vs_reference_maxabs=worst,
                                               psnr_crop0=[float(v) for v in store['psnr_crop0']])
        print(case['name'], 'oracle-vs-ref dB', worst, store['psnr_crop0'], store['evaluate_finite_crop0'], flush=True)
        assert worst < 1e-5

    if only is not None:
        with open(os.path.join(gu.GOLDEN_DIR, 'manifest.json'), 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        print('updated', sorted(only))
        return

    # ---- CAA predictors ---------------------------------------------------
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=41)
    sd = cpu_ref.to_torch_state(sd_np)
    q = T(np.array(gu.CAA_QPS, np.float32).reshape(1, -1, 1, 1, 1))
    bp = da.Base_Predictor(nf=64, num_experts=6, softmax=True).eval()
    bp.load_state_dict({k[len('BasePredictor.'):]: v for k, v in sd.items() if k.startswith('BasePredictor.')})
    se = da.SEModule(64).eval()
    se.load_state_dict({k[len('BiasePredictor.'):]: v for k, v in sd.items() if k.startswith('BiasePredictor.')})
    with torch.no_grad():
        ew_ref = bp(q)
        g_ref, _ = se(q)
        ew = cpu_ref.base_predictor(sd, q, True)
        g, _ = cpu_ref.bias_predictor(sd, cfg, q)
    d = max(float((ew_ref - ew).abs().max()), float((g_ref - g).abs().max()))
    np.savez(os.path.join(gu.GOLDEN_DIR, 'caa_predictors.npz'), ew=ew_ref.numpy(), gamma=g_ref.numpy())
    manifest['cases']['caa_predictors'] = dict(kind='caa', oracle_vs_reference_maxabs=d, wseed=41)
    print('caa', d, flush=True)
    assert d < 1e-6

    # ---- one BAE block / one branch ----------------------------------------
    for case in gu.BLOCK_CASES:
        cfg, sd_np, x, par, ew, gamma = gu.block_case_inputs(case)
        sd = cpu_ref.to_torch_state(sd_np)
        pre = 'backward_resblocks.main.0.'
        blk = sbu.ResidualBlockNoBNDynamic_drt(mid_channels=64, num_experts=6, with_se=True, init_weight=True,
                                               one_layer=True, channel_first=True, sparse_val=False).eval()
        blk.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
        h, w = x.shape[-2:]
        with torch.no_grad():
            inp = {'x': T(x), 'par': T(par).view(1, 3, 1, h, w), 'weights': T(ew), 'gamma': T(gamma), 'beta': None}
            ref = blk(inp)['x']
            mine = cpu_ref.bae_block(sd, cfg, pre, T(x), T(par).view(1, 3, 1, h, w), T(ew), T(gamma))
        d = float((ref - mine).abs().max())
        out = dict(block=ref.numpy())
        # whole forward branch (input conv + 8 blocks) on a 195-channel input
        br = bvn.ResidualBlocksWithInputConvDynamic_drt(195, 64, 8, 6, True, with_se=True, init_weight=True,
                                                        num_group=1, one_layer=True, blocktype='drt',
                                                        channel_first=True, sparse_val=False).eval()
        pb = 'forward_resblocks.'
        br.load_state_dict({k[len(pb):]: v for k, v in sd.items() if k.startswith(pb)})
        xin = gu.syn.uniform(case['seed'], 'xin', (1, 195, h, w), -1.0, 1.0)
        with torch.no_grad():
            inp = {'x': T(xin), 'par': T(par), 'weights': T(ew), 'gamma': T(gamma), 'beta': None}
            ref_b = br(inp)['x']
            mine_b = cpu_ref.resblocks(sd, cfg, 'forward_resblocks', T(xin), T(par), T(ew), T(gamma))
        d2 = float((ref_b - mine_b).abs().max())
        out['branch'] = ref_b.numpy()
        np.savez(os.path.join(gu.GOLDEN_DIR, case['name'] + '.npz'), **out)
        manifest['cases'][case['name']] = dict(kind='block+branch', oracle_vs_reference_maxabs=max(d, d2))
        print(case['name'], d, d2, flush=True)
        assert max(d, d2) < 1e-5

    # ---- side-info rasteriser (row 8f-1): the reference loader run on a synthetic on-disk clip
    import tempfile
    from PIL import Image
    Loader = ref_shim.reference_loader_class()
    for case in gu.RASTER_CASES:
        rec, rec_frame, slices, h, w = gu.raster_case_inputs(case)
        with tempfile.TemporaryDirectory() as root:
            png = os.path.join(root, 'crf25', 'png', '000')
            mvd = os.path.join(root, 'crf25', 'mv', '000')
            os.makedirs(png)
            os.makedirs(mvd)
            table = {'crf25': {'000': {}}}
            paths = []
            for f, sl in enumerate(slices):
                Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(os.path.join(png, f'{f:08d}.png'))
                np.save(os.path.join(mvd, f'{f:08d}.npy'), rec[rec_frame == f].reshape(-1, 10))
                table['crf25']['000'][str(f)] = {'slice': sl, 'QP': 20 + f}
                paths.append(os.path.join(png, f'{f:08d}.png'))
            qp_file = os.path.join(root, 'qp.json')
            with open(qp_file, 'w') as fq:
                json.dump(table, fq)
            ld = Loader(io_backend='disk', key='lq', channel_order='rgb', random_compress=False, load_mv=True,
                        load_qp_slice=True, load_base_qp=True, load_partition=True, drconv=True, qp_slice_file=qp_file)
            res = ld(dict(lq_path=paths))
        mvs_ref = np.stack([m.transpose(2, 0, 1) for m in res['mvs']]).astype(np.float32)
        par_ref = np.stack([(p_.astype(np.float32) / 255.).transpose(2, 0, 1) for p_ in res['partitions']])
        mine_mv, mine_par = cpu_ref.rasterise_side_info(rec, rec_frame, slices, h, w)
        d = max(float(np.abs(mvs_ref - mine_mv).max()), float(np.abs(par_ref - mine_par).max()))
        np.savez(os.path.join(gu.GOLDEN_DIR, case['name'] + '.npz'), mvs=mvs_ref, partitions=par_ref,
                 slices=np.array([float(np.asarray(v).reshape(-1)[0]) for v in res['slices']], np.float32),
                 QPs=np.array([float(np.asarray(v).reshape(-1)[0]) for v in res['QPs']], np.float32),
                 base_QPs=np.array([float(np.asarray(v).reshape(-1)[0]) for v in res['base_QPs']], np.float32))
        manifest['cases'][case['name']] = dict(kind='rasteriser', oracle_vs_reference_maxabs=d,
                                               nonzero_mv_fraction=float((mvs_ref != 0).mean()))
        print(case['name'], d, 'nonzero', float((mvs_ref != 0).mean()), flush=True)
        assert d == 0.0

    with open(os.path.join(gu.GOLDEN_DIR, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print('wrote', gu.GOLDEN_DIR)


if __name__ == '__main__':
    main()

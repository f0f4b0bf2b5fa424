"""CPU: the oracle (oracle/cpu_ref.py) against the golden vectors produced by the
imported reference (oracle/gen_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import cpu_ref

TOL = 2e-6   # fp32, same ATen kernels; observed 1.2e-7


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_manifest_lists_every_case():
    with open(os.path.join(gu.GOLDEN_DIR, 'manifest.json')) as f:
        man = json.load(f)
    names = [c['name'] for c in gu.GEN_CASES + gu.WARP_CASES + gu.BLOCK_CASES + gu.METRIC_CASES] + ['caa_predictors']
    for n in names:
        assert n in man['cases'], n
        assert os.path.exists(os.path.join(gu.GOLDEN_DIR, n + '.npz')), n
        assert man['cases'][n]['oracle_vs_reference_maxabs'] < 1e-5


@pytest.mark.parametrize('case', gu.GEN_CASES, ids=[c['name'] for c in gu.GEN_CASES])
def test_generator_matches_reference(case):
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    sd = cpu_ref.to_torch_state(sd_np)
    a = {k: T(v) for k, v in clip.items()}
    with torch.no_grad():
        out = cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'],
                                        a['partitions'])
    ref = gu.load_golden(case['name'])['out']
    assert out.shape == ref.shape
    assert np.abs(out.numpy() - ref).max() < TOL


@pytest.mark.parametrize('case', gu.WARP_CASES, ids=[c['name'] for c in gu.WARP_CASES])
def test_flow_warp_matches_reference(case):
    x, flow = gu.warp_case_inputs(case)
    out = cpu_ref.flow_warp(T(x), T(flow), case.get('mode', 'bilinear')).numpy()
    ref = gu.load_golden(case['name'])['out']
    assert np.abs(out - ref).max() < (TOL if case.get('mode', 'bilinear') == 'bilinear' else 1e-30)      # nearest: a copy, bit-exact


def test_flow_warp_size_mismatch_raises():
    with pytest.raises(ValueError):
        cpu_ref.flow_warp(torch.zeros(1, 4, 8, 8), torch.zeros(1, 8, 9, 2))


def test_caa_predictors_match_reference():
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd = cpu_ref.to_torch_state(gu.syn.make_state_dict(cfg, seed=41))
    q = T(np.array(gu.CAA_QPS, np.float32).reshape(1, -1, 1, 1, 1))
    g = gu.load_golden('caa_predictors')
    assert np.abs(cpu_ref.base_predictor(sd, q, True).numpy() - g['ew']).max() < 1e-6
    assert np.abs(cpu_ref.bias_predictor(sd, cfg, q)[0].numpy() - g['gamma']).max() < 1e-6


@pytest.mark.parametrize('case', gu.BLOCK_CASES, ids=[c['name'] for c in gu.BLOCK_CASES])
def test_block_and_branch_match_reference(case):
    cfg, sd_np, x, par, ew, gamma = gu.block_case_inputs(case)
    sd = cpu_ref.to_torch_state(sd_np)
    h, w = x.shape[-2:]
    g = gu.load_golden(case['name'])
    with torch.no_grad():
        blk = cpu_ref.bae_block(sd, cfg, 'backward_resblocks.main.0.', T(x), T(par).view(1, 3, 1, h, w), T(ew),
                                T(gamma))
        xin = gu.syn.uniform(case['seed'], 'xin', (1, 195, h, w), -1.0, 1.0)
        br = cpu_ref.resblocks(sd, cfg, 'forward_resblocks', T(xin), T(par), T(ew), T(gamma))
    assert np.abs(blk.numpy() - g['block']).max() < TOL
    assert np.abs(br.numpy() - g['branch']).max() < 1e-5


def test_preconditions_match_reference_errors():
    cfg, sd_np, clip = gu.gen_case_inputs(gu.GEN_CASES[0])
    sd = cpu_ref.to_torch_state(sd_np)
    a = {k: T(v) for k, v in clip.items()}
    with pytest.raises(AssertionError):      # iconvsr_ipb_par.py:51
        cpu_ref.generator_forward(sd, cfg, a['lq'][..., :32, :32], a['QPs'], a['slices'], a['mvs'][..., :32, :32],
                                  a['base_QPs'], a['partitions'][..., :32, :32])
    with pytest.raises(ValueError):          # flow_warp.py:27-29 after spatial_padding of lrs only
        cpu_ref.generator_forward(sd, cfg, a['lq'][..., :, :66], a['QPs'], a['slices'], a['mvs'][..., :, :66],
                                  a['base_QPs'], a['partitions'][..., :, :66])


@pytest.mark.parametrize('case', gu.METRIC_CASES, ids=[c['name'] for c in gu.METRIC_CASES])
def test_psnr_and_tensor2img_match_reference(case):
    """oracle vs the reference's OWN tensor2img / psnr / BasicVSR.evaluate outputs (core/misc.py:51-71,
    core/evaluation/metrics.py:170-215, restorers/basicvsr.py:119-153): uint8 images bit-exact (clamp, .5 ties,
    BGR order), PSNR to float32 resolution, inf for an identical pair, the clip mean."""
    out, gt = gu.metric_case_inputs(case)
    g = gu.load_golden(case['name'])
    nt = out.shape[1]
    for i in range(nt):
        assert np.array_equal(cpu_ref.tensor2img_uint8(T(out[0, i])), g['img_out'][i])
        assert np.array_equal(cpu_ref.tensor2img_uint8(T(gt[0, i])), g['img_gt'][i])
    assert (g['img_out'][2].astype(int) - g['img_gt'][2].astype(int)).any()          # the tie frame is not trivially equal
    for crop in (0, 3):
        ref = g[f'psnr_crop{crop}']
        assert np.isinf(ref[3]) and np.isfinite(np.delete(ref, 3)).all()
        for i in range(nt):
            v = cpu_ref.psnr_uint8(g['img_out'][i], g['img_gt'][i], crop)
            assert v == ref[i] if np.isinf(ref[i]) else abs(v - ref[i]) < 1e-5, (crop, i, v, ref[i])
        fin = list(g[f'finite_frames_crop{crop}'])
        assert abs(cpu_ref.clip_psnr(T(out[:, fin]), T(gt[:, fin]), crop) - float(g[f'evaluate_finite_crop{crop}'])) < 1e-4
        assert cpu_ref.clip_psnr(T(out), T(gt), crop) == float('inf') == float(g[f'evaluate_all_crop{crop}'])


@pytest.mark.parametrize('case', gu.METRIC_CASES, ids=[c['name'] for c in gu.METRIC_CASES])
def test_host_metrics_and_evaluate_match_reference(case):
    """the package's host-side tensor2img / psnr and BasicVSR.evaluate (CPU tensors take the numpy path) against the same fixture"""
    from pnp_vcve_amd import metrics
    from pnp_vcve_amd.restorer import BasicVSR
    out, gt = gu.metric_case_inputs(case)
    g = gu.load_golden(case['name'])
    for i in range(out.shape[1]):
        assert np.array_equal(metrics.tensor2img(T(out[:, i])), g['img_out'][i])
        for crop in (0, 3):
            v, ref = metrics.psnr(g['img_out'][i], g['img_gt'][i], crop), g[f'psnr_crop{crop}'][i]
            assert v == ref if np.isinf(ref) else abs(v - ref) < 1e-5
    for crop in (0, 3):
        m = BasicVSR.__new__(BasicVSR)
        m.test_cfg = dict(metrics=['PSNR'], crop_border=crop)
        m.allowed_metrics = metrics.ALLOWED_METRICS
        fin = list(g[f'finite_frames_crop{crop}'])
        got = BasicVSR.evaluate(m, T(out[:, fin]), T(gt[:, fin]))['PSNR']
        assert abs(float(got) - float(g[f'evaluate_finite_crop{crop}'])) < 1e-4


def test_psnr_definition():
    a = torch.rand(1, 2, 3, 16, 16)
    b = (a + 0.05).clamp(0, 1)
    v = cpu_ref.clip_psnr(a, b)
    assert 20 < v < 40
    assert cpu_ref.clip_psnr(a, a) == float('inf')


@pytest.mark.parametrize('case', gu.RASTER_CASES, ids=[c['name'] for c in gu.RASTER_CASES])
def test_rasteriser_matches_reference_loader(case):
    rec, rec_frame, slices, h, w = gu.raster_case_inputs(case)
    mvs, par = cpu_ref.rasterise_side_info(rec, rec_frame, slices, h, w)
    g = gu.load_golden(case['name'])
    assert np.array_equal(mvs, g['mvs']) and np.array_equal(par, g['partitions'])
    assert [chr(int(v)) for v in g['slices']] == list(slices)

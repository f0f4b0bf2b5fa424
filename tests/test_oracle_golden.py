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
    names = [c['name'] for c in gu.GEN_CASES + gu.WARP_CASES + gu.BLOCK_CASES] + ['caa_predictors']
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

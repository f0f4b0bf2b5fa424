"""GPU parity tests of the single ops, through the C ABI (pnp_vcve_amd.ops -> libpnpvcve_hip.so),
against the golden vectors of the imported reference and against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
from oracle import cpu_ref

pytestmark = pytest.mark.gpu

TOL_WARP = 1e-5      # fp32 bilinear gather; SURVEY.md section 7 step 3 gate
TOL_CONV = 2e-5      # exact-fp32 MFMA, K <= 1764, |activations| ~ 1
TOL_BLOCK = 5e-5


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def test_native_library_is_loaded():
    from pnp_vcve_amd import _native
    assert _native.lib().pnp_abi_version() == 5


@pytest.mark.parametrize('case', gu.WARP_CASES, ids=[c['name'] for c in gu.WARP_CASES])
def test_flow_warp_nchw_vs_reference(case):
    from pnp_vcve_amd import ops
    x, flow = gu.warp_case_inputs(case)
    mode = case.get('mode', 'bilinear')
    out = ops.flow_warp(G(x), G(flow), interpolation=mode)
    ref = gu.load_golden(case['name'])['out']
    assert maxdiff(out, ref) < (TOL_WARP if mode == 'bilinear' else 1e-30)       # nearest copies pixels: bit-exact, ties included


@pytest.mark.parametrize('case', gu.WARP_CASES, ids=[c['name'] for c in gu.WARP_CASES])
def test_mv_warp_nhwc_vs_reference(case):
    from pnp_vcve_amd import ops
    x, flow = gu.warp_case_inputs(case)
    ref = gu.load_golden(case['name'])['out']
    for n in range(x.shape[0]):
        feat = ops.nchw_to_nhwc(G(x[n:n + 1]))[0]
        mode = case.get('mode', 'bilinear')
        out = ops.mv_warp_nhwc(feat, G(flow[n, :, :, 0]), G(flow[n, :, :, 1]), interpolation=mode)
        back = ops.nhwc_to_nchw(out.unsqueeze(0))
        assert maxdiff(back, ref[n:n + 1]) < (TOL_WARP if mode == 'bilinear' else 1e-30)


def test_flow_warp_errors_like_reference():
    from pnp_vcve_amd import ops
    with pytest.raises(ValueError):
        ops.flow_warp(torch.zeros(1, 4, 8, 8, device=dev()), torch.zeros(1, 8, 9, 2, device=dev()))
    with pytest.raises(RuntimeError):
        ops.flow_warp(torch.zeros(1, 4, 8, 8), torch.zeros(1, 8, 8, 2))     # CPU tensors: no fallback


def test_layout_round_trip():
    from pnp_vcve_amd import ops
    x = torch.randn(2, 12, 9, 20, device=dev())
    y = ops.nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(y), x)


def test_caa_predictors_vs_reference():
    from pnp_vcve_amd import ops
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd = gu.syn.make_state_dict(cfg, seed=41)
    g = gu.load_golden('caa_predictors')
    ew, gamma = ops.caa_predict(gu.CAA_QPS, gu.CAA_QPS,
                                G(sd['BasePredictor.BaseNet.0.weight']), G(sd['BasePredictor.BaseNet.0.bias']),
                                G(sd['BasePredictor.BaseNet.2.weight']), G(sd['BasePredictor.BaseNet.2.bias']),
                                G(sd['BiasePredictor.fc.0.weight']), G(sd['BiasePredictor.fc.2.weight']), softmax=True)
    assert maxdiff(ew, g['ew'][0]) < 1e-6
    assert maxdiff(gamma, g['gamma'][0]) < 1e-6


@pytest.mark.parametrize('hw', [(16, 16), (24, 40), (37, 53), (64, 64), (128, 256)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_conv3x3_single_source(hw, act):
    from pnp_vcve_amd import ops
    h, w = hw
    x = gu.syn.uniform(7, f'x{h}x{w}', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(7, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(7, 'b', (64,), -0.1, 0.1)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), padding=1)
    ref = [ref, F.relu(ref), F.leaky_relu(ref, 0.1)][act]
    xs = ops.nchw_to_nhwc(G(x))[0]
    out = ops.conv3x3([xs], [ops.pack_conv3x3(G(wt))], bias=G(b), act=act)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_CONV


def test_conv3x3_identity_weights_localise_layout_bugs():
    """centre-tap identity: out == in exactly; then a one-tap shift."""
    from pnp_vcve_amd import ops
    h, w = 24, 40
    x = gu.syn.uniform(8, 'x', (1, 64, h, w), -1, 1)
    for (ky, kx) in [(1, 1), (0, 0), (2, 1), (1, 2)]:
        wt = np.zeros((64, 64, 3, 3), np.float32)
        wt[np.arange(64), np.arange(64), ky, kx] = 1.0
        ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(wt), padding=1)
        out = ops.conv3x3([ops.nchw_to_nhwc(G(x))[0]], [ops.pack_conv3x3(G(wt))])
        assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) == 0.0, (ky, kx)
    # asymmetric channel permutation (catches transposed B images)
    perm = np.roll(np.arange(64), 5)
    wt = np.zeros((64, 64, 3, 3), np.float32)
    wt[np.arange(64), perm, 1, 1] = 1.0
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(wt), padding=1)
    out = ops.conv3x3([ops.nchw_to_nhwc(G(x))[0]], [ops.pack_conv3x3(G(wt))])
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) == 0.0


@pytest.mark.parametrize('nwide', [0, 1, 2, 3])
def test_conv3x3_virtual_concat(nwide):
    """input_conv over [lr(3), wide sources...] == conv2d over the materialised cat."""
    from pnp_vcve_amd import ops
    h, w = 40, 56
    cin = 3 + 64 * nwide
    lr = gu.syn.uniform(9, 'lr', (1, 3, h, w), 0, 1)
    wides = [gu.syn.uniform(9, f's{j}', (1, 64, h, w), -1, 1) for j in range(nwide)]
    wt = gu.syn.uniform(9, f'w{nwide}', (64, cin, 3, 3), -0.05, 0.05)
    b = gu.syn.uniform(9, 'b', (64,), -0.1, 0.1)
    cat = np.concatenate([lr] + wides, axis=1)
    ref = F.leaky_relu(F.conv2d(torch.from_numpy(cat), torch.from_numpy(wt), torch.from_numpy(b), padding=1), 0.1)
    lr4 = np.concatenate([lr, np.zeros((1, 1, h, w), np.float32)], axis=1)
    srcs = [ops.nchw_to_nhwc(G(lr4))[0]] + [ops.nchw_to_nhwc(G(s))[0] for s in wides]
    wg = G(wt)
    packed = [ops.pack_conv3x3(wg, 0, 3)] + [ops.pack_conv3x3(wg, 3 + 64 * j, 64) for j in range(nwide)]
    out = ops.conv3x3(srcs, packed, bias=G(b), act=2)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_CONV


def _block_via_ops(sd, prefix, x, par, ew, gamma):
    """ResidualBlockNoBNDynamic_drt (channel_first, one_layer) out of two fused conv launches."""
    from pnp_vcve_amd import ops
    xs = ops.nchw_to_nhwc(G(x))[0]
    ewg = G(ew[0])
    w2 = ops.pack_conv3x3(G(sd[prefix + 'conv2.weight']), ew=ewg)
    b2 = (G(sd[prefix + 'conv2.bias']) * ewg[:, None]).sum(0)
    w1x1 = ops.pack_conv1x1([G(sd[prefix + k + '.weight']) for k in ('conv16x16', 'conv16x8', 'conv8x8')])
    o = ops.conv3x3([xs], [w2], bias=b2, gamma=G(gamma[0]), packed_w1x1=w1x1, par=G(par[0]), act=1)
    w1 = ops.pack_conv3x3(G(sd[prefix + 'conv1.weight']))
    y = ops.conv3x3([o], [w1], bias=G(sd[prefix + 'conv1.bias']), residual=xs, act=0)
    return ops.nhwc_to_nchw(y.unsqueeze(0))


@pytest.mark.parametrize('case', gu.BLOCK_CASES, ids=[c['name'] for c in gu.BLOCK_CASES])
def test_bae_block_vs_reference(case):
    cfg, sd, x, par, ew, gamma = gu.block_case_inputs(case)
    out = _block_via_ops(sd, 'backward_resblocks.main.0.', x, par, ew, gamma)
    ref = gu.load_golden(case['name'])['block']
    assert maxdiff(out, ref) < TOL_BLOCK


@pytest.mark.parametrize('case', gu.BLOCK_CASES, ids=[c['name'] for c in gu.BLOCK_CASES])
def test_whole_branch_through_the_c_abi_vs_reference(case):
    """ResidualBlocksWithInputConvDynamic_drt.forward (basicvsr_net.py:506-519) = input conv over the 195-channel concat +
    LeakyReLU + 8 BAE blocks, driven op by op through the C ABI (pnp_conv3x3_f32 over the virtual concat, then
    8 x pnp_bae_block_f32), against the `branch` tensor the reference itself produced (tests/golden/block_*.npz)."""
    from pnp_vcve_amd import ops
    cfg, sd, x, par, ew, gamma = gu.block_case_inputs(case)
    h, w = x.shape[-2:]
    xin = gu.syn.uniform(case['seed'], 'xin', (1, 195, h, w), -1.0, 1.0)
    br = 'forward_resblocks'
    lr4 = np.concatenate([xin[:, :3], np.zeros((1, 1, h, w), np.float32)], axis=1)
    srcs = [ops.nchw_to_nhwc(G(lr4))[0]] + [ops.nchw_to_nhwc(G(np.ascontiguousarray(xin[:, 3 + 64 * j:67 + 64 * j])))[0]
                                             for j in range(3)]
    wg = G(sd[f'{br}.input_conv.0.weight'])
    packed = [ops.pack_conv3x3(wg, 0, 3)] + [ops.pack_conv3x3(wg, 3 + 64 * j, 64) for j in range(3)]
    f = ops.conv3x3(srcs, packed, bias=G(sd[f'{br}.input_conv.0.bias']), act=2)
    ewg, parg, gam = G(ew[0]), G(par[0]), G(gamma[0])
    for i in range(cfg['num_blocks']):
        p = f'{br}.main.{i}.'
        w2 = ops.pack_conv3x3(G(sd[p + 'conv2.weight']), ew=ewg)
        b2 = (G(sd[p + 'conv2.bias']) * ewg[:, None]).sum(0)
        w1x1 = ops.pack_conv1x1([G(sd[p + k + '.weight']) for k in ('conv16x16', 'conv16x8', 'conv8x8')])
        f = ops.bae_block(f, w2, b2, gam, w1x1, parg, ops.pack_conv3x3(G(sd[p + 'conv1.weight'])), G(sd[p + 'conv1.bias']))
    out = ops.nhwc_to_nchw(f.unsqueeze(0))
    ref = gu.load_golden(case['name'])['branch']
    d = maxdiff(out, ref)
    print(case['name'], 'branch max|hip - reference| =', d, 'ref max', float(np.abs(ref).max()))
    assert out.shape == ref.shape and d < 2e-4 * max(1.0, float(np.abs(ref).max()))


def test_bae_block_partition_branch_is_live():
    """zeroing par must change the result by what the oracle says (guards a silently dead 1x1 branch)."""
    case = gu.BLOCK_CASES[2]
    cfg, sd, x, par, ew, gamma = gu.block_case_inputs(case)
    a = _block_via_ops(sd, 'backward_resblocks.main.0.', x, par, ew, gamma)
    b = _block_via_ops(sd, 'backward_resblocks.main.0.', x, par * 0, ew, gamma)
    t = cpu_ref.to_torch_state(sd)
    h, w = x.shape[-2:]
    ra = cpu_ref.bae_block(t, cfg, 'backward_resblocks.main.0.', torch.from_numpy(x),
                           torch.from_numpy(par).view(1, 3, 1, h, w), torch.from_numpy(ew), torch.from_numpy(gamma))
    rb = cpu_ref.bae_block(t, cfg, 'backward_resblocks.main.0.', torch.from_numpy(x),
                           torch.zeros(1, 3, 1, h, w), torch.from_numpy(ew), torch.from_numpy(gamma))
    assert float((ra - rb).abs().max()) > 1e-2
    assert maxdiff(a - b, ra - rb) < TOL_BLOCK


# ---------------------------------------------------------------- full-size properties (720p)
def test_warp_720p_zero_and_integer_motion_exact():
    from pnp_vcve_amd import ops
    h, w = 720, 1280
    feat = torch.rand(h, w, 64, device=dev())
    z = torch.zeros(h, w, device=dev())
    # the reference's normalise/un-normalise round trip (flow_warp.py:41-42 + ATen) costs ~3 ulp of
    # 1280 (1.2e-4 px) in the sample position -- for the CPU path and for this kernel alike -- so
    # zero / integer motion is an identity / shift only to ~1e-4 x local contrast at this size
    assert float((ops.mv_warp_nhwc(feat, z, z) - feat).abs().max()) < 5e-4
    out = ops.mv_warp_nhwc(feat, z + 3.0, z - 2.0)      # sample (x+3, y-2)
    exp = torch.zeros_like(feat)
    exp[2:, :w - 3] = feat[:h - 2, 3:]
    assert float((out - exp).abs().max()) < 5e-4
    # fractional, block-constant motion against the NCHW drop-in (two independent kernels)
    blk = (torch.randint(-32, 33, (2, h // 8, w // 8), device=dev()).float() / 4.0)
    fl = blk.repeat_interleave(8, 1).repeat_interleave(8, 2)
    a = ops.mv_warp_nhwc(feat, fl[0].contiguous(), fl[1].contiguous())
    b = ops.flow_warp(feat.permute(2, 0, 1).unsqueeze(0).contiguous(), fl.permute(1, 2, 0).unsqueeze(0).contiguous())
    assert float((a.permute(2, 0, 1).unsqueeze(0) - b).abs().max()) < 1e-6


def test_conv_720p_scaling_linearity_and_crop_consistency():
    from pnp_vcve_amd import ops
    h, w = 720, 1280
    x = torch.randn(h, w, 64, device=dev())
    wt = (torch.randn(64, 64, 3, 3, device=dev()) * 0.05)
    pw = ops.pack_conv3x3(wt)
    y = ops.conv3x3([x], [pw])
    y2 = ops.conv3x3([x * 2.0], [pw])
    assert torch.equal(y2, y * 2.0)                      # power-of-two scaling is exact in fp32
    # a 96x112 crop, away from the crop border, sees the same pixels
    cy, cx = 301, 517
    yc = ops.conv3x3([x[cy:cy + 96, cx:cx + 112].contiguous()], [pw])
    assert float((yc[1:-1, 1:-1] - y[cy + 1:cy + 95, cx + 1:cx + 111]).abs().max()) < 1e-5
    # and against ATen on the host for that crop
    ref = F.conv2d(x[cy:cy + 96, cx:cx + 112].permute(2, 0, 1).unsqueeze(0).cpu(), wt.cpu(), padding=1)
    assert maxdiff(yc.permute(2, 0, 1).unsqueeze(0), ref) < TOL_CONV * 4


@pytest.mark.parametrize('case', gu.METRIC_CASES, ids=[c['name'] for c in gu.METRIC_CASES])
def test_psnr_and_rgb8_on_device_match_the_reference_functions(case):
    """pnp_psnr_sse_f32 / pnp_frames_to_rgb8 against what the reference's own tensor2img + psnr (+ BasicVSR.evaluate) returned for
    the same frames (tests/golden/metrics_psnr_*.npz, oracle/gen_golden.py): values outside [0,1], exact .5 rounding ties, an
    identical pair (inf), crop_border 0 and 3.  The comparator is the fixture -- not the package's host code."""
    from pnp_vcve_amd import ops
    from pnp_vcve_amd.restorer import BasicVSR
    out, gt = gu.metric_case_inputs(case)
    g = gu.load_golden(case['name'])
    o, t = G(out), G(gt)
    rgb = ops.frames_to_rgb8(o[0]).cpu().numpy()                     # (T,h,w,3) RGB; the reference's images are BGR
    assert np.array_equal(rgb[..., ::-1], g['img_out'])
    assert np.array_equal(ops.frames_to_rgb8(t[0]).cpu().numpy()[..., ::-1], g['img_gt'])
    for crop in (0, 3):
        got = ops.psnr_frames(o[0], t[0], crop).numpy()
        ref = g[f'psnr_crop{crop}']
        for i in range(len(ref)):
            if np.isinf(ref[i]):
                assert np.isinf(got[i]) and got[i] > 0
            else:
                assert abs(got[i] - ref[i]) < 1e-4, (crop, i, got[i], ref[i])   # the reference's value is a float32
        m = BasicVSR.__new__(BasicVSR)
        m.test_cfg = dict(metrics=['PSNR'], crop_border=crop)
        fin = [int(i) for i in g[f'finite_frames_crop{crop}']]
        ev = BasicVSR.evaluate(m, o[:, fin], t[:, fin])['PSNR']        # CUDA tensors: the on-device statistic
        assert abs(ev - float(g[f'evaluate_finite_crop{crop}'])) < 1e-4
        assert BasicVSR.evaluate(m, o, t)['PSNR'] == float('inf') == float(g[f'evaluate_all_crop{crop}'])


def test_psnr_on_device_exact_integer_statistic_at_720p():
    from pnp_vcve_amd import ops
    # 720p frames, exact integer statistic vs the host
    x = torch.rand(2, 3, 720, 1280, device=dev())
    y = (x + 0.02 * torch.randn_like(x)).clamp(0, 1)
    g = ops.psnr_frames(x, y)
    xi = (x.clamp(0, 1) * 255).round()
    yi = (y * 255).round()
    ref = 20 * torch.log10(255.0 / ((xi - yi).double() ** 2).mean(dim=(1, 2, 3)).sqrt())
    assert float((g - ref.cpu()).abs().max()) < 1e-9


@pytest.mark.parametrize('case', gu.RASTER_CASES, ids=[c['name'] for c in gu.RASTER_CASES])
def test_rasteriser_vs_reference_loader(case):
    """bit-exact against the maps the reference's LoadImageFromFileList_ipb produced (tests/golden)."""
    from pnp_vcve_amd import ops
    rec, rec_frame, slices, h, w = gu.raster_case_inputs(case)
    mvs, par = ops.rasterise_side_info(G(rec), torch.from_numpy(rec_frame).to(dev()), slices, h, w)
    g = gu.load_golden(case['name'])
    assert np.array_equal(mvs.cpu().numpy(), g['mvs'])
    assert np.array_equal(par.cpu().numpy(), g['partitions'])


def test_rasteriser_720p_wraparound_and_overwrite_order_vs_oracle():
    """a REDS-sized frame set with thousands of records, blocks hanging over every edge (incl. the python
    wrap-around case y + h/2 < 0) and heavy overlap: the last record must win, as in the loader's loop."""
    from pnp_vcve_amd import ops
    h, w, slices = 720, 1280, 'IBBPBBP'
    rng = np.random.RandomState(3)
    rows, frames = [], []
    for f, sl in enumerate(slices):
        if sl == 'I':
            continue
        n = 6000
        sz = np.array([(16, 16), (16, 8), (8, 16), (8, 8)])[rng.randint(0, 4, n)]
        cx = rng.randint(-6, w // 4 + 6, n) * 4
        cy = rng.randint(-6, h // 4 + 6, n) * 4
        mx, my = rng.randint(-64, 65, n), rng.randint(-64, 65, n)
        dr = rng.randint(0, 2, n) * 2 - 1
        for i in range(n):
            rows.append([dr[i], sz[i, 0], sz[i, 1], cx[i] + mx[i] // 4, cy[i] + my[i] // 4, cx[i], cy[i], mx[i], my[i], 4])
            frames.append(f)
    rec, rec_frame = np.array(rows, np.float32), np.array(frames, np.int32)
    mvs, par = ops.rasterise_side_info(G(rec), torch.from_numpy(rec_frame).to(dev()), slices, h, w)
    rm, rp = cpu_ref.rasterise_side_info(rec, rec_frame, slices, h, w)
    assert np.array_equal(mvs.cpu().numpy(), rm)
    assert np.array_equal(par.cpu().numpy(), rp)
    assert (rm[3, 2:] != 0).any() and (rm[0, 2:] != 0).any()      # P frames painted the previous anchors


@pytest.mark.parametrize('with_flow', [False, True])
def test_modulated_deform_conv_vs_oracle(with_flow):
    """DCN aligner core (deform = basic | fvc).  Parity is against the oracle's restatement of the mmcv
    semantics (mmcv itself is not vendored by the reference -> unpinned), incl. samples leaving the frame."""
    from pnp_vcve_amd import ops
    h, w = 40, 56
    x = gu.syn.uniform(21, 'x', (1, 64, h, w), -1, 1)
    off = gu.syn.uniform(21, 'off', (1, 288, h, w), -3.0, 3.0)
    off[:, :, :4] *= 8.0                                     # top rows: far out of frame
    off[:, ::7] = np.round(off[:, ::7])                      # some exactly integer offsets
    ml = gu.syn.uniform(21, 'm', (1, 144, h, w), -2.0, 2.0)
    wt = gu.syn.uniform(21, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(21, 'b', (64,), -0.1, 0.1)
    flow = gu.syn.uniform(21, 'fl', (1, 2, h, w), -4.0, 4.0)
    T = torch.from_numpy
    offset_ref = T(off)
    if with_flow:                                            # iconvsr_mv.py:77
        offset_ref = offset_ref + T(flow).flip(1).repeat(1, 144, 1, 1)
    ref = cpu_ref.modulated_deform_conv2d(T(x), offset_ref, torch.sigmoid(T(ml)), T(wt), T(b), 16)
    out = ops.modulated_deform_conv_nhwc(ops.nchw_to_nhwc(G(x))[0], G(off[0]), G(ml[0]), G(wt), G(b),
                                         flow=G(flow[0]) if with_flow else None)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < 5e-5


@pytest.mark.parametrize('with_flow', [False, True])
def test_modulated_deform_conv_fp16_operands_track_fp32(with_flow):
    """PNP_PREC_F16's DCN contraction (v_mfma_f32_32x32x16_f16 over the lane-gathered k order, dcn_f16_image_kernel): a
    layout slip would be an O(1) error; operand rounding is ~1e-3 of the output scale.  Block-constant flow + small
    offsets (the LDS-window path) and a few far ones (the global fallback)."""
    from pnp_vcve_amd import ops
    h, w = 72, 80
    x = G(gu.syn.uniform(22, 'x', (h, w, 64), -1, 1))
    off = gu.syn.uniform(22, 'off', (288, h, w), -1.0, 1.0)
    off[:, :3] *= 9.0
    ml = gu.syn.uniform(22, 'm', (144, h, w), -2.0, 2.0)
    wt = gu.syn.uniform(22, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(22, 'b', (64,), -0.1, 0.1)
    blk = gu.syn.randint(22, 'fl', (2, h // 8, w // 8), -32, 32).astype(np.float32) / 4.0
    flow = G(np.ascontiguousarray(np.repeat(np.repeat(blk, 8, 1), 8, 2))) if with_flow else None
    o32 = ops.modulated_deform_conv_nhwc(x, G(off), G(ml), G(wt), G(b), flow=flow)
    o16 = ops.modulated_deform_conv_nhwc(x, G(off), G(ml), G(wt), G(b), flow=flow, fp16=True)
    for _ in range(4):                                      # run-to-run determinism (see the note in dcn.hip's tap loop)
        assert torch.equal(ops.modulated_deform_conv_nhwc(x, G(off), G(ml), G(wt), G(b), flow=flow, fp16=True), o16)
        assert torch.equal(ops.modulated_deform_conv_nhwc(x, G(off), G(ml), G(wt), G(b), flow=flow), o32)
    d = float((o16 - o32).abs().max())
    scale = float(o32.abs().max())
    print('dcn fp16 vs fp32', d, 'scale', scale)
    assert 0.0 < d < 4e-3 * scale


def test_dcn_fp16_is_bit_stable_at_720p_under_traffic():
    """The fp16 DCN kernel at full size (28 tiles per block, global-memory fallback samples included), 10 runs with unrelated
    device traffic in between: bit-identical every time and within operand rounding of the fp32 kernel everywhere.  Round 2's
    build (gather arithmetic SLP-packed into v_pk_*_f32) failed this in about one run of ten with a few hundred elements off by
    up to 0.1 (profiles/r03_dcn_hazard_report.txt); dcn.hip is built with -fno-slp-vectorize since."""
    from pnp_vcve_amd import ops
    h, w = 720, 1280
    g = torch.Generator(device=dev()).manual_seed(7)
    x = torch.randn(h, w, 64, device=dev(), generator=g)
    off = torch.randn(288, h, w, device=dev(), generator=g) * 1.5
    ml = torch.randn(144, h, w, device=dev(), generator=g)
    blk = torch.randint(-16, 17, (2, h // 8, w // 8), device=dev(), generator=g).float() / 4
    flow = blk.repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous()
    wt = torch.randn(64, 64, 3, 3, device=dev(), generator=g) * 0.05
    b = torch.randn(64, device=dev(), generator=g) * 0.1
    o32 = ops.modulated_deform_conv_nhwc(x, off, ml, wt, b, flow=flow)
    o16 = ops.modulated_deform_conv_nhwc(x, off, ml, wt, b, flow=flow, fp16=True)
    scale = float(o32.abs().max())
    assert float((o16 - o32).abs().max()) < 4e-3 * scale
    for rep in range(10):
        torch.randn(1 << 22, device=dev()).sin_()
        again = ops.modulated_deform_conv_nhwc(x, off, ml, wt, b, flow=flow, fp16=True)
        assert torch.equal(again, o16), (rep, int((again != o16).sum()))
        if rep % 3 == 0:
            assert torch.equal(ops.modulated_deform_conv_nhwc(x, off, ml, wt, b, flow=flow), o32)


@pytest.mark.parametrize('fp16', [False, True], ids=['fp32', 'fp16'])
def test_dcn_at_720p_collapses_to_the_conv_kernel_and_to_a_shifted_conv(fp16):
    """full-size properties of the DCN kernel (persistent blocks, LDS windows): with zero offsets and saturated masks it is
    the plain 3x3 conv (checked against the fused conv kernel), and with a block-constant INTEGER flow added to every offset
    ('basic', iconvsr_mv.py:77) it is the conv of the image shifted by that flow -- per 8x8 block, i.e. every window position."""
    from pnp_vcve_amd import ops
    h, w = 720, 1280
    x = torch.randn(h, w, 64, device=dev())
    wt = torch.randn(64, 64, 3, 3, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    off = torch.zeros(288, h, w, device=dev())
    ml = torch.full((144, h, w), 30.0, device=dev())             # sigmoid(30) = 1 - 9e-14
    tol = 5e-3 if fp16 else 1e-4          # fp16 operand rounding on randn features over 5.9e7 outputs: 2.1e-3 observed
    ref = ops.conv3x3([x], [ops.pack_conv3x3(wt)], bias=b)
    out = ops.modulated_deform_conv_nhwc(x, off, ml, wt, b, fp16=fp16)
    assert maxdiff(out, ref.cpu()) < tol
    # one integer flow for the whole frame: out(p) = conv(x)(p + flow) wherever p + flow and its 3x3 support stay inside
    dy, dx = 3, -5
    flow = torch.zeros(2, h, w, device=dev())
    flow[0] = dx
    flow[1] = dy
    out = ops.modulated_deform_conv_nhwc(x, off, ml, wt, b, flow=flow, fp16=fp16)
    assert maxdiff(out[8:-8, 8:-8], ref[8 + dy:h - 8 + dy, 8 + dx:w - 8 + dx].cpu()) < tol
    # a different integer flow per 8x8 block (every window gets its own origin): sample a few blocks against the shifted conv
    blk = torch.from_numpy(gu.syn.randint(33, 'f', (2, h // 8, w // 8), -8, 8).astype(np.float32)).to(dev())
    flow = blk.repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous()
    out = ops.modulated_deform_conv_nhwc(x, off, ml, wt, b, flow=flow, fp16=fp16)
    for by, bx in ((5, 7), (40, 90), (44, 3), (80, 150), (17, 101)):
        fx, fy = int(blk[0, by, bx]), int(blk[1, by, bx])
        ys, xs = slice(8 * by + 1, 8 * by + 7), slice(8 * bx + 1, 8 * bx + 7)       # interior of the block (3x3 support stays in one flow)
        got = out[ys, xs]
        want = ref[8 * by + 1 + fy:8 * by + 7 + fy, 8 * bx + 1 + fx:8 * bx + 7 + fx]
        assert maxdiff(got, want.cpu()) < tol, (by, bx, fx, fy)


@pytest.mark.parametrize('hw', [(256, 512), (260, 516), (720, 1280)])
def test_persistent_conv_is_bit_identical_to_the_tile_per_block_kernel(hw):
    """frames with >= 1024 tiles run the persistent kernel (conv_persist.hip): same arithmetic in the same
    order as conv_mfma.hip, so plain / residual (in place) / gamma+par variants must match bit for bit."""
    from pnp_vcve_amd import _native, ops
    h, w = hw
    x = torch.randn(h, w, 64, device=dev())
    r = torch.randn(h, w, 64, device=dev())
    wt = torch.randn(64, 64, 3, 3, device=dev()) * 0.05
    pw = ops.pack_conv3x3(wt)
    p1 = ops.pack_conv1x1([torch.randn(64, 64, 1, 1, device=dev()) * 0.1 for _ in range(3)])
    bias = torch.randn(64, device=dev()) * 0.1
    gamma = torch.rand(64, device=dev()) * 2
    par = torch.rand(3, h, w, device=dev())
    variants = [dict(bias=bias, act=2), dict(bias=bias, residual=r, act=0),
                dict(bias=bias, gamma=gamma, packed_w1x1=p1, par=par, act=1)]
    for kw in variants:
        a = ops.conv3x3([x], [pw], variant=_native.CONV_TILE, **kw)        # include/pnpvcve_debug.h
        b = ops.conv3x3([x], [pw], variant=_native.CONV_AUTO, **kw)
        assert torch.equal(a, b), sorted(kw)
    ref = F.leaky_relu(F.conv2d(x[:64, :96].permute(2, 0, 1).unsqueeze(0).cpu(), wt.cpu(), bias.cpu(), padding=1), 0.1)
    got = ops.conv3x3([x], [pw], bias=bias, act=2)[:63, :95].permute(2, 0, 1).unsqueeze(0)
    assert maxdiff(got, ref[..., :63, :95]) < TOL_CONV * 4


@pytest.mark.parametrize('crop', [0, 2])
def test_ssim_on_device_matches_host_definition(crop):
    from pnp_vcve_amd import ops
    from pnp_vcve_amd.metrics import ssim, tensor2img
    a = torch.rand(3, 3, 45, 77)
    b = (a + 0.08 * torch.randn_like(a))
    b[2] = a[2]
    got = ops.ssim_frames(a.to(dev()), b.to(dev()), crop)
    for i in range(3):
        ref = ssim(tensor2img(a[i]), tensor2img(b[i]), crop)
        assert abs(float(got[i]) - ref) < 1e-10, (i, float(got[i]), ref)
    assert abs(float(got[2]) - 1.0) < 1e-12


@pytest.mark.parametrize('crop', [0, 3])
def test_ssim_on_device_matches_independent_scipy_restatement(crop):
    """the fp64 SSIM kernel against the scipy.ndimage restatement of the reference's cv2 formula
    (tests/test_host_logic.py::_ssim_scipy), which shares no code with pnp_vcve_amd.metrics"""
    from pnp_vcve_amd import ops
    from pnp_vcve_amd.metrics import tensor2img
    from test_host_logic import _ssim_scipy
    a = torch.from_numpy(gu.syn.uniform01(61, 'a', (2, 3, 52, 83)))
    b = (a + 0.06 * torch.from_numpy(gu.syn.normal(61, 'n', (2, 3, 52, 83)))).clamp(0, 1)
    got = ops.ssim_frames(a.to(dev()), b.to(dev()), crop)
    for i in range(2):
        ref = _ssim_scipy(tensor2img(a[i]), tensor2img(b[i]), crop)
        assert abs(float(got[i]) - ref) < 1e-10, (i, float(got[i]), ref)


def test_frames_to_rgb8_matches_tensor2img():
    """write-back conversion on the device == the reference's tensor2img (core/misc.py:51-71), incl. ties and clamping"""
    from pnp_vcve_amd import ops
    from pnp_vcve_amd.metrics import tensor2img
    x = torch.from_numpy(gu.syn.uniform(31, 'img', (4, 3, 37, 53), -0.1, 1.1))
    x[0, 0, 0, :8] = torch.tensor([0.5, 1.5, 2.5, 3.5, 126.5, 127.5, 253.5, 254.5]) / 255.0
    q = ops.frames_to_rgb8(x.to(dev())).cpu().numpy()
    for i in range(4):
        assert np.array_equal(q[i], tensor2img(x[i])[..., ::-1])


def test_par_tile_flags_vs_numpy():
    from pnp_vcve_amd import ops
    h, w = 100, 150                                        # ragged: 13 x 10 tiles
    par = np.zeros((3, h, w), np.float32)
    par[0, 0:8, 0:16] = 1 / 255.0                          # tile (0,0): plane 0
    par[1, 8:16, 16:32] = 0.5                              # tile (1,1): plane 1
    par[2, 95:100, 140:150] = 1e-30                        # last (ragged) tile: plane 2, tiny but nonzero
    par[0, 40, 70] = -0.0                                  # negative zero is zero
    par[1, 17, 5] = 1.0
    par[2, 17, 6] = 1.0                                    # tile (2,0): planes 1 and 2
    got = ops.par_tile_flags(G(par)).cpu().numpy()
    exp = np.zeros(((h + 7) // 8, (w + 15) // 16), np.int32)
    for ty in range(exp.shape[0]):
        for tx in range(exp.shape[1]):
            blk = par[:, ty * 8:ty * 8 + 8, tx * 16:tx * 16 + 16]
            exp[ty, tx] = sum(int((blk[j] != 0).any()) << j for j in range(3))
            # bits 3..5 (r04): every value of plane j in the tile is 0 or exactly float32(1) / float32(255) -- the reference loader's value
            unit = np.float32(1.0) / np.float32(255.0)
            exp[ty, tx] |= sum(int(((blk[j] == 0) | (blk[j] == unit)).all()) << (3 + j) for j in range(3))
            # bit 6 (r05; r06: any frame size): each of its 8x8 halves is, on its pixels INSIDE the image (`blk` is cropped to it), all
            # zero or has exactly ONE live plane that is constant there (what conv_wino.hip folds into its weights); a half wholly
            # outside counts as empty
            ok = True
            for half in (blk[:, :, :8], blk[:, :, 8:]):
                live = [j for j in range(3) if (half[j] != 0).any()]
                ok = ok and (not live or (len(live) == 1 and (half[live[0]] == half[live[0]].flat[0]).all()))
            exp[ty, tx] |= int(ok) << 6
    assert np.array_equal(got, exp)
    # (0,0): plane 0 constant on both halves; (1,1): plane 1 constant; (2,0): two planes live in one half; last tile: ragged (4 x 6 pixels
    # inside, plane 2 constant on them; its right half lies outside); (11,8): plane 2 live on one row only; (5,5): empty
    assert got[0, 0] == 1 | 56 | 64 and got[1, 1] == 2 | 8 | 32 | 64 and got[2, 0] == 6 | 8 and got[-1, -1] == 4 | 8 | 16 | 64 and got[5, 5] == 56 | 64
    assert got[11, 8] == 4 | 8 | 16


def test_torch_custom_ops_call_the_hip_kernels():
    """torch.ops.pnpvcve.flow_warp / mv_warp == the ctypes wrappers (same C-ABI calls underneath)."""
    import pnp_vcve_amd  # noqa: F401
    from pnp_vcve_amd import ops
    case = gu.WARP_CASES[1]
    x, flow = gu.warp_case_inputs(case)
    a = torch.ops.pnpvcve.flow_warp(G(x), G(flow))
    assert torch.equal(a, ops.flow_warp(G(x), G(flow)))
    assert maxdiff(a, gu.load_golden(case['name'])['out']) < TOL_WARP
    feat = G(gu.syn.uniform(41, 'f', (40, 56, 64), -1, 1))
    fx = G(gu.syn.uniform(41, 'fx', (40, 56), -3, 3))
    fy = G(gu.syn.uniform(41, 'fy', (40, 56), -3, 3))
    assert torch.equal(torch.ops.pnpvcve.mv_warp(feat, fx, fy), ops.mv_warp_nhwc(feat, fx, fy))
    with pytest.raises(RuntimeError):
        torch.ops.pnpvcve.generator_forward(987654, feat, feat, feat, torch.zeros(3, 1, 1))


# ------------------------------------------------------------------ torch.ops.pnpvcve.* (SURVEY 8b's custom-op list)
def test_custom_ops_equal_their_ctypes_wrappers_and_the_aten_definition():
    """torch.ops.pnpvcve.{conv3x3, expert_mix, bae_block, pixel_shuffle_conv}: each equals the ops.* wrapper bit for bit
    and the ATen definition of what it replaces within the conv tolerance."""
    import pnp_vcve_amd  # noqa: F401
    from pnp_vcve_amd import ops
    P = torch.ops.pnpvcve
    h, w = 40, 48
    x = torch.randn(h, w, 64, device=dev())
    xn = x.permute(2, 0, 1).unsqueeze(0).cpu()
    experts = torch.randn(6, 64, 64, 3, 3, device=dev()) * 0.05
    att = torch.softmax(torch.randn(6, device=dev()), 0)
    # expert_mix == Dynamic_conv2d_se's mm(attention, weight) then packed (sr_backbone_utils.py:198-199)
    w2p = P.expert_mix(experts, att)
    assert torch.equal(w2p, ops.pack_conv3x3(experts, ew=att))
    w2 = torch.einsum('e,eoikl->oikl', att.cpu(), experts.cpu())
    b2 = torch.randn(64, device=dev()) * 0.1
    gamma = torch.rand(64, device=dev()) * 2
    # conv3x3
    got = P.conv3x3([x], [w2p], b2, gamma, None, None, None, 1)
    assert torch.equal(got, ops.conv3x3([x], [w2p], bias=b2, gamma=gamma, act=1))
    ref = F.relu((F.conv2d(xn, w2, b2.cpu(), padding=1)) * gamma.cpu().view(1, -1, 1, 1))
    assert maxdiff(got.permute(2, 0, 1).unsqueeze(0), ref) < TOL_CONV * 4
    # bae_block == x + conv1(relu(gamma * (conv2(x) + b2) + sum_j par_j * conv1x1_j(x))) + b1
    w1 = torch.randn(64, 64, 3, 3, device=dev()) * 0.05
    b1 = torch.randn(64, device=dev()) * 0.1
    k1 = [torch.randn(64, 64, 1, 1, device=dev()) * 0.1 for _ in range(3)]
    par = (torch.rand(3, h, w, device=dev()) > 0.5).float() * torch.rand(3, h, w, device=dev())
    w1p, k1p = ops.pack_conv3x3(w1), ops.pack_conv1x1(k1)
    got = P.bae_block(x, w2p, b2, gamma, k1p, par, w1p, b1)
    assert torch.equal(got, ops.bae_block(x, w2p, b2, gamma, k1p, par, w1p, b1))
    two = ops.conv3x3([ops.conv3x3([x], [w2p], bias=b2, gamma=gamma, packed_w1x1=k1p, par=par, act=1)], [w1p], bias=b1,
                      residual=x)
    assert torch.equal(got, two)
    dy = sum(F.conv2d(xn, k1[j].cpu()) * par[j].cpu() for j in range(3))
    mid = F.relu(F.conv2d(xn, w2, b2.cpu(), padding=1) * gamma.cpu().view(1, -1, 1, 1) + dy)
    ref = xn + F.conv2d(mid, w1.cpu(), b1.cpu(), padding=1)
    assert maxdiff(got.permute(2, 0, 1).unsqueeze(0), ref) < 5e-5
    # pixel_shuffle_conv == PixelShufflePack.forward (+ the head's leaky-relu)
    wu = torch.randn(256, 64, 3, 3, device=dev()) * 0.05
    bu = torch.randn(256, device=dev()) * 0.1
    pk = ops.pack_pixel_shuffle(wu, bu)
    got = P.pixel_shuffle_conv(x, pk, 2)
    assert got.shape == (2 * h, 2 * w, 64) and torch.equal(got, ops.pixel_shuffle_conv(x, pk, 2))
    ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(xn, wu.cpu(), bu.cpu(), padding=1), 2), 0.1)
    assert maxdiff(got.permute(2, 0, 1).unsqueeze(0), ref) < TOL_CONV * 4

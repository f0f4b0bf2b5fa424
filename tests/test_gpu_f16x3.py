"""GPU tests of the opt-in split-fp16 conv path (PNP_PREC_F16X3; csrc/conv_f16x3.hip), through the C ABI.

Every operand of the 64-channel convs is carried as two fp16 numbers (hi + lo / 2048) and a product is three fp16 MFMAs with
fp32 accumulation.  Unlike PNP_PREC_F16 the mode is held to the SAME gates as the exact fp32 path: the conv op against an fp64
contraction of the UNROUNDED operands at the fp32 kernel's tolerance, the generator against the golden outputs of the imported
reference at 1e-4 (north_star's gate is 1e-3).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
from oracle import cpu_ref

pytestmark = pytest.mark.gpu

TOL_CONV = 2e-5       # tests/test_gpu_ops.py's tolerance for the exact-fp32 MFMA conv
TOL_PATH = 5e-6       # tests/test_gpu_generator.py's tolerance for the exact-fp32 path (north_star: 1e-3; observed 1.2-1.8e-7)


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def D(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


@pytest.mark.parametrize('hw', [(16, 16), (24, 40), (37, 53), (64, 64), (128, 256), (180, 320)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_f16x3_conv_single_source_matches_fp64_on_unrounded_operands(hw, act):
    from pnp_vcve_amd import ops
    h, w = hw
    x = gu.syn.uniform(31, f'x{h}x{w}', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(31, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(31, 'b', (64,), -0.1, 0.1)
    res = gu.syn.uniform(31, f'r{h}x{w}', (1, 64, h, w), -1, 1)
    ref = F.conv2d(D(x), D(wt), D(b), padding=1)
    ref = [ref, F.relu(ref), F.leaky_relu(ref, 0.1)][act] + D(res)
    xs = ops.nchw_to_nhwc(G(x))[0]
    rs = ops.nchw_to_nhwc(G(res))[0]
    pw = ops.pack_conv3x3(G(wt))
    out = ops.conv3x3_f16x3([xs], [pw], bias=G(b), residual=rs, act=act)
    d3 = maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref)
    d32 = maxdiff(ops.nhwc_to_nchw(ops.conv3x3([xs], [pw], bias=G(b), residual=rs, act=act).unsqueeze(0)), ref)
    d16 = maxdiff(ops.nhwc_to_nchw(ops.conv3x3([xs], [ops.f16_image(pw)], bias=G(b), residual=rs, act=act, fp16=True).unsqueeze(0)), ref)
    print(f'{h}x{w} act {act}: |d| vs fp64: f16x3 {d3:.2e}, fp32 MFMA {d32:.2e}, fp16 operands {d16:.2e}')
    assert d3 < TOL_CONV / 4          # observed ~1e-6, like the fp32 kernel
    assert d16 > 20 * d3              # plain fp16 operands are two orders worse: the split really ran


def test_f16x3_conv_identity_weights_reproduce_22_bits_and_localise_layout_bugs():
    """Identity weights: out = hi + lo / 2048 of x, i.e. x to 2^-22 relative; a wrong tap or channel shows as O(1)."""
    from pnp_vcve_amd import ops
    h, w = 24, 40
    x = gu.syn.uniform(32, 'x', (1, 64, h, w), -1, 1)
    perm = np.roll(np.arange(64), 5)
    for (ky, kx) in [(1, 1), (0, 0), (2, 1), (1, 2)]:
        wt = np.zeros((64, 64, 3, 3), np.float32)
        wt[np.arange(64), perm, ky, kx] = 1.0
        ref = F.conv2d(D(x), D(wt), padding=1)
        out = ops.conv3x3_f16x3([ops.nchw_to_nhwc(G(x))[0]], [ops.pack_conv3x3(G(wt))])
        assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) <= 2.0 ** -22, (ky, kx)


@pytest.mark.parametrize('scale', [1e-3, 3e-5, 1e-6])
def test_f16x3_small_magnitudes_keep_their_relative_accuracy(scale):
    """Activations far below fp16's normal range (6.1e-5): hi underflows towards 0 and lo * 2048 carries the value."""
    from pnp_vcve_amd import ops
    h, w = 32, 48
    x = gu.syn.uniform(33, 'x', (1, 64, h, w), -1, 1) * np.float32(scale)
    wt = gu.syn.uniform(33, 'w', (64, 64, 3, 3), -0.06, 0.06)
    ref = F.conv2d(D(x), D(wt), padding=1)
    out = ops.conv3x3_f16x3([ops.nchw_to_nhwc(G(x))[0]], [ops.pack_conv3x3(G(wt))])
    d = maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref)
    print(f'scale {scale:g}: max|d| {d:.2e} of |ref|max {float(ref.abs().max()):.2e}')
    assert d < max(1e-5 * float(ref.abs().max()), 1e-9)


@pytest.mark.parametrize('nwide', [1, 2, 3])
@pytest.mark.parametrize('with_lr', [False, True])
def test_f16x3_conv_virtual_concat_is_one_launch(nwide, with_lr):
    """input_conv over [lr(3), wide...] (iconvsr_ipb_par.py:90,125) in ONE launch of the multi-source kernel: the RGB frame as two
    32-deep chunks in front, then one pass per 64-channel source into the same accumulators; 40x56 runs the 4x16-tile variant,
    the 264x272 case below the 8x16 one.  (With a trace buffer the r03 launch chain runs instead: see the last assertion.)"""
    from pnp_vcve_amd import ops
    h, w = 40, 56
    cin = (3 if with_lr else 0) + 64 * nwide
    lr = gu.syn.uniform(34, 'lr', (1, 3, h, w), 0, 1)
    wides = [gu.syn.uniform(34, f's{j}', (1, 64, h, w), -1, 1) for j in range(nwide)]
    wt = gu.syn.uniform(34, f'w{cin}', (64, cin, 3, 3), -0.05, 0.05)
    b = gu.syn.uniform(34, 'b', (64,), -0.1, 0.1)
    cat = np.concatenate(([lr] if with_lr else []) + wides, axis=1)
    ref = F.leaky_relu(F.conv2d(D(cat), D(wt), D(b), padding=1), 0.1)
    lr4 = np.concatenate([lr, np.zeros((1, 1, h, w), np.float32)], axis=1)
    srcs = ([ops.nchw_to_nhwc(G(lr4))[0]] if with_lr else []) + [ops.nchw_to_nhwc(G(s))[0] for s in wides]
    wg = G(wt)
    c0 = 3 if with_lr else 0
    packed = ([ops.pack_conv3x3(wg, 0, 3)] if with_lr else []) + [ops.pack_conv3x3(wg, c0 + 64 * j, 64) for j in range(nwide)]
    out = ops.conv3x3_f16x3(srcs, packed, bias=G(b), act=2)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_CONV / 2
    # the traced variant keeps the launch chain (RGB link on the exact fp32 kernel, partial sums through `out`): same result to
    # the last few bits, not bit for bit (one long accumulation vs one per source)
    dbg = torch.zeros(512 * 8, dtype=torch.int64, device=dev())
    chain = ops.conv3x3_f16x3(srcs, packed, bias=G(b), act=2, trace=dbg)
    assert maxdiff(chain, out.cpu()) < 2e-6


@pytest.mark.parametrize('nwide,with_lr', [(1, True), (2, True), (3, True), (3, False)])
def test_f16x3_multi_source_kernel_on_8x16_tiles_with_ragged_edges(nwide, with_lr):
    """the same conv on a frame of more than 256 8x16 tiles (264 x 272: ragged right column of tiles, 33 x 17 = 561 tiles) against fp64"""
    from pnp_vcve_amd import ops
    h, w = 264, 272
    cin = (3 if with_lr else 0) + 64 * nwide
    lr = gu.syn.uniform(37, 'lr', (1, 3, h, w), 0, 1)
    wides = [gu.syn.uniform(37, f's{j}', (1, 64, h, w), -1, 1) for j in range(nwide)]
    wt = gu.syn.uniform(37, f'w{cin}', (64, cin, 3, 3), -0.05, 0.05)
    b = gu.syn.uniform(37, 'b', (64,), -0.1, 0.1)
    cat = np.concatenate(([lr] if with_lr else []) + wides, axis=1)
    ref = F.leaky_relu(F.conv2d(D(cat), D(wt), D(b), padding=1), 0.1)
    lr4 = np.concatenate([lr, np.zeros((1, 1, h, w), np.float32)], axis=1)
    srcs = ([ops.nchw_to_nhwc(G(lr4))[0]] if with_lr else []) + [ops.nchw_to_nhwc(G(s))[0] for s in wides]
    wg = G(wt)
    c0 = 3 if with_lr else 0
    packed = ([ops.pack_conv3x3(wg, 0, 3)] if with_lr else []) + [ops.pack_conv3x3(wg, c0 + 64 * j, 64) for j in range(nwide)]
    out = ops.conv3x3_f16x3(srcs, packed, bias=G(b), act=2)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_CONV / 2
    again = ops.conv3x3_f16x3(srcs, packed, bias=G(b), act=2)
    assert torch.equal(again, out)


def test_f16x3_conv_unsupported_shapes_are_refused_not_silently_rerouted():
    from pnp_vcve_amd import ops
    h, w = 16, 16
    lr4 = torch.zeros(h, w, 4, device=dev())
    wt = gu.syn.uniform(35, 'w', (64, 3, 3, 3), -0.05, 0.05)
    with pytest.raises(RuntimeError):          # an RGB-only conv has no split form
        ops.conv3x3_f16x3([lr4], [ops.pack_conv3x3(G(wt), 0, 3)])
    x = torch.zeros(h, w, 64, device=dev())
    w64 = ops.pack_conv3x3(G(gu.syn.uniform(35, 'w64', (64, 64, 3, 3), -0.05, 0.05)))
    with pytest.raises(RuntimeError):          # several sources exclude a residual
        ops.conv3x3_f16x3([x, x], [w64, w64], residual=x)


def _front_half_inputs(seed, h, w, par_scale):
    x = gu.syn.uniform(seed, f'x{h}', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(seed, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(seed, 'b', (64,), -0.1, 0.1)
    gam = gu.syn.uniform(seed, 'g', (64,), 0.5, 1.5)
    w1 = [gu.syn.uniform(seed, f'w1_{j}', (64, 64, 1, 1), -0.1, 0.1) for j in range(3)]
    cls = (gu.syn.uniform(seed, f'c{h}', ((h + 7) // 8, (w + 7) // 8), 0, 3).astype(np.int64)).clip(0, 2)
    par = np.zeros((3, h, w), np.float32)
    for j in range(3):
        par[j] = np.kron((cls == j).astype(np.float32), np.ones((8, 8), np.float32))[:h, :w]
    par *= np.float32(par_scale)
    return x, wt, b, gam, w1, par


@pytest.mark.parametrize('hw', [(32, 48), (72, 88), (37, 53)])
@pytest.mark.parametrize('skip', [False, True])
def test_f16x3_bae_front_half_matches_fp64(hw, skip):
    """relu(gamma * (conv3x3(x) + b) + sum_j par_j * conv1x1_j(x)) (sr_backbone_utils.py:305-311): the split kernel sums each
    1x1 branch on its own and scales it by par_j(pixel) on the output side, in fp32.  With and without per-tile branch
    skipping (one-hot maps: most tiles need one or two of the three branches)."""
    from pnp_vcve_amd import ops
    h, w = hw
    x, wt, b, gam, w1, par = _front_half_inputs(36, h, w, 200.0 / 255.0)
    ref = F.conv2d(D(x), D(wt), D(b), padding=1) * D(gam).view(1, 64, 1, 1)
    for j in range(3):
        ref = ref + D(par[j]).view(1, 1, h, w) * F.conv2d(D(x), D(w1[j]))
    ref = F.relu(ref)
    xs = ops.nchw_to_nhwc(G(x))[0]
    pg = G(par)
    flags = ops.par_tile_flags(pg) if skip else None
    out = ops.conv3x3_f16x3([xs], [ops.pack_conv3x3(G(wt))], bias=G(b), gamma=G(gam),
                            packed_w1x1=ops.pack_conv1x1([G(v) for v in w1]), par=pg, par_flags=flags, act=1)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_CONV / 2


def test_f16x3_dense_float_partition_map():
    """A dense float partition map (every plane nonzero everywhere; configs with par_scale): all three branches on every tile."""
    from pnp_vcve_amd import ops
    h, w = 40, 64
    x, wt, b, gam, w1, _ = _front_half_inputs(37, h, w, 1.0)
    par = gu.syn.uniform(37, 'pd', (3, h, w), 0.05, 1.0)
    ref = F.conv2d(D(x), D(wt), D(b), padding=1) * D(gam).view(1, 64, 1, 1)
    for j in range(3):
        ref = ref + D(par[j]).view(1, 1, h, w) * F.conv2d(D(x), D(w1[j]))
    xs = ops.nchw_to_nhwc(G(x))[0]
    pg = G(par)
    out = ops.conv3x3_f16x3([xs], [ops.pack_conv3x3(G(wt))], bias=G(b), gamma=G(gam),
                            packed_w1x1=ops.pack_conv1x1([G(v) for v in w1]), par=pg, par_flags=ops.par_tile_flags(pg), act=0)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_CONV / 2


def test_f16x3_is_run_to_run_deterministic():
    from pnp_vcve_amd import ops
    h, w = 180, 320
    x, wt, b, gam, w1, par = _front_half_inputs(38, h, w, 1.0 / 255.0)
    xs = ops.nchw_to_nhwc(G(x))[0]
    pg = G(par)
    args = dict(bias=G(b), gamma=G(gam), packed_w1x1=ops.pack_conv1x1([G(v) for v in w1]), par=pg,
                par_flags=ops.par_tile_flags(pg), act=1)
    pw = [ops.pack_conv3x3(G(wt))]
    first = ops.conv3x3_f16x3([xs], pw, **args).clone()
    for _ in range(5):
        torch.randn(1 << 22, device=dev()).sin_()
        assert torch.equal(ops.conv3x3_f16x3([xs], pw, **args), first)


# ---------------------------------------------------------------- whole path
def _run(case, precision):
    import pnp_vcve_amd as P
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
    m = m.to(dev()).eval()
    m.precision = precision
    a = {k: G(v) for k, v in clip.items()}
    with torch.no_grad():
        out = m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])
    return out.cpu(), clip


@pytest.mark.parametrize('case', gu.GEN_CASES, ids=[c['name'] for c in gu.GEN_CASES])
def test_f16x3_generator_meets_the_fp32_gate_on_every_golden_case(case, record_property):
    """Every golden case of the imported reference (tests/golden/), at the exact-fp32 path's own tolerance."""
    ref = torch.from_numpy(gu.load_golden(case['name'])['out'])
    out3, clip = _run(case, 'f16x3')
    out32, _ = _run(case, 'fp32')
    d3 = float((out3 - ref).abs().max())
    d32 = float((out32 - ref).abs().max())
    record_property('f16x3_maxabs', d3)
    print(f"{case['name']}: f16x3 path max|d| = {d3:.3e} (fp32 path {d32:.3e})")
    assert d3 < TOL_PATH
    assert not torch.equal(out3, out32)           # the split kernels really ran
    if ref.shape[-2:] == clip['lq'].shape[-2:]:   # north_star's statistic
        gt = (torch.from_numpy(clip['lq']) + 0.02 * torch.from_numpy(
            gu.syn.uniform(5, 'gt' + case['name'], clip['lq'].shape, -1, 1))).clamp(0, 1)
        assert abs(cpu_ref.clip_psnr(ref, gt) - cpu_ref.clip_psnr(out3, gt)) < 1e-3


def test_f16x3_precision_switch_resizes_buffers_and_round_trips():
    import pnp_vcve_amd as P
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    n32 = m._packed_floats
    assert m.precision == 'fp32'
    m.precision = 'f16x3'
    assert m.precision == 'f16x3' and m.fp16_enabled is False and m._packed_floats == 2 * n32
    m.precision = 'fp16'
    assert m.fp16_enabled is True and m._packed_floats == n32 + n32 // 2
    m.precision = 'fp32'
    assert m._packed_floats == n32
    with pytest.raises(ValueError):
        m.precision = 'bf16'


@pytest.mark.parametrize('deform', ['basic', 'fvc'])
def test_f16x3_with_the_deformable_aligners(deform):
    """deform='basic' | 'fvc': the offset convs and the DCN contraction stay on the exact fp32 kernels, the BAE blocks split."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2, deform=deform)
    import pnp_vcve_amd as P
    sd_np = gu.syn.make_state_dict(cfg, seed=181, par_gain=10.0)
    clip = gu.syn.make_clip(seed=182, n=1, t=3, h=72, w=104, slices='IBBBP', block=4, par_classes=3)
    outs = {}
    for prec in ('fp32', 'f16x3'):
        m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
        m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
        m = m.to(dev()).eval()
        m.precision = prec
        a = {k: G(v) for k, v in clip.items()}
        with torch.no_grad():
            outs[prec] = m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions']).cpu()
    d = float((outs['fp32'] - outs['f16x3']).abs().max())
    print(f'deform {deform}: f16x3 vs fp32 max|d| {d:.3e}')
    assert 0 < d < TOL_PATH


def test_f16x3_full_width_720p_tracks_fp32():
    """BASELINE configs[2]'s frame size (T = 2, 4 blocks): 7200 tiles per launch, every XCD band, ragged nothing."""
    import pnp_vcve_amd as P
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=4)
    sd_np = gu.syn.make_state_dict(cfg, seed=191, par_gain=10.0)
    clip = gu.syn.make_clip(seed=192, n=1, t=2, h=720, w=1280, slices='IBBBP', block=8, par_classes=3)
    outs = {}
    for prec in ('fp32', 'f16x3'):
        m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
        m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
        m = m.to(dev()).eval()
        m.precision = prec
        a = {k: G(v) for k, v in clip.items()}
        with torch.no_grad():
            outs[prec] = m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions']).cpu()
    d = float((outs['fp32'] - outs['f16x3']).abs().max())
    print(f'720p: f16x3 vs fp32 max|d| {d:.3e}')
    assert 0 < d < TOL_PATH


def test_f16x3_operands_saturate_instead_of_overflowing():
    """include/pnpvcve.h: a value beyond fp16's range saturates at +-65504 AS A WHOLE (hi = 65504, lo = 0) instead of turning into
    inf / NaN -- in the halo split, in the weight image and in the partition branch's re-split of par_j * x (the
    remainder of the re-split was converted without a clamp and overflowed to inf -> NaN through the MFMA)."""
    from pnp_vcve_amd import ops
    h, w = 32, 48
    x, wt, b, gam, w1, par = _front_half_inputs(38, h, w, 1.0)
    x = x.copy()
    x[0, 3, 9, 11] = 1e6                      # the activation itself is out of range
    x[0, 5, 20, 30] = 3e4                     # in range, but par_j * x below is not
    par = par * np.float32(8.0)
    xs = ops.nchw_to_nhwc(G(x))[0]
    pg = G(par)
    out = ops.conv3x3_f16x3([xs], [ops.pack_conv3x3(G(wt))], bias=G(b), gamma=G(gam),
                            packed_w1x1=ops.pack_conv1x1([G(v) for v in w1]), par=pg, par_flags=ops.par_tile_flags(pg), act=0)
    assert torch.isfinite(out).all()
    # the same arithmetic in fp64 on the values the kernel is specified to see: x and par_j * x clamped to +-65504
    xc = np.clip(x, -65504.0, 65504.0)
    ref = F.conv2d(D(xc), D(wt), D(b), padding=1) * D(gam).view(1, 64, 1, 1)
    for j in range(3):
        pxc = torch.clamp(D(par[j]).view(1, 1, h, w) * D(xc), -65504.0, 65504.0)
        ref = ref + F.conv2d(pxc, D(w1[j]))
    got = ops.nhwc_to_nchw(out.unsqueeze(0)).cpu().double()
    assert float((got - ref).abs().max()) < 1e-6 * float(ref.abs().max())
    # an out-of-range weight saturates the same way
    wbig = wt.copy()
    wbig[7, 3, 1, 1] = -2e5
    o2 = ops.conv3x3_f16x3([ops.nchw_to_nhwc(G(np.clip(x, -1, 1)))[0]], [ops.pack_conv3x3(G(wbig))], bias=G(b))
    r2 = F.conv2d(D(np.clip(x, -1, 1)), D(np.clip(wbig, -65504.0, 65504.0)), D(b), padding=1)
    assert torch.isfinite(o2).all() and float((ops.nhwc_to_nchw(o2.unsqueeze(0)).cpu().double() - r2).abs().max()) < 1e-6 * float(r2.abs().max())


@pytest.mark.parametrize('hw', [(40, 56), (264, 272)])
def test_f16x3_front_half_fast_path_on_binary_partition_maps(hw):
    """Tiles whose partition values are all 0 or exactly 1/255 (flag bits 3..5; what the reference's loader writes,
    loading_ipb.py: one-hot uint8 planes / 255.) contract the 1x1 branches with weight images scaled by 1/255 at pack time and a MASKED
    A operand instead of re-splitting par_j(pixel) * x per fragment.  Against fp64 at the conv tolerance; against the general path to
    the last bits; and a frame in which ONE tile carries another value must still be right everywhere (that tile falls back)."""
    from pnp_vcve_amd import ops
    h, w = hw
    x, wt, b, gam, w1, par = _front_half_inputs(39, h, w, 1.0)
    par = (par * (np.float32(1.0) / np.float32(255.0))).astype(np.float32)        # one-hot / 255 per 8x8 block, like the loader
    assert set(np.unique(par).tolist()) == {0.0, float(np.float32(1.0) / np.float32(255.0))}
    w1 = [v * np.float32(40.0) for v in w1]                                       # make the branches count: sum |par * conv1x1| ~ 0.1

    def ref_of(pm):
        r = F.conv2d(D(x), D(wt), D(b), padding=1) * D(gam).view(1, 64, 1, 1)
        for j in range(3):
            r = r + D(pm[j]).view(1, 1, h, w) * F.conv2d(D(x), D(w1[j]))
        return F.relu(r)

    xs = ops.nchw_to_nhwc(G(x))[0]
    pw, p1 = ops.pack_conv3x3(G(wt)), ops.pack_conv1x1([G(v) for v in w1])

    def run(pm, scaled):
        pg = G(pm)
        fl = ops.par_tile_flags(pg)
        return ops.nhwc_to_nchw(ops.conv3x3_f16x3([xs], [pw], bias=G(b), gamma=G(gam), packed_w1x1=p1, par=pg, par_flags=fl, act=1,
                                                  scaled_w1x1=scaled).unsqueeze(0)), fl

    fast, fl = run(par, True)
    gen, _ = run(par, False)
    assert int(((fl >> 3) & 7).min()) == 7                                        # every tile qualifies
    ref = ref_of(par)
    assert float((ref - ref_of(par * 0)).abs().max()) > 1e-2                      # the branches are visible
    assert maxdiff(fast, ref) < TOL_CONV / 2 and maxdiff(gen, ref) < TOL_CONV / 2
    d = maxdiff(fast, gen.cpu())
    assert 0 < d < 2e-6                                                           # another rounding point, the same result
    # one tile with a different value (0.5 on a nonzero block of plane 1): its flag bit drops and it takes the general path
    pm = par.copy()
    ys, xs_ = np.nonzero(pm[1])
    pm[1, ys[0], xs_[0]] = 0.5
    mixed, fl2 = run(pm, True)
    assert int(((fl2 >> 3) & 7).min()) < 7 and int((((fl2 >> 3) & 7) == 7).sum()) >= fl2.numel() - 1
    assert maxdiff(mixed, ref_of(pm)) < TOL_CONV / 2


@pytest.mark.parametrize('scaled', [True, False])
def test_f16x3_front_half_is_bit_stable_under_unrelated_traffic(scaled):
    """Both branch forms give the same bits launch after launch while another kernel keeps the memory system busy.  r04: an
    intermediate build whose general re-split was float-VECTOR code (v_pk_*_f32 feeding the MFMA operands of the same dealt block)
    differed from itself by 7e-4 on 40 % of the pixels; the scalar form, built with -fno-slp-vectorize, is what this pins."""
    from pnp_vcve_amd import ops
    h, w = 180, 320
    x, wt, b, gam, w1, par = _front_half_inputs(38, h, w, 1.0)
    par = (par * (np.float32(1.0) / np.float32(255.0))).astype(np.float32)
    xs, pg = ops.nchw_to_nhwc(G(x))[0], G(par)
    assert int(((ops.par_tile_flags(pg) >> 3) & 7).min()) == 7                   # scaled: every tile takes the masked-operand form
    args = dict(bias=G(b), gamma=G(gam), packed_w1x1=ops.pack_conv1x1([G(v) for v in w1]), par=pg, par_flags=ops.par_tile_flags(pg),
                act=1, scaled_w1x1=scaled)
    pw = [ops.pack_conv3x3(G(wt))]
    first = ops.conv3x3_f16x3([xs], pw, **args).clone()
    for _ in range(6):
        torch.randn(1 << 22, device='cuda').sin_()
        assert torch.equal(ops.conv3x3_f16x3([xs], pw, **args), first)


@pytest.mark.parametrize('hw', [(264, 272), (720, 1280)], ids=lambda s: '%dx%d' % s)
def test_f16x3_tile_queue_hands_out_the_same_tiles_and_resets_itself(hw):
    """PNP_OPT_TILE_QUEUE / pnp_conv3x3_f16x3_ex(tile_queue): blocks draw their tiles from per-XCD ticket counters instead of
    walking a static share (561 and 7200 tiles on 512 blocks here).  Which block computes a tile does not enter its value: the
    result is bit-identical; the last block to leave zeroes the 9 counters, so back-to-back launches need no host reset."""
    from pnp_vcve_amd import ops
    h, w = hw
    x, wt, b, gam, w1, par = _front_half_inputs(40, h, w, 1.0)
    par = (par * (np.float32(1.0) / np.float32(255.0))).astype(np.float32)
    xs, pg = ops.nchw_to_nhwc(G(x))[0], G(par)
    res = torch.randn(h, w, 64, device=dev(), generator=torch.Generator(device=dev()).manual_seed(5))
    pw = [ops.pack_conv3x3(G(wt))]
    q = torch.zeros(16, dtype=torch.int32, device=dev())
    front = dict(bias=G(b), gamma=G(gam), packed_w1x1=ops.pack_conv1x1([G(v) for v in w1]), par=pg, par_flags=ops.par_tile_flags(pg),
                 act=1, scaled_w1x1=True)
    back = dict(bias=G(b), residual=res)
    for kw in (front, back):
        ref = ops.conv3x3_f16x3([xs], pw, **kw)
        for _ in range(3):
            assert torch.equal(ops.conv3x3_f16x3([xs], pw, tile_queue=q, **kw), ref)
            assert int(q.abs().sum()) == 0
    # several sources in one launch (the MS instantiation walks its passes inside a tile)
    ref = ops.conv3x3_f16x3([xs, res], pw + pw, bias=G(b), act=2)
    assert torch.equal(ops.conv3x3_f16x3([xs, res], pw + pw, bias=G(b), act=2, tile_queue=q), ref) and int(q.abs().sum()) == 0
    with pytest.raises(TypeError):
        ops.conv3x3_f16x3([xs], pw, tile_queue=torch.zeros(16, device=dev()))

"""GPU tests of the opt-in fp16-operand conv path (BASELINE configs[4]; csrc/conv_f16.hip), through the C ABI.

Two gates:
  * kernel exactness: with the SAME fp16-rounded operands the kernel must match an fp64 host contraction to
    fp32-accumulation accuracy -- this is what catches layout / indexing bugs;
  * path accuracy: the whole generator with fp16_enabled against the fp32 golden output of the imported
    reference -- the rounding cost of the precision choice, reported and bounded (SURVEY.md section 7 step 7:
    "tolerance reported, not gated at 1e-3"), plus the PSNR statistic of north_star.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
from oracle import cpu_ref

pytestmark = pytest.mark.gpu

TOL_F16_KERNEL = 3e-5     # fp32 accumulation of <= 2300 fp16 x fp16 products, |sum| ~ 1
TOL_F16_PATH = 2e-2       # enhanced frames in [0,1]: fp16 operand rounding through 2 x 8 blocks x 7 frames
TOL_F16_PSNR_DB = 5e-2    # |PSNR(fp16 path) - PSNR(fp32 reference)| against the same ground truth


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def r16(a):
    """round to fp16 (saturating like the kernel), back to fp64"""
    return torch.as_tensor(a).float().clamp(-65504, 65504).half().double()


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


@pytest.mark.parametrize('hw', [(16, 16), (24, 40), (37, 53), (64, 64), (128, 256), (180, 320)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_f16_conv_single_source_exact_on_rounded_operands(hw, act):
    from pnp_vcve_amd import ops
    h, w = hw
    x = gu.syn.uniform(17, f'x{h}x{w}', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(17, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(17, 'b', (64,), -0.1, 0.1)
    res = gu.syn.uniform(17, f'r{h}x{w}', (1, 64, h, w), -1, 1)
    ref = F.conv2d(r16(x), r16(wt), torch.from_numpy(b).double(), padding=1)
    ref = [ref, F.relu(ref), F.leaky_relu(ref, 0.1)][act] + torch.from_numpy(res).double()
    xs = ops.nchw_to_nhwc(G(x))[0]
    rs = ops.nchw_to_nhwc(G(res))[0]
    out = ops.conv3x3([xs], [ops.f16_image(ops.pack_conv3x3(G(wt)))], bias=G(b), residual=rs, act=act, fp16=True)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_F16_KERNEL


def test_f16_conv_identity_weights_localise_layout_bugs():
    from pnp_vcve_amd import ops
    h, w = 24, 40
    x = gu.syn.uniform(18, 'x', (1, 64, h, w), -1, 1)
    perm = np.roll(np.arange(64), 5)
    for (ky, kx) in [(1, 1), (0, 0), (2, 1), (1, 2)]:
        wt = np.zeros((64, 64, 3, 3), np.float32)
        wt[np.arange(64), perm, ky, kx] = 1.0
        ref = F.conv2d(r16(x), torch.from_numpy(wt).double(), padding=1)
        out = ops.conv3x3([ops.nchw_to_nhwc(G(x))[0]], [ops.f16_image(ops.pack_conv3x3(G(wt)))], fp16=True)
        assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) == 0.0, (ky, kx)


@pytest.mark.parametrize('nwide', [1, 2, 3])
@pytest.mark.parametrize('with_lr', [False, True])
def test_f16_conv_virtual_concat_is_a_launch_chain(nwide, with_lr):
    """input_conv over [lr(3), wide...]: one launch per 64-channel source, partial sums through `out`."""
    from pnp_vcve_amd import ops
    h, w = 40, 56
    cin = (3 if with_lr else 0) + 64 * nwide
    lr = gu.syn.uniform(19, 'lr', (1, 3, h, w), 0, 1)
    wides = [gu.syn.uniform(19, f's{j}', (1, 64, h, w), -1, 1) for j in range(nwide)]
    wt = gu.syn.uniform(19, f'w{cin}', (64, cin, 3, 3), -0.05, 0.05)
    b = gu.syn.uniform(19, 'b', (64,), -0.1, 0.1)
    cat = np.concatenate(([lr] if with_lr else []) + wides, axis=1)
    ref = F.leaky_relu(F.conv2d(r16(cat), r16(wt), torch.from_numpy(b).double(), padding=1), 0.1)
    lr4 = np.concatenate([lr, np.zeros((1, 1, h, w), np.float32)], axis=1)
    srcs = ([ops.nchw_to_nhwc(G(lr4))[0]] if with_lr else []) + [ops.nchw_to_nhwc(G(s))[0] for s in wides]
    wg = G(wt)
    c0 = 3 if with_lr else 0
    packed = ([ops.pack_conv3x3(wg, 0, 3)] if with_lr else []) + \
        [ops.pack_conv3x3(wg, c0 + 64 * j, 64) for j in range(nwide)]
    out = ops.conv3x3(srcs, [ops.f16_image(p) for p in packed], bias=G(b), act=2, fp16=True)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_F16_KERNEL


def test_f16_conv_unsupported_shapes_are_refused_not_silently_rerouted():
    from pnp_vcve_amd import ops
    h, w = 16, 16
    lr4 = torch.zeros(h, w, 4, device=dev())
    wt = gu.syn.uniform(20, 'w', (64, 3, 3, 3), -0.05, 0.05)
    with pytest.raises(RuntimeError):
        ops.conv3x3([lr4], [ops.f16_image(ops.pack_conv3x3(G(wt), 0, 3))], fp16=True)


@pytest.mark.parametrize('hw', [(32, 48), (72, 88)])
def test_f16_bae_front_half_exact_on_rounded_operands(hw):
    """relu(gamma * (conv3x3(x) + b) + sum_j par_j * conv1x1_j(x)) with the kernel's rounding points:
    x, W -> fp16; (x * par_j) -> fp16 (the 1x1 branches are a K extension whose A operand is scaled)."""
    from pnp_vcve_amd import ops
    h, w = hw
    x = gu.syn.uniform(21, f'x{h}', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(21, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(21, 'b', (64,), -0.1, 0.1)
    gam = gu.syn.uniform(21, 'g', (64,), 0.5, 1.5)
    w1 = [gu.syn.uniform(21, f'w1_{j}', (64, 64, 1, 1), -0.1, 0.1) for j in range(3)]
    cls = (gu.syn.uniform(21, f'c{h}', (h // 8, w // 8), 0, 3).astype(np.int64)).clip(0, 2)
    par = np.zeros((3, h, w), np.float32)
    for j in range(3):
        par[j] = np.kron((cls == j).astype(np.float32), np.ones((8, 8), np.float32))
    par *= np.float32(1.0 / 255.0) * 200.0      # any per-pixel scale; not exactly representable in fp16
    ref = (F.conv2d(r16(x), r16(wt), torch.from_numpy(b).double(), padding=1)
           * torch.from_numpy(gam).double().view(1, 64, 1, 1))
    for j in range(3):
        xs_j = (torch.from_numpy(x).half() * torch.from_numpy(par[j]).half().view(1, 1, h, w)).double()
        ref = ref + F.conv2d(xs_j, r16(w1[j]))
    ref = F.relu(ref)
    xs = ops.nchw_to_nhwc(G(x))[0]
    out = ops.conv3x3([xs], [ops.f16_image(ops.pack_conv3x3(G(wt)))], bias=G(b), gamma=G(gam),
                      packed_w1x1=ops.f16_image(ops.pack_conv1x1([G(v) for v in w1])), par=G(par), act=1, fp16=True)
    assert maxdiff(ops.nhwc_to_nchw(out.unsqueeze(0)), ref) < TOL_F16_KERNEL


def test_f16_operands_saturate_instead_of_overflowing():
    from pnp_vcve_amd import ops
    h, w = 16, 32
    x = np.zeros((1, 64, h, w), np.float32)
    x[0, 3, 5, 7] = 1e6
    wt = np.zeros((64, 64, 3, 3), np.float32)
    wt[np.arange(64), np.arange(64), 1, 1] = 1.0
    out = ops.conv3x3([ops.nchw_to_nhwc(G(x))[0]], [ops.f16_image(ops.pack_conv3x3(G(wt)))], fp16=True)
    o = ops.nhwc_to_nchw(out.unsqueeze(0)).cpu()
    assert torch.isfinite(o).all() and float(o[0, 3, 5, 7]) == 65504.0


# ---------------------------------------------------------------- whole path
def _run(case, fp16, options=()):
    import pnp_vcve_amd as P
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
    m = m.to(dev()).eval()
    m.fp16_enabled = fp16
    for opt, val in options:                      # pnp_generator_set_option (per-generator state)
        m.set_option(opt, val)
    a = {k: G(v) for k, v in clip.items()}
    with torch.no_grad():
        out = m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])
    return out.cpu(), clip


F16_GEN_CASES = [c for c in gu.GEN_CASES if c['name'] in (
    'gen_t7_128x128', 'gen_parfloat_72x88', 'gen_vsr_64x64', 'gen_n2_mixed_64x64', 'gen_two_layer_64x64',
    'gen_channel_last_64x64')]
assert len(F16_GEN_CASES) == 6


@pytest.mark.parametrize('case', F16_GEN_CASES, ids=[c['name'] for c in F16_GEN_CASES])
def test_f16_generator_close_to_fp32_reference(case, record_property):
    ref = torch.from_numpy(gu.load_golden(case['name'])['out'])
    out16, clip = _run(case, True)
    out32, _ = _run(case, False)
    d16 = float((out16 - ref).abs().max())
    d32 = float((out32 - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    record_property('fp16_maxabs', d16)
    print(f"{case['name']}: fp16 path max|d| = {d16:.3e} (fp32 path {d32:.3e}), |ref|max = {scale:.3g}")
    assert d32 / scale < 1e-3                     # switching back restores the exact path
    assert d16 / scale < TOL_F16_PATH
    assert d16 > d32                              # the fp16 kernels really ran
    # north_star's statistic: PSNR against a synthetic ground truth, both paths
    if ref.shape[-2:] == clip['lq'].shape[-2:]:
        gt = torch.from_numpy(clip['lq']) + 0.02 * torch.from_numpy(
            gu.syn.uniform(5, 'gt' + case['name'], clip['lq'].shape, -1, 1))
        p_ref = cpu_ref.clip_psnr(ref, gt.clamp(0, 1))
        p_16 = cpu_ref.clip_psnr(out16, gt.clamp(0, 1))
        print(f"  PSNR ref {p_ref:.4f} dB, fp16 {p_16:.4f} dB")
        assert abs(p_ref - p_16) < TOL_F16_PSNR_DB


def test_f16_precision_switch_resizes_buffers_and_round_trips():
    import pnp_vcve_amd as P
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    n32 = m._packed_floats
    assert m.fp16_enabled is False
    m.fp16_enabled = True
    assert m.fp16_enabled is True and m._packed_floats == n32 + n32 // 2
    m.fp16_enabled = False
    assert m._packed_floats == n32


def test_f16_intermediate_maps_are_bit_identical_to_fp32_storage():
    """The BAE-block intermediate is only ever an MFMA A operand: writing it rounded (fp16 map) instead of rounding it
    in its reader must not change a single bit of the clip."""
    from pnp_vcve_amd import _native
    for name in ('gen_parfloat_72x88', 'gen_channel_last_64x64', 'gen_two_layer_64x64', 'gen_vsr_64x64'):
        case = [c for c in gu.GEN_CASES if c['name'] == name][0]
        a, _ = _run(case, True, options=[(_native.OPT_F16_MAPS, 0)])
        b, _ = _run(case, True)
        assert torch.equal(a, b), name


def test_f16_randomised_configs_track_the_fp32_path():
    """Seeded draws over the constructor switches / shapes (incl. the x4 heads and both DCN aligners, whose offset
    convs also run on the fp16 kernels): the fp16 path must stay finite and close to the fp32 path of the same build."""
    import pnp_vcve_amd as P
    rng = np.random.RandomState(77)
    worst = 0.0
    for trial in range(10):
        with_bias = bool(rng.randint(2))
        cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG,
                   num_blocks=int(rng.randint(1, 4)), num_experts=int(rng.choice([2, 6])),
                   with_cat=bool(rng.randint(2)), align_key=bool(rng.randint(2)), vsr=bool(rng.randint(3) == 0),
                   expert_softmax=True, with_bias=with_bias, with_se=with_bias and bool(rng.randint(2)),
                   use_base_qp=True, one_layer=bool(rng.randint(2)), channel_first=bool(rng.randint(2)),
                   deform=str(rng.choice(['vos', 'vos', 'basic', 'fvc'])))
        n, t = int(rng.choice([1, 2])), int(rng.randint(1, 5))
        h, w = 64 + 4 * int(rng.randint(0, 6)), 64 + 4 * int(rng.randint(0, 10))
        sd_np = gu.syn.make_state_dict(cfg, seed=3000 + trial, par_gain=1.0)
        clip = gu.syn.make_clip(seed=4000 + trial, n=n, t=t, h=h, w=w, slices='IBBBP',
                                block=4, qp_mode='ipb', crf=[15, 35][:n] if n > 1 else 25)
        m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
        m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
        m = m.to(dev()).eval()
        a = {k: G(v) for k, v in clip.items()}
        outs = []
        for f16 in (False, True):
            m.fp16_enabled = f16
            with torch.no_grad():
                outs.append(m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions']).cpu())
        assert torch.isfinite(outs[1]).all(), (trial, cfg)
        scale = max(1.0, float(outs[0].abs().max()))
        d = float((outs[0] - outs[1]).abs().max()) / scale
        worst = max(worst, d)
        print(f'trial {trial}: deform={cfg["deform"]} vsr={cfg["vsr"]} {n}x{t}x{h}x{w}  max|fp16-fp32|/scale = {d:.3e}')
        assert 0.0 < d < TOL_F16_PATH, (trial, cfg, (n, t, h, w), d, scale)
    print('worst', worst)


def test_f16_full_width_720p_tracks_fp32():
    """Full-size frame (all 7200 tiles incl. every image edge, the persistent fp16 kernels, the RGB head): the fp16 path
    against the fp32 path of the same build on a 2-frame 720p clip -- PSNR statistic and max-abs."""
    import pnp_vcve_amd as P
    from pnp_vcve_amd import ops
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    sd_np = gu.syn.make_state_dict(cfg, seed=111, par_gain=10.0)
    clip = gu.syn.make_clip(seed=112, n=1, t=2, h=720, w=1280, slices='IBBBP')
    m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
    m = m.to(dev()).eval()
    a = {k: G(v) for k, v in clip.items()}
    outs = []
    for f16 in (False, True):
        m.fp16_enabled = f16
        with torch.no_grad():
            outs.append(m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions']))
    d = float((outs[0] - outs[1]).abs().max())
    p32 = float(ops.psnr_frames(outs[0][0], a['gt'][0]).mean())
    p16 = float(ops.psnr_frames(outs[1][0], a['gt'][0]).mean())
    print(f'720p: max|fp16 - fp32| = {d:.3e}, PSNR {p32:.5f} vs {p16:.5f} dB')
    assert 0.0 < d < TOL_F16_PATH and abs(p32 - p16) < 1e-3


@pytest.mark.parametrize('hw', [(180, 320), (128, 128), (68, 100)], ids=lambda s: '%dx%d' % s)
def test_small_frame_f16_kernel_is_bit_identical_to_the_persistent_one(hw):
    """frames with < 1024 tiles run the 64 -> 64 convs on conv3x3_f16_small_kernel (one tile per block, weight chunks
    streamed through a 3-slot ring, three blocks per CU): same k order, same fp32 accumulation, same epilogue arithmetic as
    the persistent fp16 kernel -- the clip must not change by a single bit (PNP_OPT_SMALL_F16 0 / 1), with fp16 and with
    fp32 intermediate maps, ragged tiles included."""
    import pnp_vcve_amd as P
    from pnp_vcve_amd import _native
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=3)
    sd_np = gu.syn.make_state_dict(cfg, seed=131, par_gain=10.0)
    clip = gu.syn.make_clip(seed=132, n=1, t=3, h=hw[0], w=hw[1], slices='IBBBP', block=4, par_classes=3)
    m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
    m = m.to(dev()).eval()
    m.fp16_enabled = True
    a = {k: G(v) for k, v in clip.items()}

    def run():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])

    for maps16 in (1, 0):
        m.set_option(_native.OPT_F16_MAPS, maps16)
        m.set_option(_native.OPT_SMALL_F16, 0)
        ref = run().clone()
        m.set_option(_native.OPT_SMALL_F16, 1)
        out = run()
        assert torch.isfinite(out).all() and torch.equal(out, ref), (hw, maps16, float((out - ref).abs().max()))



# ---------------------------------------------------------------- r03: fp16 mirrors, branch skipping, one-launch input conv
def _h16(t):
    """the kernels' rounding of an fp32 value to an fp16 MFMA operand: saturate, round to nearest even"""
    return t.clamp(-65504, 65504).half()


def _block_par(seed, h, w, scale=200.0 / 255.0, empty_every=3):
    """one-hot partition planes per 8x8 block, every `empty_every`-th block row without any record"""
    cls = gu.syn.randint(seed, f'cls{h}x{w}', ((h + 7) // 8, (w + 7) // 8), 0, 2)
    par = np.zeros((3, h, w), np.float32)
    for j in range(3):
        blk = (cls == j).astype(np.float32)
        blk[::empty_every] = 0.0
        par[j] = np.kron(blk, np.ones((8, 8), np.float32))[:h, :w]
    return par * np.float32(scale)


# small G=1 (ragged), small G=1, small G=2, then >= 1024 tiles (the resident-weight kernel), the last one ragged in both directions
SIZES_F16 = [(40, 56), (180, 320), (128, 512), (256, 512), (264, 520)]


@pytest.mark.parametrize('hw', SIZES_F16, ids=lambda s: '%dx%d' % s)
@pytest.mark.parametrize('with_par', [False, True], ids=['plain', 'par'])
def test_f16_mirror_and_fp16_sources_are_bit_identical(hw, with_par):
    """(1) the fp16 mirror a producer writes next to its fp32 output is exactly the consumer's operand rounding of that output;
    (2) reading a source through its fp16 mirror instead of rounding the fp32 map on the fly changes no bit -- for fp32 output,
    fp16 output and fp32 + mirror output, with and without the partition branches."""
    from pnp_vcve_amd import ops
    h, w = hw
    x = ops.nchw_to_nhwc(G(gu.syn.uniform(51, f'x{h}x{w}', (1, 64, h, w), -1, 1)))[0]
    res = ops.nchw_to_nhwc(G(gu.syn.uniform(51, f'r{h}x{w}', (1, 64, h, w), -1, 1)))[0]
    wt = ops.f16_image(ops.pack_conv3x3(G(gu.syn.uniform(51, 'w', (64, 64, 3, 3), -0.06, 0.06))))
    b, gam = G(gu.syn.uniform(51, 'b', (64,), -0.1, 0.1)), G(gu.syn.uniform(51, 'g', (64,), 0.5, 1.5))
    kw = dict(bias=b, gamma=gam, act=1)
    if with_par:
        kw.update(packed_w1x1=ops.f16_image(ops.pack_conv1x1([G(gu.syn.uniform(51, f'w1_{j}', (64, 64, 1, 1), -0.1, 0.1))
                                                              for j in range(3)])), par=G(_block_par(51, h, w)))
    base = ops.conv3x3_f16_maps([x], [wt], residual=res, **kw)
    assert torch.equal(base, ops.conv3x3([x], [wt], residual=res, fp16=True, **kw))          # the r02 entry point
    out, m16 = ops.conv3x3_f16_maps([x], [wt], residual=res, mirror=True, **kw)
    assert torch.equal(out, base) and torch.equal(m16, _h16(base))
    x16 = _h16(x)
    assert torch.equal(ops.conv3x3_f16_maps([x16], [wt], residual=res, **kw), base)
    out, m16 = ops.conv3x3_f16_maps([x16], [wt], residual=res, mirror=True, **kw)
    assert torch.equal(out, base) and torch.equal(m16, _h16(base))
    o16 = ops.conv3x3_f16_maps([x], [wt], out_f16=True, **kw)                                   # fp16 output: no residual
    assert torch.equal(o16, _h16(ops.conv3x3_f16_maps([x], [wt], **kw)))
    assert torch.equal(ops.conv3x3_f16_maps([x16], [wt], out_f16=True, **kw), o16)              # fp16 in, fp16 out (new)


@pytest.mark.parametrize('hw', SIZES_F16 + [(720, 1280)], ids=lambda s: '%dx%d' % s)
def test_f16_partition_branch_skipping_is_value_identical(hw):
    """fp16 kernels: a 1x1 partition branch whose plane is zero over a whole 8x16 tile adds exact zeros -- skipping its four
    k-steps (pnp_par_tile_flags_f32) leaves every output value unchanged, for fp32 and for fp16 output, on maps where tiles
    need 0, 1, 2 or 3 branches."""
    from pnp_vcve_amd import ops
    h, w = hw
    x = torch.randn(h, w, 64, device=dev(), generator=torch.Generator(device=dev()).manual_seed(3))
    wt = ops.f16_image(ops.pack_conv3x3(G(gu.syn.uniform(52, 'w', (64, 64, 3, 3), -0.06, 0.06))))
    w1 = ops.f16_image(ops.pack_conv1x1([G(gu.syn.uniform(52, f'w1_{j}', (64, 64, 1, 1), -0.1, 0.1)) for j in range(3)]))
    par = _block_par(52, h, w)
    par[:, : h // 4] = 0.0                                   # a band without any record (an I-frame-like region)
    par[:, h // 2: h // 2 + 8, : w // 2] = 0.3               # tiles that need all three branches, non-binary values
    par = G(par)
    flags = ops.par_tile_flags(par)
    # (bits 0..2: the plane is nonzero somewhere in the tile; bits 3..5: it is binary-valued there, the split kernel's fast path)
    assert set(int(v) for v in (flags & 7).unique().tolist()) >= {0, 7} and flags.numel() == ((h + 7) // 8) * ((w + 15) // 16)
    kw = dict(bias=G(gu.syn.uniform(52, 'b', (64,), -0.1, 0.1)), gamma=G(gu.syn.uniform(52, 'g', (64,), 0.5, 1.5)),
              packed_w1x1=w1, par=par, act=1)
    for src in (x, _h16(x)):                                  # fp32 source rounded on the fly / fp16 mirror
        for extra in (dict(), dict(out_f16=True)):
            a = ops.conv3x3_f16_maps([src], [wt], **kw, **extra)
            b = ops.conv3x3_f16_maps([src], [wt], par_flags=flags, **kw, **extra)
            assert torch.equal(a, b), (hw, src.dtype, extra, float((a.float() - b.float()).abs().max()))
    a = ops.conv3x3_f16_maps([x], [wt], **kw)
    # and the flags are honoured at all: a map that lies about a live plane changes the result
    lie = torch.zeros_like(flags)
    assert not torch.equal(ops.conv3x3_f16_maps([x], [wt], par_flags=lie, **kw), a)


@pytest.mark.parametrize('hw', [(40, 56), (180, 320), (256, 512)], ids=lambda s: '%dx%d' % s)
@pytest.mark.parametrize('nwide,with_lr', [(1, True), (2, False), (2, True), (3, False), (3, True)],     # (1, False) is the ordinary
                         ids=['rgb+1', '2', 'rgb+2', '3', 'rgb+3'])                                     # single-source kernel
def test_f16_input_conv_in_one_launch_equals_the_launch_chain(hw, nwide, with_lr):
    """conv3x3_f16_multi_kernel (all wide sources read through fp16 mirrors, one launch) against the chain of single-source
    launches through fp32 partial sums: same MFMA chains per source, folded in the chain's order -- bit-identical, and so is
    the fp16 mirror of the result."""
    from pnp_vcve_amd import ops
    h, w = hw
    cin = (3 if with_lr else 0) + 64 * nwide
    lr = gu.syn.uniform(53, 'lr', (1, 3, h, w), 0, 1)
    wides = [ops.nchw_to_nhwc(G(gu.syn.uniform(53, f's{j}_{h}', (1, 64, h, w), -1, 1)))[0] for j in range(nwide)]
    wg = G(gu.syn.uniform(53, f'w{cin}', (64, cin, 3, 3), -0.05, 0.05))
    b = G(gu.syn.uniform(53, 'b', (64,), -0.1, 0.1))
    lr4 = ops.nchw_to_nhwc(G(np.concatenate([lr, np.zeros((1, 1, h, w), np.float32)], axis=1)))[0]
    c0 = 3 if with_lr else 0
    packed = ([ops.f16_image(ops.pack_conv3x3(wg, 0, 3))] if with_lr else []) + \
        [ops.f16_image(ops.pack_conv3x3(wg, c0 + 64 * j, 64)) for j in range(nwide)]
    head = [lr4] if with_lr else []
    chain = ops.conv3x3_f16_maps(head + wides, packed, bias=b, act=2, chain=True)
    assert torch.equal(chain, ops.conv3x3(head + wides, packed, bias=b, act=2, fp16=True))
    one, m16 = ops.conv3x3_f16_maps(head + [_h16(s) for s in wides], packed, bias=b, act=2, mirror=True)
    assert torch.equal(one, chain), float((one - chain).abs().max())
    assert torch.equal(m16, _h16(chain))
    assert torch.equal(ops.conv3x3_f16_maps(head + [_h16(s) for s in wides], packed, bias=b, act=2), chain)


def test_warp_fp16_output_is_the_rounded_fp32_warp():
    from pnp_vcve_amd import ops
    h, w = 72, 120
    feat = torch.randn(h, w, 64, device=dev()) * 3
    feat[5, 7, 3] = 1e6                                       # saturates instead of overflowing
    blk = (torch.randint(-32, 33, (2, h // 8, w // 8), device=dev()).float() / 4.0)
    fl = blk.repeat_interleave(8, 1).repeat_interleave(8, 2)
    a = ops.mv_warp_nhwc(feat, fl[0].contiguous(), fl[1].contiguous())
    b = ops.mv_warp_nhwc_f16(feat, fl[0].contiguous(), fl[1].contiguous())
    assert b.dtype == torch.float16 and torch.equal(b, _h16(a)) and torch.isfinite(b).all()


def _gen_model(cfg, sd_np):
    import pnp_vcve_amd as P
    m = P.build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict(cpu_ref.to_torch_state(sd_np), strict=True)
    return m.to(dev()).eval()


MIRROR_CASES = [
    ('default_180x320', dict(), dict(n=1, t=4, h=180, w=320)),
    ('default_t1_64x96', dict(), dict(n=1, t=1, h=64, w=96)),
    ('nocat_noalign_72x88', dict(with_cat=False, align_key=False), dict(n=1, t=3, h=72, w=88)),
    ('channel_last_two_layer_n2', dict(channel_first=False, one_layer=False, num_blocks=2), dict(n=2, t=3, h=64, w=64)),
    ('vsr_64x80', dict(vsr=True, num_blocks=2), dict(n=1, t=2, h=64, w=80)),
    ('default_720p', dict(num_blocks=2), dict(n=1, t=3, h=720, w=1280)),
]


@pytest.mark.parametrize('name,over,shape', MIRROR_CASES, ids=[c[0] for c in MIRROR_CASES])
def test_f16_mirrors_and_one_launch_input_conv_leave_the_clip_bit_identical(name, over, shape):
    """PNP_OPT_F16_MIRRORS (fp16 copies of x / the slots / the aligned key frame written by their producers, input conv in one
    launch, branch skipping on the fp16 kernels) against the r02 schedule (fp32 maps rounded by their readers, launch chain):
    not a bit of the enhanced clip changes; neither does it with the partition flags off."""
    from pnp_vcve_amd import _native
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, **over)
    sd_np = gu.syn.make_state_dict(cfg, seed=141, par_gain=10.0)
    crf = [15, 35] if shape['n'] == 2 else 25
    clip = gu.syn.make_clip(seed=142, slices='IBBBP', block=4, par_classes=3, qp_mode='ipb', crf=crf, **shape)
    m = _gen_model(cfg, sd_np)
    m.fp16_enabled = True
    a = {k: G(v) for k, v in clip.items()}

    def run():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])

    m.set_option(_native.OPT_F16_MIRRORS, 0)
    m.set_option(_native.OPT_PAR_SKIP, 0)
    ref = run().clone()
    m.set_option(_native.OPT_PAR_SKIP, 1)
    assert torch.equal(run(), ref), 'branch skipping alone'
    m.set_option(_native.OPT_F16_MIRRORS, 1)
    out = run()
    assert torch.isfinite(out).all() and torch.equal(out, ref), float((out - ref).abs().max())
    m.set_option(_native.OPT_F16_CHAIN_MIRRORS, 1)          # + the running map inside a branch (off by default)
    assert torch.equal(run(), ref), 'chain mirrors'
    m.set_option(_native.OPT_PAR_SKIP, 0)
    assert torch.equal(run(), ref)


@pytest.mark.parametrize('hw', [(180, 320), (720, 1280)], ids=lambda s: '%dx%d' % s)
def test_f16_kernels_are_run_to_run_deterministic(hw):
    """conv3x3_f16_small_kernel (180x320) / conv3x3_f16_kernel (720p) and the one-launch input conv: the same clip twice,
    and once more after other work has gone through the device, must agree bit for bit (guards the class of hazard seen in the
    fp16 DCN kernel, csrc/dcn.hip)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    sd_np = gu.syn.make_state_dict(cfg, seed=151, par_gain=10.0)
    clip = gu.syn.make_clip(seed=152, n=1, t=3, h=hw[0], w=hw[1], slices='IBBBP', block=4, par_classes=3)
    m = _gen_model(cfg, sd_np)
    m.fp16_enabled = True
    a = {k: G(v) for k, v in clip.items()}

    def run():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions']).clone()

    first = run()
    for rep in range(3):
        torch.randn(1 << 22, device=dev()).sin_()           # unrelated traffic between the runs
        assert torch.equal(run(), first), rep


def _pins():
    import json
    import os
    with open(os.path.join(gu.GOLDEN_DIR, 'f16_pins_r02.json')) as f:
        return json.load(f)['cases']


import f16_pin_cases as _pc  # noqa: E402


@pytest.mark.parametrize('case', _pc.PIN_CASES, ids=[c['name'] for c in _pc.PIN_CASES])
def test_f16_path_is_bit_identical_to_the_round_2_build(case):
    """The fp16 path of THIS build (fp16 mirrors, one-launch input conv, branch skipping, ...) against SHA-256 pins of the
    outputs the round-2 build produced on an MI355X (tests/f16_pin_cases.py, tools/gen_f16_pins.py): same k order and rounding
    points, hence the same bits (-0.0 canonicalised)."""
    import pnp_vcve_amd as P
    pin = _pins()[case['name']]
    out = _pc.run_case(case, gu.syn, P.build_backbone, torch)
    got = _pc.digest(out)
    assert got['shape'] == pin['shape']
    assert got['sha256'] == pin['sha256'], (case['name'], got['mean'], pin['mean'], got['first'], pin['first'])


@pytest.mark.parametrize('deform', ['basic', 'fvc'])
def test_f16_generator_with_dcn_aligner_is_run_to_run_deterministic(deform):
    """deform='basic' | 'fvc' in fp16 at a size with several tiles per block (264 x 520): the whole generator four times, bit
    for bit (the DCN kernel's fp16 instantiation was the one non-deterministic kernel of round 2)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2, deform=deform)
    sd_np = gu.syn.make_state_dict(cfg, seed=171, par_gain=10.0)
    clip = gu.syn.make_clip(seed=172, n=1, t=3, h=264, w=520, slices='IBBBP', block=4, par_classes=3)
    m = _gen_model(cfg, sd_np)
    m.fp16_enabled = True
    a = {k: G(v) for k, v in clip.items()}

    def run():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions']).clone()

    first = run()
    assert torch.isfinite(first).all()
    for rep in range(3):
        torch.randn(1 << 22, device=dev()).sin_()
        assert torch.equal(run(), first), rep

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
    for name in ('gen_parfloat_72x88', 'gen_channel_last_64x64', 'gen_two_layer_64x64'):
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


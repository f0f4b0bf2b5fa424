"""GPU parity of the Winograd F(2x2,3x3) conv (csrc/conv_wino.hip, PNP_OPT_WINOGRAD): op level against ATen in fp64 and against the
direct MFMA kernel, whole generator against the goldens of the imported reference and against the direct path at 720p.

Gates.  The Winograd form is fp32 arithmetic in another summation order plus the +-1 input transform; on unit-scale maps one conv
lands 3-6e-7 (max-abs) from an fp64 contraction (the direct kernel: 1.5-3e-7).  Op level: 2e-6 on unit-scale data.  Whole generator
(33 convs per frame, recurrent over the clip): 2e-5 against the reference's goldens -- north_star's gate is 1e-3 -- and every
golden's smallest ingredient sensitivity (4.1e-5, tests/golden/manifest.json) stays above it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu

pytestmark = pytest.mark.gpu

TOL_OP = 2e-6
TOL_GEN = 2e-5


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def nhwc(x):
    from pnp_vcve_amd import ops
    return ops.nchw_to_nhwc(G(x))[0]


def nchw(y):
    from pnp_vcve_amd import ops
    return ops.nhwc_to_nchw(y.unsqueeze(0)).cpu()


def ref_conv(x, wt, b=None, gamma=None, w1x1=None, par=None, residual=None, act=0):
    """fp64 on the host: act(gamma * (conv3x3 + b) + sum_j par_j * conv1x1_j(x)) + residual   (sr_backbone_utils.py:310-313,329)"""
    xd = torch.from_numpy(x).double()
    y = F.conv2d(xd, torch.from_numpy(wt).double(), None if b is None else torch.from_numpy(b).double(), padding=1)
    if gamma is not None:
        y = y * torch.from_numpy(gamma).double().view(1, -1, 1, 1)
    if w1x1 is not None:
        for j in range(3):
            y = y + torch.from_numpy(par[j]).double()[None, None] * F.conv2d(xd, torch.from_numpy(w1x1[j]).double())
    y = [y, F.relu(y), F.leaky_relu(y, 0.1)][act]
    if residual is not None:
        y = y + torch.from_numpy(residual).double()
    return y


def par_maps(seed, h, w, scale=1.0 / 255.0, block=8, classes=3, empty_rows=0):
    """one-hot per block partition planes like the loader's (values 0 or `scale`); `empty_rows` leading block rows have no record"""
    rng = np.random.RandomState(seed)
    cls = rng.randint(0, classes, ((h + block - 1) // block, (w + block - 1) // block))
    cls = np.repeat(np.repeat(cls, block, 0), block, 1)[:h, :w]
    par = np.stack([(cls == j).astype(np.float32) * np.float32(scale) for j in range(3)])
    par[:, :empty_rows * block] = 0
    return par


@pytest.mark.parametrize('hw', [(16, 16), (24, 40), (37, 53), (64, 64), (72, 88), (128, 256)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_wino_conv_vs_fp64(hw, act):
    """plain 64 -> 64 conv + bias + activation at sizes with whole, ragged and single 16x16 tiles"""
    from pnp_vcve_amd import ops
    h, w = hw
    x = gu.syn.uniform(7, f'x{h}x{w}', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(7, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(7, 'b', (64,), -0.1, 0.1)
    ref = ref_conv(x, wt, b, act=act)
    u = ops.wino_image(ops.pack_conv3x3(G(wt)))
    out = ops.conv3x3_wino(nhwc(x), u, bias=G(b), act=act)
    d = float((nchw(out).double() - ref).abs().max())
    direct = ops.conv3x3([nhwc(x)], [ops.pack_conv3x3(G(wt))], bias=G(b), act=act)
    dd = float((nchw(direct).double() - ref).abs().max())
    print(hw, act, 'winograd vs fp64', d, ' direct vs fp64', dd)
    assert d < TOL_OP


def test_wino_conv_identity_and_shift_weights_localise_layout_bugs():
    """centre-tap identity and one-tap shifts: every transform position, the even/odd halo column order and the output
    scatter are exercised with values whose Winograd sums are exact (x in {-4..4} / 8)"""
    from pnp_vcve_amd import ops
    h, w = 40, 56
    x = np.round(gu.syn.uniform(8, 'x', (1, 64, h, w), -4, 4)).astype(np.float32) / 8
    for (ky, kx) in [(1, 1), (0, 0), (2, 1), (1, 2), (0, 2), (2, 0)]:
        wt = np.zeros((64, 64, 3, 3), np.float32)
        wt[np.arange(64), (np.arange(64) * 7 + 3) % 64, ky, kx] = 1.0          # a channel permutation too (asymmetric B operand)
        ref = ref_conv(x, wt)
        out = ops.conv3x3_wino(nhwc(x), ops.wino_image(ops.pack_conv3x3(G(wt))))
        assert torch.equal(nchw(out).double(), ref), (ky, kx)


def test_shipped_library_does_not_show_the_packed_fp32_signature():
    """tools/repro/wino_packed_f32_hazard.py's pattern on the SHIPPED library: identity weights on a 32x32 random map.  The
    builds with hipcc-generated v_pk_{add,mul}_f32 in conv_wino.hip returned 1/256 of these values wrong, in fixed (lane, register)
    slots (profiles/r05_wino_packed_f32_hazard.txt); the shipped build is compiled with the packed-fp32 target feature off (CPU test:
    tests/test_isa_invariants.py) and must be clean here -- tile kernel, unit kernel, with a residual, and the multi-source kernel"""
    from pnp_vcve_amd import ops
    h, w = 32, 32
    g = torch.Generator(device=dev()).manual_seed(32)
    x = torch.randn(h, w, 64, device=dev(), generator=g)
    r = torch.randn(h, w, 64, device=dev(), generator=g)
    wt = torch.zeros(64, 64, 3, 3, device=dev())
    wt[torch.arange(64), torch.arange(64), 1, 1] = 1.0
    u = ops.wino_image(ops.pack_conv3x3(wt))
    # (random values: the +-1 transforms round, so a right result is within ~1e-6 of x; the signature's wrong values were off by up to 3.4)
    for units in (False, True):
        assert float((ops.conv3x3_wino(x, u, units=units) - x).abs().max()) < 1e-5
        assert float((ops.conv3x3_wino(x, u, residual=r, units=units) - (x + r)).abs().max()) < 1e-5
    wt2 = torch.zeros(64, 67, 3, 3, device=dev())
    wt2[torch.arange(64), 3 + torch.arange(64), 1, 1] = 1.0
    lr4 = torch.zeros(h, w, 4, device=dev())
    urgb = ops.wino_rgb_image(ops.pack_conv3x3(wt2, cbase=0, csrc=3))
    img = torch.empty(1, 65536, device=dev())
    img[0] = ops.wino_image(ops.pack_conv3x3(wt2, cbase=3, csrc=64))
    assert float((ops.conv3x3_wino_ms([lr4, x], [urgb, img[0]]) - x).abs().max()) < 1e-5


@pytest.mark.parametrize('hw', [(64, 64), (40, 72), (48, 50)])
@pytest.mark.parametrize('scale', [1.0 / 255.0, 1.0])
def test_wino_front_half_with_partition_branches(hw, scale):
    """relu(gamma * (conv3x3(x; W) + b) + sum_j par_j * conv1x1_j(x)): the branches accumulate in the transform domain, the gain
    lives in the transformed weights; tile flags on (branch skipping) and off; float-valued and loader-valued maps"""
    from pnp_vcve_amd import ops
    h, w = hw
    x = gu.syn.uniform(9, f'x{h}', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(9, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(9, 'b', (64,), -0.1, 0.1)
    gamma = gu.syn.uniform(9, 'g', (64,), 0.0, 1.0)
    gamma[5] = 0.0                                                              # Hsigmoid can return an exact zero
    w1 = [gu.syn.uniform(9, f'w1{j}', (64, 64, 1, 1), -0.3, 0.3) * (1.0 / scale if scale < 1 else 1.0) * 0.1 for j in range(3)]
    par = par_maps(11, h, w, scale, empty_rows=2)
    if scale == 1.0:
        par = par * gu.syn.uniform(9, 'pf', (3, h, w), 0.2, 1.0)                # general float maps
    ref = ref_conv(x, wt, b, gamma, w1, par, act=1)
    u = ops.wino_image(ops.pack_conv3x3(G(wt)), G(gamma))
    up = ops.wino_par_image(ops.pack_conv1x1([G(v) for v in w1]))
    flags = ops.par_tile_flags(G(par))
    assert int((flags & 7).min()) == 0 and int((flags & 7).max()) > 0          # some tiles skip every branch, some do not
    outs = [ops.conv3x3_wino(nhwc(x), u, bias=G(b), gamma=G(gamma), wino_w1x1=up, par=G(par), par_flags=f, act=1)
            for f in (None, flags)]
    assert torch.equal(outs[0], outs[1])                                        # skipped branches add exact zeros
    mag = float(ref.abs().max())
    d = float((nchw(outs[0]).double() - ref).abs().max())
    print(hw, scale, 'max|winograd - fp64| =', d, 'max|ref| =', mag)
    assert d < TOL_OP * max(1.0, mag)


@pytest.mark.parametrize('block', [4, 8, 16])
def test_wino_front_half_folds_a_block_constant_plane_into_the_weights(block):
    """a wave whose 8x8 quadrant sees ONE live partition plane with ONE value (the loader's one-hot / 255 maps on 8x8 or larger codec
    blocks) adds  p w_j / 4 [+ -; - +]  to the B fragments of positions (1,1) (1,2) (2,1) (2,2) instead of running the branch as MFMAs; 4x4
    blocks straddle every quadrant (nothing folds); all against fp64, tile kernel and unit kernel bit for bit the same"""
    from pnp_vcve_amd import ops
    h, w = 72, 88
    x = gu.syn.uniform(19, 'x', (1, 64, h, w), -1, 1)
    wt = gu.syn.uniform(19, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(19, 'b', (64,), -0.1, 0.1)
    gamma = gu.syn.uniform(19, 'g', (64,), 0.0, 1.0)
    w1 = [gu.syn.uniform(19, f'w1{j}', (64, 64, 1, 1), -0.3, 0.3) * 25.5 for j in range(3)]
    par = par_maps(23, h, w, 1.0 / 255.0, block=block, empty_rows=1)
    ref = ref_conv(x, wt, b, gamma, w1, par, act=1)
    u = ops.wino_image(ops.pack_conv3x3(G(wt)), G(gamma))
    up = ops.wino_par_image(ops.pack_conv1x1([G(v) for v in w1]))
    kw = dict(bias=G(b), gamma=G(gamma), wino_w1x1=up, par=G(par), act=1)
    out = ops.conv3x3_wino(nhwc(x), u, par_flags=ops.par_tile_flags(G(par)), **kw)
    assert torch.equal(out, ops.conv3x3_wino(nhwc(x), u, **kw))                       # with / without branch skipping
    assert torch.equal(out, ops.conv3x3_wino(nhwc(x), u, units=True, **kw))           # tile kernel / unit kernel
    d = float((nchw(out).double() - ref).abs().max())
    print(block, 'max|winograd - fp64| =', d, 'max|ref| =', float(ref.abs().max()))
    assert d < TOL_OP * max(1.0, float(ref.abs().max()))


def test_wino_back_half_with_residual_and_branches_with_residual():
    """x + conv1(o) + b (sr_backbone_utils.py:313,329) and the channel-last order (branches AND residual in one launch, :314-327)"""
    from pnp_vcve_amd import ops
    h, w = 56, 72
    x = gu.syn.uniform(10, 'x', (1, 64, h, w), -1, 1)
    res = gu.syn.uniform(10, 'r', (1, 64, h, w), -2, 2)
    wt = gu.syn.uniform(10, 'w', (64, 64, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(10, 'b', (64,), -0.1, 0.1)
    u = ops.wino_image(ops.pack_conv3x3(G(wt)))
    out = ops.conv3x3_wino(nhwc(x), u, bias=G(b), residual=nhwc(res))
    assert float((nchw(out).double() - ref_conv(x, wt, b, residual=res)).abs().max()) < TOL_OP * 2
    w1 = [gu.syn.uniform(10, f'w1{j}', (64, 64, 1, 1), -3.0, 3.0) for j in range(3)]
    par = par_maps(12, h, w)
    up = ops.wino_par_image(ops.pack_conv1x1([G(v) for v in w1]))
    out = ops.conv3x3_wino(nhwc(x), u, bias=G(b), wino_w1x1=up, par=G(par), residual=nhwc(res))
    assert float((nchw(out).double() - ref_conv(x, wt, b, w1x1=w1, par=par, residual=res)).abs().max()) < TOL_OP * 2


def test_wino_conv_720p_crop_consistency_scaling_and_determinism():
    """BASELINE's frame size: 3600 tiles on 256 persistent blocks (14 or 15 tiles per block, in-place halo refill between them).
    Power-of-two scaling is exact; a 96x112 crop sees the same pixels; ten runs are bit-identical; vs the direct kernel 2e-6"""
    from pnp_vcve_amd import ops
    h, w = 720, 1280
    gen = torch.Generator(device=dev()).manual_seed(720)        # (seeded: the 1e-5 bounds below sit 2x above what this data gives;
    x = torch.randn(h, w, 64, device=dev(), generator=gen)      #  an unseeded draw once in ~500 runs came out above them)
    wt = torch.randn(64, 64, 3, 3, device=dev(), generator=gen) * 0.05
    u = ops.wino_image(ops.pack_conv3x3(wt))
    y = ops.conv3x3_wino(x, u)
    assert torch.equal(ops.conv3x3_wino(x * 2.0, u), y * 2.0)
    for _ in range(9):
        assert torch.equal(ops.conv3x3_wino(x, u), y)
    direct = ops.conv3x3([x], [ops.pack_conv3x3(wt)])
    assert float((y - direct).abs().max()) < 1e-5                               # |x| up to ~5 here: 2e-6 relative to the map's scale
    cy, cx = 304, 512                                                           # crop aligned to the 16x16 tiling or not: same values
    for oy, ox in ((0, 0), (3, 5)):
        yc = ops.conv3x3_wino(x[cy + oy:cy + oy + 96, cx + ox:cx + ox + 112].contiguous(), u)
        assert float((yc[1:-1, 1:-1] - y[cy + oy + 1:cy + oy + 95, cx + ox + 1:cx + ox + 111]).abs().max()) < 1e-5
    ref = F.conv2d(x[cy:cy + 96, cx:cx + 112].permute(2, 0, 1).unsqueeze(0).double().cpu(), wt.double().cpu(), padding=1)
    got = y[cy + 1:cy + 95, cx + 1:cx + 111].permute(2, 0, 1).double().cpu()
    assert float((got - ref[0, :, 1:-1, 1:-1]).abs().max()) < 1e-5


def test_wino_quadrant_units_equal_whole_tiles_bit_for_bit():
    """720p: an XCD band holds 450 tiles for 32 blocks = 14 whole rounds + 2 tiles, which are cut into 8 quadrant units (one per block,
    the waves splitting the output channels).  The same pixels computed inside a 64x128 frame (one whole tile per block, no units) must
    come out bit for bit the same: plain + activation, residual, partition branches with and without skipping; also on a frame whose
    last tiles are ragged (quadrants partly or wholly outside the image)"""
    from pnp_vcve_amd import ops
    wt = torch.randn(64, 64, 3, 3, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    gamma = torch.rand(64, device=dev())
    w1 = [torch.randn(64, 64, 1, 1, device=dev()) * 0.1 for _ in range(3)]
    u, ug, up = ops.wino_image(ops.pack_conv3x3(wt)), ops.wino_image(ops.pack_conv3x3(wt), gamma), ops.wino_par_image(ops.pack_conv1x1(w1))
    for h, w in ((720, 1280), (715, 1270)):
        x = torch.randn(h, w, 64, device=dev())
        res = torch.randn(h, w, 64, device=dev())
        par = G(par_maps(31, h, w, 1.0 / 255.0))
        tiles_x, ntiles = (w + 15) // 16, ((w + 15) // 16) * ((h + 15) // 16)
        assert ntiles == 3600                                      # 8 bands of 450: tiles 448, 449 of every band are unit tiles
        kinds = {
            'plain': lambda xx, rr, pp, ff: ops.conv3x3_wino(xx, u, bias=b, act=2),
            'residual': lambda xx, rr, pp, ff: ops.conv3x3_wino(xx, u, bias=b, residual=rr),
            'branches': lambda xx, rr, pp, ff: ops.conv3x3_wino(xx, ug, bias=b, gamma=gamma, wino_w1x1=up, par=pp, par_flags=ff, act=1),
            'branches, no skipping': lambda xx, rr, pp, ff: ops.conv3x3_wino(xx, ug, bias=b, gamma=gamma, wino_w1x1=up, par=pp, act=1),
        }
        for name, fn in kinds.items():
            full = fn(x, res, par, ops.par_tile_flags(par))
            for band in (0, 3, 7):
                t0 = band * 450 + 448                               # the band's first unit tile (the second one is its right neighbour or wraps)
                ty, tx = t0 // tiles_x, t0 % tiles_x
                # (offsets are multiples of 8: the same 2x2 output tiles AND the same 8x8 quadrants, i.e. the same waves fold their
                #  partition plane into the weights -- a quadrant that straddles two codec blocks runs the branch as MFMAs instead, which
                #  differs in the last bits)
                y0, x0 = max(0, 16 * ty - 16), max(0, min(16 * tx - 32, (w - 128) // 8 * 8))
                y1, x1 = min(h, y0 + 64), min(w, x0 + 136)
                xc, rc, pc = x[y0:y1, x0:x1].contiguous(), res[y0:y1, x0:x1].contiguous(), par[:, y0:y1, x0:x1].contiguous()
                crop = fn(xc, rc, pc, ops.par_tile_flags(pc))
                # interior of the crop (its border row / column sees zeros where the frame has pixels), frame borders included
                iy0, ix0 = (1 if y0 > 0 else 0), (1 if x0 > 0 else 0)
                iy1, ix1 = (y1 - y0 - 1 if y1 < h else y1 - y0), (x1 - x0 - 1 if x1 < w else x1 - x0)
                assert torch.equal(crop[iy0:iy1, ix0:ix1], full[y0 + iy0:y0 + iy1, x0 + ix0:x0 + ix1]), (name, h, w, band)


@pytest.mark.parametrize('hw', [(16, 16), (37, 53), (64, 64), (128, 128), (180, 320)])
def test_wino_unit_kernel_equals_the_tile_kernel_bit_for_bit(hw):
    """small frames: one block per 8x8 quadrant unit (conv3x3_wino_quad_kernel) -- the same arithmetic in the same order as a whole
    tile of the persistent kernel: plain + activation, residual, branches with and without skipping, branches + residual"""
    from pnp_vcve_amd import ops
    h, w = hw
    x = torch.randn(h, w, 64, device=dev())
    res = torch.randn(h, w, 64, device=dev())
    wt = torch.randn(64, 64, 3, 3, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    gamma = torch.rand(64, device=dev())
    w1 = [torch.randn(64, 64, 1, 1, device=dev()) * 0.1 for _ in range(3)]
    par = G(par_maps(33, h, w, 1.0 / 255.0, empty_rows=1))
    flags = ops.par_tile_flags(par)
    u, ug, up = ops.wino_image(ops.pack_conv3x3(wt)), ops.wino_image(ops.pack_conv3x3(wt), gamma), ops.wino_par_image(ops.pack_conv1x1(w1))
    for kw in (dict(wino_w=u, bias=b, act=2), dict(wino_w=u, bias=b, residual=res), dict(wino_w=u),
               dict(wino_w=ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, par_flags=flags, act=1),
               dict(wino_w=ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, act=1),
               dict(wino_w=ug, bias=b, gamma=gamma, wino_w1x1=up, par=par, par_flags=flags, residual=res)):
        assert torch.equal(ops.conv3x3_wino(x, units=True, **kw), ops.conv3x3_wino(x, **kw)), sorted(kw)


@pytest.mark.parametrize('hw', [(16, 16), (37, 53), (128, 128)])
@pytest.mark.parametrize('nwide', [1, 2, 3])
def test_wino_input_conv_unit_kernel_equals_the_tile_kernel_bit_for_bit(hw, nwide):
    """the input conv over [frame, 1-3 wide sources] one block per quadrant unit (conv3x3_wino_quad_ms_kernel) against the persistent
    multi-source kernel: same per-accumulator order (frame first, then source by source, step by step)"""
    from pnp_vcve_amd import ops
    h, w = hw
    cin = 3 + 64 * nwide
    lr4 = torch.zeros(h, w, 4, device=dev())
    lr4[..., :3] = torch.rand(h, w, 3, device=dev())
    wide = [torch.randn(h, w, 64, device=dev()) for _ in range(nwide)]
    wt = torch.randn(64, cin, 3, 3, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    imgs = torch.empty(nwide, 65536, device=dev())
    for k in range(nwide):
        imgs[k] = ops.wino_image(ops.pack_conv3x3(wt, cbase=3 + 64 * k, csrc=64))
    urgb = ops.wino_rgb_image(ops.pack_conv3x3(wt, cbase=0, csrc=3))
    args = ([lr4] + wide, [urgb] + [imgs[k] for k in range(nwide)])
    for act in (0, 2):
        assert torch.equal(ops.conv3x3_wino_ms(*args, bias=b, act=act, units=True), ops.conv3x3_wino_ms(*args, bias=b, act=act))
    assert torch.equal(ops.conv3x3_wino_ms(*args, units=True), ops.conv3x3_wino_ms(*args))


@pytest.mark.parametrize('hw', [(16, 16), (40, 72), (37, 53), (128, 160)])
@pytest.mark.parametrize('nwide', [1, 2, 3])
def test_wino_input_conv_over_the_virtual_concat(hw, nwide):
    """lrelu(conv3x3(cat([lr, wide sources...])) + b) (basicvsr_net.py:484 over iconvsr_ipb_par.py:90,125's concat, never materialised):
    the frame as four RGB chunks of 16 MFMAs that start the accumulators, then one 16-chunk segment per 64-channel source"""
    from pnp_vcve_amd import ops
    h, w = hw
    cin = 3 + 64 * nwide
    lr = gu.syn.uniform(21, f'lr{h}', (1, 3, h, w), 0, 1)
    wide = [gu.syn.uniform(21, f'x{k}{h}', (1, 64, h, w), -1, 1) for k in range(nwide)]
    wt = gu.syn.uniform(21, f'w{nwide}', (64, cin, 3, 3), -0.06, 0.06)
    b = gu.syn.uniform(21, 'b', (64,), -0.1, 0.1)
    ref = F.leaky_relu(F.conv2d(torch.from_numpy(np.concatenate([lr] + wide, 1)).double(), torch.from_numpy(wt).double(),
                                torch.from_numpy(b).double(), padding=1), 0.1)
    lr4 = torch.zeros(h, w, 4, device=dev())
    lr4[..., :3] = G(lr)[0].permute(1, 2, 0)
    imgs = torch.empty(nwide, 65536, device=dev())                      # ONE tensor: the images within 4 GiB of each other
    for k in range(nwide):
        imgs[k] = ops.wino_image(ops.pack_conv3x3(G(wt), cbase=3 + 64 * k, csrc=64))
    urgb = ops.wino_rgb_image(ops.pack_conv3x3(G(wt), cbase=0, csrc=3))
    out = ops.conv3x3_wino_ms([lr4] + [nhwc(x) for x in wide], [urgb] + [imgs[k] for k in range(nwide)], bias=G(b), act=2)
    d = float((nchw(out).double() - ref).abs().max())
    direct = ops.conv3x3([lr4] + [nhwc(x) for x in wide],
                         [ops.pack_conv3x3(G(wt), cbase=0, csrc=3)] + [ops.pack_conv3x3(G(wt), cbase=3 + 64 * k, csrc=64) for k in range(nwide)],
                         bias=G(b), act=2)
    dd = float((nchw(direct).double() - ref).abs().max())
    print(hw, nwide, 'winograd vs fp64', d, ' direct vs fp64', dd, ' max|ref|', float(ref.abs().max()))
    assert d < TOL_OP * max(1.0, float(ref.abs().max()))


def test_wino_input_conv_720p_against_the_direct_kernel_and_itself():
    from pnp_vcve_amd import ops
    h, w = 720, 1280
    g = torch.Generator(device=dev()).manual_seed(5)
    lr4 = torch.rand(h, w, 4, device=dev(), generator=g)
    lr4[..., 3] = 0
    xs = [torch.randn(h, w, 64, device=dev(), generator=g) for _ in range(3)]
    wt = torch.randn(64, 195, 3, 3, device=dev(), generator=g) * 0.03
    b = torch.randn(64, device=dev(), generator=g) * 0.1
    packs = [ops.pack_conv3x3(wt, cbase=0, csrc=3)] + [ops.pack_conv3x3(wt, cbase=3 + 64 * k, csrc=64) for k in range(3)]
    imgs = torch.stack([ops.wino_image(p) for p in packs[1:]])
    urgb = ops.wino_rgb_image(packs[0])
    direct = ops.conv3x3([lr4] + xs, packs, bias=b, act=2)
    y = ops.conv3x3_wino_ms([lr4] + xs, [urgb] + [imgs[k] for k in range(3)], bias=b, act=2)
    assert float((y - direct).abs().max()) < 2e-5                      # |out| up to ~10 here
    for _ in range(5):
        assert torch.equal(ops.conv3x3_wino_ms([lr4] + xs, [urgb] + [imgs[k] for k in range(3)], bias=b, act=2), y)
    y2 = ops.conv3x3_wino_ms([lr4] + xs[:2], [urgb, imgs[0], imgs[1]], bias=b, act=2)
    d2 = ops.conv3x3([lr4] + xs[:2], packs[:3], bias=b, act=2)
    assert float((y2 - d2).abs().max()) < 2e-5


def test_wino_op_refuses_what_it_cannot_do():
    from pnp_vcve_amd import ops
    x = torch.zeros(32, 32, 64, device=dev())
    u = torch.zeros(65536, device=dev())
    with pytest.raises(RuntimeError):
        ops.conv3x3_wino(x.cpu(), u)                                            # no CPU fallback
    with pytest.raises(ValueError):
        ops.conv3x3_wino(torch.zeros(32, 32, 4, device=dev()), u)
    with pytest.raises(RuntimeError):
        ops.conv3x3_wino(x, u, wino_w1x1=torch.zeros(12288, device=dev()))      # branches without a partition map: PNP_ERR_BAD_ARG
    # residual bodies of the tile kernels carry no activation (the reference adds the residual to a bare conv, sr_backbone_utils.py:313,329):
    # refused there (PNP_ERR_UNSUPPORTED), computed by the unit kernels -- act(conv) + residual
    r = torch.randn(32, 32, 64, device=dev())
    uu = ops.wino_image(ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev()) * 0.05))
    with pytest.raises(RuntimeError):
        ops.conv3x3_wino(x + 1.0, uu, residual=r, act=1)
    y = ops.conv3x3_wino(x + 1.0, uu, residual=r, act=1, units=True)
    assert torch.equal(y, torch.relu(ops.conv3x3_wino(x + 1.0, uu, units=True)) + r)
    # the op's gamma scales only the bias; the conv term's gain is whatever wino_image() folded into the image: they must agree
    g = torch.rand(64, device=dev())
    pw = ops.pack_conv3x3(torch.randn(64, 64, 3, 3, device=dev()))
    with pytest.raises(ValueError):
        ops.conv3x3_wino(x, ops.wino_image(pw), gamma=g)
    with pytest.raises(ValueError):
        ops.conv3x3_wino(x, ops.wino_image(pw, g))
    ops.conv3x3_wino(x, ops.wino_image(pw, g), gamma=g)


# ------------------------------------------------------------------------------------------------- whole generator
def build(cfg, sd_np, wino):
    from pnp_vcve_amd import _native
    from pnp_vcve_amd.registry import build_backbone
    m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    m = m.to(dev()).eval()
    m.set_option(_native.OPT_WINOGRAD, wino)
    return m


def run(m, clip):
    a = {k: torch.from_numpy(v).to(dev()) for k, v in clip.items()}
    with torch.no_grad():
        return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])


@pytest.mark.parametrize('case', gu.GEN_CASES, ids=[c['name'] for c in gu.GEN_CASES])
def test_generator_winograd_vs_reference_golden(case):
    """every golden of the imported reference with PNP_OPT_WINOGRAD = 2 (every frame size): all constructor variants, x4 heads,
    sparse_val, two-layer / channel-last blocks, n = 2"""
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = build(cfg, sd_np, 2)
    out = run(m, clip).cpu().numpy()
    ref = gu.load_golden(case['name'])['out']
    d = float(np.abs(out - ref).max())
    print(case['name'], 'max|hip winograd - reference| =', d)
    assert out.shape == ref.shape and d < TOL_GEN
    from pnp_vcve_amd import _native
    m.set_option(_native.OPT_WINOGRAD, 0)                                       # off again -> the direct path's own result
    assert float(np.abs(run(m, clip).cpu().numpy() - ref).max()) < 5e-6


def test_generator_winograd_auto_mode_takes_the_unit_kernel_on_small_frames_and_the_tile_kernel_at_720p():
    from pnp_vcve_amd import _native, synthetic as syn
    case = gu.GEN_CASES[0]
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    a, b, c = (run(build(cfg, sd_np, o), clip) for o in (0, 1, 2))
    # 24 tiles: auto = quadrant units, 2 = the tile kernels: the same values bit for bit; 0 = direct kernels
    assert torch.equal(b, c)
    assert 0 < float((a - b).abs().max()) < TOL_GEN
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd = syn.make_state_dict(cfg, seed=2025)
    clip = syn.make_clip(seed=77, n=1, t=3, h=720, w=1280, slices='IBBBP', qp_mode='qp', crf=25, block=8, par_classes=3)
    m = build(cfg, sd, 0)
    direct = run(m, clip)
    m.set_option(_native.OPT_WINOGRAD, 1)
    w1 = run(m, clip)
    d = float((w1 - direct).abs().max())
    print('720p T=3: max|winograd - direct| =', d)
    assert 0 < d < TOL_GEN
    assert torch.equal(run(m, clip), w1)                                        # deterministic
    m.set_option(_native.OPT_PAR_SKIP, 0)
    assert torch.equal(run(m, clip), w1)                                        # branch skipping adds exact zeros here too


def test_generator_i_frame_front_halves_behind_the_device_side_gate():
    """front halves on the tile kernels are launched twice behind a device-side gate on the frame's partition word: the fold-only kernel
    iff every 8x8 quadrant of the frame is all zero or one constant plane, the branch kernel otherwise (channel-last blocks, branches +
    residual in one launch: plain conv iff an I frame's map is all zero, branch kernel otherwise).  Same bits as the ungated schedule
    (PNP_OPT_PAR_SKIP off: the branch kernel everywhere) for an I frame without records, one WITH records, and a frame whose quadrants
    straddle the codec blocks"""
    from pnp_vcve_amd import _native, synthetic as syn
    # 192x256: whole quadrants everywhere; 180x320 (the reference's third config, HR_davis_LR_128x128_IPB_LR_test.py:43-47): a last row of
    # quadrants 4 pixels high; 148x216: ragged in both directions and a flag tile whose right half lies outside.  A quadrant cut by the
    # frame's edge counts with the pixels it has -- in par_tile_flags' bit 6, in the fold-only kernel and in the branch kernel's own
    # per-wave decision (pixels outside take the nearest inside value) alike, so the gate stays bit-neutral
    # 128x128 and 100x132 (63 tiles, ragged both ways): the quadrant-unit kernels of small frames behind the same gate
    # (conv3x3_wino_quad_gated_kernel: fold-only unit body | branch unit body)
    for extra, (h, w) in (({}, (192, 256)), ({'channel_first': False}, (192, 256)), ({}, (180, 320)), ({'channel_first': False}, (148, 216)),
                          ({}, (128, 128)), ({'channel_first': False}, (128, 128)), ({}, (100, 132)), ({'channel_first': False}, (100, 132))):
        cfg = dict(syn.DEFAULT_GENERATOR_CFG)
        cfg.update(extra)
        sd = syn.make_state_dict(cfg, seed=2025)
        clip = syn.make_clip(seed=99, n=1, t=4, h=h, w=w, slices='IBBBP', qp_mode='qp', crf=25, block=8, par_classes=3)
        assert float(np.abs(clip['partitions'][0, 0]).max()) == 0.0 and float(np.abs(clip['partitions'][0, 1]).max()) > 0.0
        m = build(cfg, sd, 1)
        for with_records in (False, True, 'straddling'):
            if with_records is True:
                clip['partitions'][0, 0] = clip['partitions'][0, 1]         # an I frame that does carry records
            if with_records == 'straddling':                                # a frame whose quadrants straddle codec blocks: not foldable,
                clip['partitions'][0, 2] = np.roll(clip['partitions'][0, 2], 4, axis=-1)      # the gate picks the branch kernel
            m.set_option(_native.OPT_PAR_SKIP, 1)
            gated = run(m, clip)
            m.set_option(_native.OPT_PAR_SKIP, 0)                           # no tile flags, no gate: the branch kernel on every frame
            assert torch.equal(run(m, clip), gated), (extra, h, w, with_records)


def test_generator_winograd_720p_vs_oracle():
    """the headline shape against the oracle itself: 2 x 3 x 720 x 1280 (I then P: MV alignment, partition branches, both sweeps)"""
    from oracle import cpu_ref
    from pnp_vcve_amd import synthetic as syn
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd = syn.make_state_dict(cfg, seed=2025)
    clip = syn.make_clip(seed=4242, n=1, t=2, h=720, w=1280, slices=[73, 80], qp_mode='qp', crf=25, block=8, par_classes=3)
    out = run(build(cfg, sd, 1), clip).cpu()
    c = {k: torch.from_numpy(v) for k, v in clip.items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd), cfg, c['lq'], c['QPs'], c['slices'], c['mvs'], c['base_QPs'],
                                        c['partitions'])
    d = float((out - ref).abs().max())
    print('720p winograd vs oracle:', d)
    assert d < TOL_GEN


@pytest.mark.parametrize('hw', [(192, 256), (180, 320)])
@pytest.mark.parametrize('channel_first', [True, False])
def test_generator_gated_front_halves_vs_oracle(channel_first, hw):
    """a frame size that takes the TILE kernels behind the device-side gate (192x256 = 192 tiles, whole 8x8 quadrants everywhere),
    against the oracle itself: channel-first blocks run the fold-only body of conv3x3_wino_gated_kernel<false>, channel-last
    blocks (sr_backbone_utils.py:314-327: branches AND residual in one launch) the fold-only + residual body of
    conv3x3_wino_gated_kernel<true>; a P frame whose quadrants straddle the codec blocks takes the branch bodies of the same kernels"""
    from oracle import cpu_ref
    from pnp_vcve_amd import _native, synthetic as syn
    cfg = dict(syn.DEFAULT_GENERATOR_CFG, channel_first=channel_first)
    sd = syn.make_state_dict(cfg, seed=2025)
    h, w = hw              # (180x320: configs[4]'s own frame size, a ragged last row of quadrants -- folded since round 6)
    clip = syn.make_clip(seed=606, n=1, t=4, h=h, w=w, slices='IBBBP', qp_mode='qp', crf=25, block=8, par_classes=3)
    clip['partitions'][0, 2] = np.roll(clip['partitions'][0, 2], 4, axis=-1)        # frame 2: not foldable -> branch kernel
    m = build(cfg, sd, 1)
    assert m.get_option(_native.OPT_PAR_SKIP) == 1                                  # the gate is on by default
    out = run(m, clip).cpu()
    c = {k: torch.from_numpy(v) for k, v in clip.items()}
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd), cfg, c['lq'], c['QPs'], c['slices'], c['mvs'], c['base_QPs'],
                                        c['partitions'])
    d = float((out - ref).abs().max())
    print(hw, 'gated, channel_first =', channel_first, ': max|hip - oracle| =', d)
    assert d < TOL_GEN

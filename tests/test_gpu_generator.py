"""GPU parity of the whole BAE/CAA forward (pnp_generator_forward through the registry class)
against the golden vectors produced by the imported reference."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import cpu_ref

pytestmark = pytest.mark.gpu

# north_star's gate is 1e-3 on enhanced frames.  The exact-fp32 path and the split-fp16 path deliver 1.2-1.8e-7 on every golden
# case; they are held to 5e-6, i.e. at least 8x below the smallest effect any ingredient has on any golden case in which it acts
# (tests/golden/manifest.json 'sensitivity': min 4.1e-5 = base*2 on gen_vsr_nocat_e4_64x64), so a build that ignored an input in
# one configuration cannot pass that configuration's golden.
TOL = 5e-6


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def build(cfg, sd_np):
    from pnp_vcve_amd.registry import build_backbone
    m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    return m.to(dev()).eval()


def run(m, clip):
    a = {k: torch.from_numpy(v).to(dev()) for k, v in clip.items()}
    with torch.no_grad():
        return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])


@pytest.mark.parametrize('case', gu.GEN_CASES, ids=[c['name'] for c in gu.GEN_CASES])
def test_generator_vs_reference_golden(case):
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    out = run(build(cfg, sd_np), clip).cpu().numpy()
    ref = gu.load_golden(case['name'])['out']
    assert out.shape == ref.shape
    d = float(np.abs(out - ref).max())
    print(case['name'], 'max|hip - reference| =', d)
    assert d < TOL


def test_psnr_delta_vs_reference_is_negligible():
    case = gu.GEN_CASES[0]
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    out = run(build(cfg, sd_np), clip).cpu()
    ref = torch.from_numpy(gu.load_golden(case['name'])['out'])
    gt = torch.from_numpy(clip['gt'])
    assert abs(cpu_ref.clip_psnr(out, gt) - cpu_ref.clip_psnr(ref, gt)) < 0.01


def test_reference_error_behaviour():
    case = gu.GEN_CASES[0]
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = build(cfg, sd_np)
    small = {k: (v[..., :32, :32] if v.shape[-1] > 1 else v) for k, v in clip.items()}
    with pytest.raises(AssertionError):                 # iconvsr_ipb_par.py:51
        run(m, {k: np.ascontiguousarray(v) for k, v in small.items()})
    odd = {k: (v[..., :, :66] if v.shape[-1] > 1 else v) for k, v in clip.items()}
    with pytest.raises(ValueError):                     # flow_warp.py:27-29
        run(m, {k: np.ascontiguousarray(v) for k, v in odd.items()})
    with pytest.raises(RuntimeError):                   # CPU tensors: loud failure, no fallback
        a = {k: torch.from_numpy(v) for k, v in clip.items()}
        m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])


def test_repack_after_weight_update():
    case = gu.GEN_CASES[1]
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = build(cfg, sd_np)
    a = run(m, clip)
    with torch.no_grad():
        m.conv_last.bias.add_(0.25)
    b = run(m, clip)
    assert float((b - a - 0.25).abs().max()) < 1e-6


def test_deterministic_and_batch_order_independent():
    case = gu.GEN_CASES[5]          # n = 2, different key patterns
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = build(cfg, sd_np)
    a = run(m, clip)
    b = run(m, clip)
    assert torch.equal(a, b)
    swapped = {k: np.ascontiguousarray(v[::-1]) for k, v in clip.items()}
    c = run(m, swapped)
    assert torch.equal(c[0], a[1]) and torch.equal(c[1], a[0])


def test_720p_clip_crop_consistency_and_oracle_spot_check():
    """BASELINE configs[2] shape (T shortened to 3): the 720p result must agree, far from the crop
    border, with the same network run on a crop -- on the GPU and on the oracle."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=77, par_gain=10.0)
    clip = gu.syn.make_clip(seed=770, n=1, t=3, h=720, w=1280, slices='IBBBP', qp_mode='qp', crf=25, mv_range=8)
    m = build(cfg, sd_np)
    full = run(m, clip)
    assert full.shape == (1, 3, 3, 720, 1280)
    assert torch.isfinite(full).all()
    cy, cx, ch, cw = 200, 480, 320, 320
    crop = {k: (np.ascontiguousarray(v[..., cy:cy + ch, cx:cx + cw]) if v.shape[-1] > 1 else v) for k, v in clip.items()}
    part = run(m, crop)
    # receptive field: (17 convs + |mv| <= 2 px) per recurrent step, <= 2*3 steps + 2 head convs
    mrg = 128
    d = float((part[..., mrg:-mrg, mrg:-mrg] - full[..., cy + mrg:cy + ch - mrg, cx + mrg:cx + cw - mrg]).abs().max())
    print('720p crop consistency', d)
    assert d < 1e-5
    # oracle on a smaller crop (CPU, seconds)
    oy, ox, oh, ow = 296, 576, 128, 128
    oc = {k: (np.ascontiguousarray(v[..., oy:oy + oh, ox:ox + ow]) if v.shape[-1] > 1 else v) for k, v in clip.items()}
    t = {k: torch.from_numpy(v) for k, v in oc.items()}
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), cfg, t['lq'], t['QPs'], t['slices'], t['mvs'],
                                        t['base_QPs'], t['partitions'])
    got = run(m, oc).cpu()
    assert float((got - ref).abs().max()) < TOL


@pytest.mark.parametrize('shape', [(1, 1, 64, 64), (1, 2, 64, 64), (1, 3, 68, 76), (3, 2, 64, 100), (1, 9, 64, 64)],
                         ids=lambda s: 'n%d_t%d_%dx%d' % s)
def test_edge_shapes_vs_oracle(shape):
    """T = 1 (no alignment at all), T = 2, ragged tiles (sizes that are multiples of 4 but not of the
    8x16 tile), n = 3, and a 9-frame clip with two interior key frames -- against the pinned oracle."""
    n, t, h, w = shape
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=300 + t, par_gain=10.0)
    clip = gu.syn.make_clip(seed=3000 + h + w, n=n, t=t, h=h, w=w, slices='IBBBP', qp_mode='qp',
                            crf=[15, 25, 35][:n] if n > 1 else 25, block=4)
    out = run(build(cfg, sd_np), clip).cpu()
    c = {k: torch.from_numpy(v) for k, v in clip.items()}
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), cfg, c['lq'], c['QPs'], c['slices'], c['mvs'],
                                        c['base_QPs'], c['partitions'])
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) < TOL


def test_per_frame_base_qp_values_are_honoured():
    """base_QPs is an input tensor: distinct values per frame must select distinct expert mixtures
    (the dedup on the host is by value, not by clip)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=55, par_gain=10.0)
    clip = gu.syn.make_clip(seed=555, n=1, t=4, h=64, w=64)
    clip['base_QPs'] = (np.array([15, 35, 15, 51], np.float32) / 255.0).reshape(1, 4, 1, 1, 1)
    out = run(build(cfg, sd_np), clip).cpu()
    c = {k: torch.from_numpy(v) for k, v in clip.items()}
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), cfg, c['lq'], c['QPs'], c['slices'], c['mvs'],
                                        c['base_QPs'], c['partitions'])
    assert float((out - ref).abs().max()) < TOL


@pytest.mark.parametrize('deform', ['basic', 'fvc'])
def test_deformable_aligners_vs_oracle(deform):
    """deform='basic'|'fvc' (iconvsr_ipb.py:19-22): flow-guided modulated deformable alignment.  Not used
    by the shipped configs; parity is against the oracle's restatement of mmcv's op (unpinned)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, deform=deform)
    sd_np = gu.syn.make_state_dict(cfg, seed=71, par_gain=10.0)
    clip = gu.syn.make_clip(seed=710, n=1, t=3, h=64, w=72, slices='IBBBP', block=4, mv_range=16)
    m = build(cfg, sd_np)
    assert 'deform_align.conv_offset.2.weight' in m.state_dict()
    out = run(m, clip).cpu()
    c = {k: torch.from_numpy(v) for k, v in clip.items()}
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), cfg, c['lq'], c['QPs'], c['slices'], c['mvs'],
                                        c['base_QPs'], c['partitions'])
        vos = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), dict(cfg, deform='vos'), c['lq'], c['QPs'],
                                        c['slices'], c['mvs'], c['base_QPs'], c['partitions'])
    assert float((ref - vos).abs().max()) > 1e-3          # the aligner matters in this case
    assert float((out - ref).abs().max()) < TOL


def test_long_clip_100_frames_vs_oracle():
    """the reference evaluates whole 100-frame REDS clips (configs/HR_davis_LR_128x128.py:202): T = 100 keeps
    100 feature maps live in the workspace, 4 CAA launches, many key frames."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)        # 2 blocks keep the CPU oracle at seconds
    sd_np = gu.syn.make_state_dict(cfg, seed=88, par_gain=10.0)
    clip = gu.syn.make_clip(seed=880, n=1, t=100, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=35)
    out = run(build(cfg, sd_np), clip).cpu()
    c = {k: torch.from_numpy(v) for k, v in clip.items()}
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), cfg, c['lq'], c['QPs'], c['slices'], c['mvs'],
                                        c['base_QPs'], c['partitions'])
    assert out.shape == (1, 100, 3, 64, 64)
    assert float((out - ref).abs().max()) < TOL


def test_long_clip_720p_fits_and_runs():
    """T = 30 at 1280x720: 30 feature slots (7 GB) + inputs; finite output and first/last-frame sanity."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=89)
    m = build(cfg, sd_np)
    t, h, w = 30, 720, 1280
    g = torch.Generator(device='cuda').manual_seed(1)
    lq = torch.rand(1, t, 3, h, w, device='cuda', generator=g)
    mvs = (torch.randint(-16, 17, (1, t, 4, h // 8, w // 8), device='cuda', generator=g).float() / 4
           ).repeat_interleave(8, 3).repeat_interleave(8, 4).contiguous()
    par = torch.zeros(1, t, 3, h, w, device='cuda')
    sl = torch.tensor([73.0 if i == 0 else (80.0 if i % 4 == 0 else 66.0) for i in range(t)], device='cuda').view(1, t, 1, 1, 1)
    qp = torch.full((1, t, 1, 1, 1), 28 / 255.0, device='cuda')
    bq = torch.full((1, t, 1, 1, 1), 25 / 255.0, device='cuda')
    with torch.no_grad():
        out = m(lq, qp, sl, mvs, bq, par)
    assert out.shape == lq.shape and torch.isfinite(out).all()
    assert float((out - lq).abs().mean()) < 0.2


def _gpu_clip(t, h, w, seed, pattern, mv_q=16):
    """a synthetic clip made on the device (a 100-frame 720p clip is 3.7 GB of inputs: no host copies)"""
    g = torch.Generator(device='cuda').manual_seed(seed)
    lq = torch.rand(1, t, 3, h, w, device='cuda', generator=g)
    mvs = (torch.randint(-mv_q, mv_q + 1, (1, t, 4, h // 8, w // 8), device='cuda', generator=g).float() / 4
           ).repeat_interleave(8, 3).repeat_interleave(8, 4).contiguous()
    cls = torch.randint(0, 3, (1, t, 1, h // 8, w // 8), device='cuda', generator=g)
    par = (torch.cat([(cls == j) for j in range(3)], dim=2).float() / 255.0).repeat_interleave(8, 3).repeat_interleave(8, 4).contiguous()
    sl = torch.tensor(pattern, dtype=torch.float32, device='cuda').view(1, t, 1, 1, 1)
    qp = torch.full((1, t, 1, 1, 1), 28 / 255.0, device='cuda')
    bq = torch.full((1, t, 1, 1, 1), 25 / 255.0, device='cuda')
    return dict(lq=lq, QPs=qp, slices=sl, mvs=mvs, base_QPs=bq, partitions=par)


def _fwd(m, c):
    with torch.no_grad():
        return m(c['lq'], c['QPs'], c['slices'], c['mvs'], c['base_QPs'], c['partitions'])


def test_reference_clip_length_t100_at_720p_shipped_config():
    """The reference evaluates whole 100-frame REDS clips (configs/HR_davis_LR_128x128.py:202) at 1280x720: 100 live feature
    slots = 24.5 GB of workspace + 4.8 GB of inputs and output, slot offsets far beyond 4 GiB.  Shipped configuration.  Every
    frame depends on every other one here (neighbour features chain through all 100 frames), so no crop can reproduce it; what is
    checked: finite output, the residual head stays near the input, and -- after the cached workspace has been overwritten with
    NaN bit patterns -- a second forward returns the same bits (every byte the schedule reads, it wrote in this call)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    m = build(cfg, gu.syn.make_state_dict(cfg, seed=89))
    t, h, w = 100, 720, 1280
    c = _gpu_clip(t, h, w, 5, [73.0 if i == 0 else (80.0 if i % 4 == 0 else 66.0) for i in range(t)])
    out = _fwd(m, c)
    assert out.shape == (1, t, 3, h, w) and torch.isfinite(out).all()
    assert float((out - c['lq']).abs().mean()) < 0.2
    (key, ws), = m._workspace.items()
    assert ws.numel() > 24 * 10**9 and key[1:4] == (t, h, w)
    ws.fill_(0xFF)                                        # NaN bit patterns everywhere in the 24.5 GB
    again = _fwd(m, c)
    assert torch.equal(again, out)
    del out, again
    torch.cuda.empty_cache()


def test_t100_at_720p_equals_a_crop_far_from_the_border():
    """T = 100 at 1280x720 against T = 100 on a 384x384 crop, all frames, bit-level tolerance.  With `with_cat=False` and only the
    clip ends as key frames (all-B slices) a frame depends on its own pixels and on the two end frames only (backward feature of
    the last frame -> every backward feature; forward feature of frame 0 -> every forward feature): a receptive field of
    4 x 17 convolution pixels + 2 MV hops of <= 4 px, whatever T is -- so the central 128x128 of the crop must reproduce the
    full-frame result.  This is the full-size check of slot addressing / the long-clip schedule that the shipped configuration
    (every frame coupled to every other one) cannot give."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, with_cat=False)
    m = build(cfg, gu.syn.make_state_dict(cfg, seed=90, par_gain=10.0))
    t, h, w = 100, 720, 1280
    c = _gpu_clip(t, h, w, 6, [73.0] + [66.0] * (t - 1), mv_q=16)
    full = _fwd(m, c)
    assert torch.isfinite(full).all()
    y0, x0, s = 168, 448, 384                       # crop origin on the 8x8 block grid, 128 px of margin around the compared centre
    cc = {k: (v[..., y0:y0 + s, x0:x0 + s].contiguous() if v.shape[-1] == w else v) for k, v in c.items()}
    crop = _fwd(m, cc)
    a = full[..., y0 + 128:y0 + 256, x0 + 128:x0 + 256]
    b = crop[..., 128:256, 128:256]
    d = float((a - b).abs().max())
    print('T=100 720p vs 384x384 crop, central 128x128, all 100 frames: max|d| =', d)
    assert d < 1e-5
    # and the property is not vacuous: nearer to the crop border the two DO differ (zero padding reaches in)
    assert float((full[..., y0:y0 + 16, x0:x0 + s] - crop[..., 0:16, :]).abs().max()) > 1e-4
    del full, crop
    torch.cuda.empty_cache()


def test_randomised_configs_and_shapes_vs_oracle():
    """12 seeded random draws over the constructor switches, clip length, frame size, slice pattern and batch
    size -- every draw against the pinned oracle (the reference's own option space)."""
    rng = np.random.RandomState(20261002)
    tried_split = 0
    for trial in range(12):
        with_bias = bool(rng.randint(2))
        with_se = with_bias and bool(rng.randint(2))
        cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG,
                   num_blocks=int(rng.randint(1, 4)), num_experts=int(rng.choice([2, 6, 10])),
                   with_cat=bool(rng.randint(2)), align_key=bool(rng.randint(2)), vsr=bool(rng.randint(4) == 0),
                   expert_softmax=bool(rng.randint(2)), with_bias=with_bias, with_se=with_se,
                   use_base_qp=True if with_bias else bool(rng.randint(2)),
                   one_layer=bool(rng.randint(2)), channel_first=bool(rng.randint(2)),
                   deform=str(rng.choice(['vos', 'vos', 'basic', 'fvc'])))
        n, t = int(rng.choice([1, 1, 2])), int(rng.randint(1, 6))
        h, w = 64 + 4 * int(rng.randint(0, 6)), 64 + 4 * int(rng.randint(0, 10))
        pattern = [73] + [int(rng.choice([66, 66, 80, 73])) for _ in range(t - 1)]
        sd_np = gu.syn.make_state_dict(cfg, seed=1000 + trial, par_gain=10.0)
        qp_mode = str(rng.choice(['qp', 'ipb']))
        par_scale = float(rng.choice([1 / 255.0, 1.0]))
        clip = gu.syn.make_clip(seed=2000 + trial, n=n, t=t, h=h, w=w, slices=pattern, block=4, qp_mode=qp_mode,
                                crf=[15, 35][:n] if n > 1 else 25, par_scale=par_scale)
        m = build(cfg, sd_np)
        out = run(m, clip).cpu()
        c = {k: torch.from_numpy(v) for k, v in clip.items()}
        with torch.no_grad():
            ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), cfg, c['lq'], c['QPs'], c['slices'],
                                            c['mvs'], c['base_QPs'], c['partitions'])
        # some draws (par = 1 with the 10x partition gain, two dynamic layers) blow the activations up to 1e8 in
        # the reference itself: the gate is relative to the output scale
        scale = max(1.0, float(ref.abs().max()))
        d = float((out - ref).abs().max()) / scale
        assert out.shape == ref.shape and d < TOL, (trial, cfg, (n, t, h, w), pattern, d, scale)
        # the same draw in split fp16, where its operands are representable: hi = fp16(x) saturates at 65504 (include/pnpvcve.h), and
        # the par = 1 draws reach 1e8 -- those are the exact fp32 path's alone
        if par_scale < 0.5 and scale < 1e3:
            m.precision = 'f16x3'
            d3 = float((run(m, clip).cpu() - ref).abs().max()) / scale
            assert d3 < TOL, ('f16x3', trial, cfg, (n, t, h, w), pattern, d3, scale)
            tried_split += 1
    assert tried_split >= 3


@pytest.mark.parametrize('fp16', [False, True])
def test_batch_samples_run_concurrently_and_match_one_at_a_time(fp16):
    """n = 10 small clips with mixed crf / slice-independent side info: the batch goes through the multi-context path
    (8 contexts on side streams, samples 8 and 9 reuse contexts 0 and 1) and must equal sample-by-sample calls
    bit for bit; a following call on the caller's stream sees the finished output (join)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=61, par_gain=10.0)
    clip = gu.syn.make_clip(seed=62, n=10, t=4, h=64, w=80, slices='IBBBP', block=4, qp_mode='ipb',
                            crf=[15, 25, 35, 25, 15, 35, 20, 30, 15, 35])
    m = build(cfg, sd_np)
    m.fp16_enabled = fp16
    out = run(m, clip)
    chk = out.sum()                      # consumer on the caller's stream, right behind the join
    assert m.MAX_CONTEXTS == 8 and m._workspace and next(iter(m._workspace))[0] == 8
    singles = []
    for b in range(10):
        one = {k: v[b:b + 1] for k, v in clip.items()}
        singles.append(run(m, one))
        assert next(iter(m._workspace))[0] == 1
    ref = torch.cat(singles)
    assert torch.equal(out, ref)
    assert float(chk) == float(ref.sum())


@pytest.mark.parametrize('precision', ['fp32', 'f16x3', 'fp16'])
def test_large_frames_run_two_samples_on_two_streams_and_match_one_at_a_time(precision):
    """A large frame fills the chip, but every persistent conv launch ends with a partial round of tiles and a dispatch gap: a
    batch of large frames runs TWO samples at a time on two streams so that they fill each other's tails (r04: +5 % at 720p).
    Three 720p samples (the third reuses context 0) must equal sample-by-sample calls bit for bit, and a consumer on the
    caller's stream right behind the call sees the finished output."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    m = build(cfg, gu.syn.make_state_dict(cfg, seed=63, par_gain=10.0))
    m.precision = precision
    clip = gu.syn.make_clip(seed=64, n=3, t=2, h=720, w=1280, slices='IBBBP', crf=[15, 25, 35], par_classes=3)
    out = run(m, clip)
    chk = out.double().sum()
    assert m.LARGE_FRAME_CONTEXTS == 2 and next(iter(m._workspace))[0] == 2
    singles = []
    for b in range(3):
        singles.append(run(m, {k: v[b:b + 1] for k, v in clip.items()}))
        assert next(iter(m._workspace))[0] == 1
    ref = torch.cat(singles)
    assert out.shape == (3, 2, 3, 720, 1280) and torch.equal(out, ref) and float(chk) == float(ref.double().sum())
    assert not torch.equal(out[0], out[1])


@pytest.mark.parametrize('fp16', [False, True])
@pytest.mark.parametrize('n', [1, 3])
def test_hip_graph_replay_is_bit_identical_to_eager(fp16, n):
    """use_graphs: the clip's launches are captured once per (shape, side info) and replayed; new pixel data goes
    through the graph's static buffers.  Different side info = different graph (QP values are kernel arguments)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=71, par_gain=10.0)
    m = build(cfg, sd_np)
    m.fp16_enabled = fp16
    clips = [gu.syn.make_clip(seed=72 + i, n=n, t=4, h=64, w=96, slices='IBBBP', block=4, qp_mode='ipb',
                              crf=[15, 25, 35][:n] if n > 1 else 25) for i in range(3)]
    eager = [run(m, c).clone() for c in clips]
    m.use_graphs = True
    for rep in range(2):
        for c, e in zip(clips, eager):
            assert torch.equal(run(m, c), e)
    assert len(m._graphs) == 1                 # same shape and side info: one capture, five replays
    other = gu.syn.make_clip(seed=80, n=n, t=4, h=64, w=96, slices='allP', block=4, qp_mode='ipb',
                             crf=[15, 25, 35][:n] if n > 1 else 25)
    g = run(m, other)
    assert len(m._graphs) == 2
    m.use_graphs = False
    assert torch.equal(g, run(m, other))


def test_forward_workspace_errors_through_the_raw_abi():
    """PNP_ERR_WORKSPACE (1003) for a short or misaligned workspace, PNP_ERR_BAD_ARG (1001) for n < 1 -- nothing launched."""
    import ctypes
    from pnp_vcve_amd import _native
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=1)
    m = build(cfg, gu.syn.make_state_dict(cfg, seed=91))
    clip = gu.syn.make_clip(seed=92, n=1, t=2, h=64, w=64, slices='IBBBP')
    out = run(m, clip)                                   # packs the weights
    L = _native.lib()
    a = {k: torch.from_numpy(v).to(dev()) for k, v in clip.items()}
    need = int(L.pnp_generator_workspace_bytes(m._handle, 2, 64, 64))
    ws = torch.empty(need + 512, device=dev(), dtype=torch.uint8)
    side = torch.stack([a['slices'].reshape(1, 2), a['QPs'].reshape(1, 2), a['base_QPs'].reshape(1, 2)]).float().cpu()
    fp = ctypes.POINTER(ctypes.c_float)
    P = lambda x: ctypes.c_void_p(x.data_ptr())      # noqa: E731
    res = torch.zeros_like(out)

    def call(ws_ptr, ws_bytes, n=1):
        return L.pnp_generator_forward(m._handle, P(m._flat), P(m._packed), P(a['lq']), P(a['mvs']), P(a['partitions']),
                                       ctypes.cast(side.data_ptr(), fp), ctypes.cast(side.data_ptr() + 8, fp),
                                       ctypes.cast(side.data_ptr() + 16, fp), P(res), ctypes.c_void_p(ws_ptr), ws_bytes,
                                       n, 2, 64, 64, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    assert call(ws.data_ptr(), need - 1) == 1003
    assert call(ws.data_ptr() + 16, need) == 1003
    assert call(ws.data_ptr(), need, n=0) == 1001
    torch.cuda.synchronize()
    assert float(res.abs().max()) == 0.0
    assert call(ws.data_ptr(), need) == 0
    torch.cuda.synchronize()
    assert torch.equal(res, out)


@pytest.mark.parametrize('hw', [(256, 512), (72, 96)], ids=['persistent_kernel', 'tile_kernel_4x16'])
@pytest.mark.parametrize('par_kind', ['one_hot_blocks', 'dense_float', 'all_zero'])
def test_partition_branch_skipping_is_bit_identical(par_kind, hw):
    """Both conv kernels skip a 1x1 partition branch on tiles where its plane is zero (per-tile flags,
    pnp_par_tile_flags_f32): exact zeros dropped, so the clip must not change by a single bit."""
    from pnp_vcve_amd import _native
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    sd_np = gu.syn.make_state_dict(cfg, seed=95, par_gain=10.0)
    clip = gu.syn.make_clip(seed=96, n=1, t=3, h=hw[0], w=hw[1], slices='IBBBP', block=8,
                            par_scale=1.0 if par_kind == 'dense_float' else 1 / 255.0)
    if par_kind == 'dense_float':
        clip['partitions'] = gu.syn.uniform(97, 'pf', clip['partitions'].shape, 0.0, 1.0)
    elif par_kind == 'all_zero':
        clip['partitions'] = np.zeros_like(clip['partitions'])
    else:
        planes = (clip['partitions'][0, 1:] != 0).sum(1)             # P/B frames: at most one plane per pixel
        assert planes.max() == 1 and planes.mean() > 0.5 and float(np.abs(clip['partitions'][0, 0]).max()) == 0.0
    m = build(cfg, sd_np)
    m.set_option(_native.OPT_PAR_SKIP, 0)
    ref = run(m, clip).clone()
    m.set_option(_native.OPT_PAR_SKIP, 1)
    out = run(m, clip)
    assert torch.equal(out, ref)
    if par_kind == 'one_hot_blocks':          # and the branch is live: zeroing the map changes the result
        clip0 = dict(clip, partitions=np.zeros_like(clip['partitions']))
        assert float((run(m, clip0) - out).abs().max()) > 1e-5


@pytest.mark.parametrize('vsr', [False, True])
def test_conv_last_on_the_vector_alus_matches_the_mfma_kernel(vsr):
    """conv_last (64 -> 3) runs on the VALUs with scalar weights (conv_last.hip); against the MFMA kernel it replaces
    (same fp32 products, different summation order) on a ragged frame, both output modes."""
    from pnp_vcve_amd import _native
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=1, vsr=vsr)
    sd_np = gu.syn.make_state_dict(cfg, seed=101, par_gain=10.0)
    clip = gu.syn.make_clip(seed=102, n=1, t=2, h=68, w=100, slices='IBBBP', block=4)
    m = build(cfg, sd_np)
    m.set_option(_native.OPT_CONV_LAST_VALU, 0)
    ref = run(m, clip).clone()
    m.set_option(_native.OPT_CONV_LAST_VALU, 1)
    out = run(m, clip)
    d = float((out - ref).abs().max())
    assert out.shape == ref.shape and 0.0 < d < 2e-6, d


# ------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] sizes (HR_davis_LR_128x128_IPB_LR_test.py: 180x320 frames, +x4 heads, mixed crf15/25/35)
# and the headline size directly against the oracle
# ------------------------------------------------------------------------------------------------------------
_ORACLE_CACHE = {}


def _oracle(cfg, sd_np, clip, key=None):
    """the pinned CPU oracle on a whole clip; `key` caches the result for tests that share (cfg, weights, clip) seeds"""
    if key is not None and key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    t = {k: torch.from_numpy(v) for k, v in clip.items()}
    with torch.no_grad():
        out = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), cfg, t['lq'], t['QPs'], t['slices'], t['mvs'],
                                        t['base_QPs'], t['partitions'])
    if key is not None:
        _ORACLE_CACHE[key] = out
    return out


def test_lr180_clip_fp32_vs_oracle():
    """configs[4] as the config ships (vsr=False): a 180x320 clip (T = 3) enhanced at 180x320, whole generator vs the
    pinned oracle.  180 = 22.5 tiles of 8 rows: ragged bottom tiles on the small-tile kernel."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=401, par_gain=10.0)
    clip = gu.syn.make_clip(seed=4010, n=1, t=3, h=180, w=320, slices='IBBBP', qp_mode='ipb', crf=25, block=4)
    out = run(build(cfg, sd_np), clip).cpu()
    ref = _oracle(cfg, sd_np, clip)
    d = float((out - ref).abs().max())
    print('180x320 T=3 fp32 max|hip - oracle| =', d)
    assert out.shape == (1, 3, 3, 180, 320) and d < TOL


def test_lr180_x4_heads_fp32_vs_oracle():
    """configs[4] as BASELINE.json describes it (vsr=True): LR 180x320 -> 720x1280 through the two PixelShufflePack
    heads (iconvsr_ipb_par.py:135-142), T = 2, vs the oracle."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, vsr=True)
    sd_np = gu.syn.make_state_dict(cfg, seed=402, par_gain=10.0)
    clip = gu.syn.make_clip(seed=4020, n=1, t=2, h=180, w=320, slices='IBBBP', qp_mode='ipb', crf=35, block=4)
    out = run(build(cfg, sd_np), clip).cpu()
    ref = _oracle(cfg, sd_np, clip)
    d = float((out - ref).abs().max())
    print('180x320 -> 720x1280 T=2 fp32 max|hip - oracle| =', d)
    assert out.shape == (1, 2, 3, 720, 1280) and d < TOL


@pytest.mark.parametrize('vsr', [False, True], ids=['enhance', 'x4'])
def test_lr180_mixed_crf_batch_fp16_vs_fp32(vsr):
    """configs[4]: n = 3 clips of crf 15/25/35 (three different expert mixtures in one batch) with the fp16 MFMA convs
    against the fp32 path of the same build: PSNR delta < 1e-3 dB per clip (north_star's PSNR gate), max-abs 2e-2."""
    from pnp_vcve_amd.ops import psnr_frames
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, vsr=vsr)
    sd_np = gu.syn.make_state_dict(cfg, seed=403)
    clip = gu.syn.make_clip(seed=4030, n=3, t=3, h=180, w=320, slices='IBBBP', qp_mode='ipb', crf=[15, 25, 35], block=4)
    m = build(cfg, sd_np)
    o32 = run(m, clip)
    m.fp16_enabled = True
    o16 = run(m, clip)
    gt = torch.from_numpy(clip['gt']).to(dev())
    if vsr:
        gt = gt.repeat_interleave(4, -1).repeat_interleave(4, -2).contiguous()
    p32, p16 = psnr_frames(o32, gt).mean(dim=1), psnr_frames(o16, gt).mean(dim=1)
    print('fp16 vs fp32 per-clip PSNR', p32.tolist(), p16.tolist(), 'max-abs', float((o16 - o32).abs().max()))
    assert o16.shape == o32.shape and torch.isfinite(o16).all()
    assert float((p16 - p32).abs().max()) < 1e-3
    assert 0.0 < float((o16 - o32).abs().max()) < 2e-2
    # the three clips really ran three different mixtures: crf only enters through base_QPs
    same = dict(clip, base_QPs=np.full_like(clip['base_QPs'], 25 / 255.0))
    m.fp16_enabled = False
    assert float((run(m, same)[0] - o32[0]).abs().max()) > 1e-6


@pytest.mark.parametrize('vsr', [False, True], ids=['enhance', 'x4'])
def test_lr180_fp16_small_frame_kernel_vs_oracle(vsr):
    """configs[4] at its own workload, fp16 MFMA operands, against the PINNED ORACLE (not against this build's fp32 path):
    180x320 is 450 tiles, so every 64->64 conv runs conv3x3_f16_small_kernel.  Gates: max-abs 2e-2 on [0,1] frames and
    |PSNR delta| < 1e-3 dB vs the oracle's output (north_star's PSNR gate); T = 3 (enhance) / T = 2 (x4 heads)."""
    from pnp_vcve_amd import _native
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, vsr=vsr)
    sd_np = gu.syn.make_state_dict(cfg, seed=405)
    clip = gu.syn.make_clip(seed=4050, n=1, t=2 if vsr else 3, h=180, w=320, slices='IBBBP', qp_mode='ipb', crf=25, block=4)
    m = build(cfg, sd_np)
    m.fp16_enabled = True
    assert m.get_option(_native.OPT_SMALL_F16) == 1
    out = run(m, clip).cpu()
    ref = _oracle(cfg, sd_np, clip, key=f'lr180_405_4050_{vsr}')
    gt = torch.from_numpy(clip['gt'])
    if vsr:
        gt = gt.repeat_interleave(4, -1).repeat_interleave(4, -2)
    d = float((out - ref).abs().max())
    dp = cpu_ref.clip_psnr(out, gt) - cpu_ref.clip_psnr(ref, gt)
    print(f'180x320 fp16 (vsr={vsr}) vs oracle: max-abs {d:.3e}, PSNR delta {dp:+.2e} dB')
    assert out.shape == ref.shape and torch.isfinite(out).all()
    assert 1e-6 < d < 2e-2 and abs(dp) < 1e-3
    # the small-frame kernel really was the one that ran: the persistent fp16 kernel gives the same bits
    m.set_option(_native.OPT_SMALL_F16, 0)
    assert torch.equal(run(m, clip).cpu(), out)
    # fp16 maps between the launches (block intermediates, and through the whole x4 head: pixel shuffle -> pixel shuffle -> conv_hr
    # -> conv_last) against fp32 storage of the same maps: every one of them is only ever read as an MFMA A operand -> same bits
    m.set_option(_native.OPT_SMALL_F16, 1)
    m.set_option(_native.OPT_F16_MAPS, 0)
    assert torch.equal(run(m, clip).cpu(), out)


@pytest.mark.parametrize('vsr', [False, True], ids=['enhance', 'x4'])
def test_lr180_split_fp16_vs_oracle(vsr):
    """configs[4]'s workload in split fp16 (PNP_PREC_F16X3) against the PINNED ORACLE at the exact-fp32 path's tolerance: the 64->64
    convs on conv3x3_f16x3_kernel (450 tiles: 57 persistent blocks per XCD, one tile each), the x4 heads on the fp32 kernels."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, vsr=vsr)
    sd_np = gu.syn.make_state_dict(cfg, seed=405)
    clip = gu.syn.make_clip(seed=4050, n=1, t=2 if vsr else 3, h=180, w=320, slices='IBBBP', qp_mode='ipb', crf=25, block=4)
    m = build(cfg, sd_np)
    m.precision = 'f16x3'
    out = run(m, clip).cpu()
    ref = _oracle(cfg, sd_np, clip, key=f'lr180_405_4050_{vsr}')
    d = float((out - ref).abs().max())
    print(f'180x320 split fp16 (vsr={vsr}) vs oracle: max-abs {d:.3e}')
    assert out.shape == ref.shape and d < TOL
    m.precision = 'fp32'
    o32 = run(m, clip).cpu()
    assert not torch.equal(o32, out) and float((o32 - ref).abs().max()) < TOL


def _headline_clip():
    """BASELINE configs[2] itself: 7 x 3 x 720 x 1280, IBBBP cadence (I B B B P B B) -- every source pattern of the input conv occurs
    on the persistent-kernel size: sequence ends (zero neighbour), neighbour == key frame (one warped source, summed weights),
    neighbour != key frame (4-source conv with the unwarped neighbour; iconvsr_ipb_par.py:86-90,121-125)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG)
    sd_np = gu.syn.make_state_dict(cfg, seed=404, par_gain=10.0)
    clip = gu.syn.make_clip(seed=4040, n=1, t=7, h=720, w=1280, slices='IBBBP', qp_mode='qp', crf=25, par_classes=3)
    return cfg, sd_np, clip


@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
def test_720p_headline_clip_t7_vs_oracle(precision):
    """The headline workload -- the whole 7-frame 720p clip, not a 2-frame sample and not a crop -- against the pinned oracle
    (iconvsr_ipb_par.py:71-147), exact fp32 and split fp16, at the golden tolerance.  About 2 minutes of CPU on the box's host
    cores, once: both precisions share the oracle's result."""
    cfg, sd_np, clip = _headline_clip()
    m = build(cfg, sd_np)
    m.precision = precision
    out = run(m, clip).cpu()
    ref = _oracle(cfg, sd_np, clip, key='p720_t7_404_4040')
    d = float((out - ref).abs().max())
    per_frame = [float((out[0, i] - ref[0, i]).abs().max()) for i in range(7)]
    print(f'720p T=7 {precision} max|hip - oracle| = {d:.3e}; per frame', ' '.join(f'{v:.1e}' for v in per_frame))
    assert out.shape == (1, 7, 3, 720, 1280) and d < TOL
    gt = torch.from_numpy(clip['gt'])
    assert abs(cpu_ref.clip_psnr(out, gt) - cpu_ref.clip_psnr(ref, gt)) < 1e-3


# ------------------------------------------------------------------------------------------------------------
# host-side behaviour fixed in round 2
# ------------------------------------------------------------------------------------------------------------
def test_param_data_writes_need_invalidate_packed_and_state_dict_loads_do_not():
    """the native weight images are cached on (data_ptr, _version): an in-place write through `.data` changes neither, so
    invalidate_packed() is the documented way to make it visible; load_state_dict() and .to() invalidate by themselves."""
    case = gu.GEN_CASES[1]
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = build(cfg, sd_np)
    a = run(m, clip)
    m.conv_last.bias.data.add_(0.25)            # no _version bump
    m.invalidate_packed()
    b = run(m, clip)
    assert float((b - a - 0.25).abs().max()) < 1e-6
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd['conv_last.bias'] = sd['conv_last.bias'] - 0.25
    m.load_state_dict(sd)                       # invalidates on its own
    assert float((run(m, clip) - a).abs().max()) < 1e-6


def test_unused_side_tensors_may_be_none_like_in_the_reference():
    """use_base_qp=False never reads base_QPs (iconvsr_ipb_par.py:45): None must be accepted; a tensor the configuration
    does read raises TypeError instead of AttributeError."""
    case = [c for c in gu.GEN_CASES if c['name'] == 'gen_nobias_nosoftmax_qp_64x64'][0]
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = build(cfg, sd_np)
    a = {k: torch.from_numpy(v).to(dev()) for k, v in clip.items()}
    with torch.no_grad():
        out = m(a['lq'], a['QPs'], a['slices'], a['mvs'], None, a['partitions'])
        with pytest.raises(TypeError):
            m(a['lq'], None, a['slices'], a['mvs'], None, a['partitions'])
    ref = gu.load_golden(case['name'])['out']
    assert float(np.abs(out.cpu().numpy() - ref).max()) < TOL


def test_sparse_val_is_one_clip_at_a_time():
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, sparse_val=True, num_blocks=1)
    m = build(cfg, gu.syn.make_state_dict(cfg, seed=1))
    clip = gu.syn.make_clip(seed=2, n=2, t=2, h=64, w=64)
    with pytest.raises(NotImplementedError):     # the reference reads feature[0] only (sr_backbone_utils.py:262-275)
        run(m, clip)


def test_sparse_val_follows_the_training_flag():
    """The reference takes sparse_conv only when `self.sparse_val and not self.training` (sr_backbone_utils.py:308,322,
    basicvsr_net.py:511): in train() mode a sparse_val=True model evaluates the dense par * conv1x1 formula, i.e. exactly
    what a sparse_val=False model computes; back in eval() it returns to the sparse golden."""
    case = [c for c in gu.GEN_CASES if c['name'] == 'gen_sparse_val_64x64'][0]
    cfg, sd_np, clip = gu.gen_case_inputs(case)
    m = build(cfg, sd_np)
    ref = gu.load_golden(case['name'])['out']
    dense = run(build(dict(cfg, sparse_val=False), sd_np), clip)
    m.train()
    with torch.no_grad():
        tr = run(m, clip)
    assert torch.equal(tr, dense)
    assert float((tr.cpu() - torch.from_numpy(ref)).abs().max()) > 1e-3          # the two formulas differ on this map
    m.eval()
    assert float(np.abs(run(m, clip).cpu().numpy() - ref).max()) < TOL
    # a batch is fine in train() mode (dense path), refused in eval() mode (the reference's feature[0] indexing)
    two = {k: np.concatenate([v, v]) for k, v in clip.items()}
    m.train()
    assert run(m, two).shape[0] == 2
    m.eval()
    with pytest.raises(NotImplementedError):
        run(m, two)


def test_1080p_vs_oracle_and_2160p_across_precisions():
    """Frames larger than any BASELINE config: 1920x1080 (16,200 tiles) against the pinned oracle in exact fp32 and split fp16, and
    3840x2160 (64,800 tiles; a 64-channel map is 2.1 GB, the 32-bit offset guard's last size class) consistent across the three
    precisions.  T = 2, 2 blocks per branch (the oracle run is ~25 s of CPU)."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    sd_np = gu.syn.make_state_dict(cfg, seed=407, par_gain=10.0)
    for (h, w) in ((1080, 1920), (2160, 3840)):
        clip = gu.syn.make_clip(seed=4070, n=1, t=2, h=h, w=w, slices='IBBBP', block=8, par_classes=3)
        outs = {}
        for prec in ('fp32', 'fp16', 'f16x3'):
            m = build(cfg, sd_np)
            m.precision = prec
            outs[prec] = run(m, clip).cpu()
            assert torch.isfinite(outs[prec]).all(), (h, w, prec)
            del m
        d16 = float((outs['fp16'] - outs['fp32']).abs().max())
        d3 = float((outs['f16x3'] - outs['fp32']).abs().max())
        print(f'{w}x{h}: max|fp16 - fp32| {d16:.2e}, max|f16x3 - fp32| {d3:.2e}')
        assert 1e-7 < d16 < 2e-2 and d3 < TOL
        if h == 1080:
            ref = _oracle(cfg, sd_np, clip)
            assert float((outs['fp32'] - ref).abs().max()) < TOL and float((outs['f16x3'] - ref).abs().max()) < TOL


@pytest.mark.parametrize('precision', ['fp32', 'f16x3'])
@pytest.mark.parametrize('hw', [(128, 128), (720, 1280)], ids=lambda s: '%dx%d' % s)
def test_exact_and_split_paths_are_run_to_run_deterministic_under_unrelated_traffic(precision, hw):
    """conv3x3_persist_kernel / conv3x3_mfma_kernel (fp32) and conv3x3_f16x3_kernel (8x16 tiles at 720p, 4x16 at 128x128): the same
    clip three more times with unrelated work through the device in between must agree bit for bit.  Guards every conv kernel
    against the class of hazard found in round 3's fp16 DCN kernel (hipcc packing gather arithmetic next to MFMA operands: wrong,
    run-to-run varying sums) -- a compiler bump could reintroduce it in any file; the fp16 and DCN kernels have their own such tests."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    sd_np = gu.syn.make_state_dict(cfg, seed=171, par_gain=10.0)
    clip = gu.syn.make_clip(seed=172, n=1, t=3, h=hw[0], w=hw[1], slices='IBBBP', par_classes=3)
    m = build(cfg, sd_np)
    m.precision = precision
    a = {k: torch.from_numpy(v).to(dev()) for k, v in clip.items()}

    def once():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions']).clone()

    first = once()
    assert torch.isfinite(first).all()
    for rep in range(3):
        junk = torch.randn(1 << 24, device=dev())
        junk.sin_().mul_(junk)                               # unrelated traffic and arithmetic between the runs
        assert torch.equal(once(), first), (precision, hw, rep)


def test_split_path_tile_queue_switch_changes_no_bit():
    """PNP_OPT_TILE_QUEUE (default on): the split-fp16 conv launches of a 720p clip with and without the tile queue give the same
    bits, two samples on two streams included (each context has its own queue in the workspace)."""
    from pnp_vcve_amd import _native
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    sd_np = gu.syn.make_state_dict(cfg, seed=173, par_gain=10.0)
    clip = gu.syn.make_clip(seed=174, n=2, t=3, h=720, w=1280, slices='IBBBP', par_classes=3)
    m = build(cfg, sd_np)
    m.precision = 'f16x3'
    assert m.get_option(_native.OPT_TILE_QUEUE) == 1
    with_queue = run(m, clip)
    m.set_option(_native.OPT_TILE_QUEUE, 0)
    assert torch.equal(run(m, clip), with_queue)
    m.set_option(_native.OPT_TILE_QUEUE, 1)
    assert torch.equal(run(m, clip), with_queue)


def test_split_path_tile_queue_under_hipgraph_replay():
    """use_graphs replays a clip as one hipGraph: the queue's per-clip zeroing is a node of it and every launch leaves the counters
    zeroed for the next one -- replays of a 264x272 clip (561 tiles on 512 blocks: the queue is active) equal the eager result.
    (r04: with hipMemsetAsync as that node the SECOND replay read pointer-like garbage as tickets -- wrong frames after a
    five-minute launch; ROCm 7.2, tools/repro/graph_memset_node.py.  The zeroing is a kernel now.)"""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    sd_np = gu.syn.make_state_dict(cfg, seed=175, par_gain=10.0)
    clip = gu.syn.make_clip(seed=176, n=1, t=3, h=264, w=272, slices='IBBBP', par_classes=3)
    m = build(cfg, sd_np)
    m.precision = 'f16x3'
    eager = run(m, clip)
    m.use_graphs = True
    for _ in range(3):
        assert torch.equal(run(m, clip), eager)


@pytest.mark.parametrize('prec,extra,hw,n', [
    ('fp32', {}, (264, 272), 1), ('fp16', {}, (264, 272), 1), ('f16x3', {}, (264, 272), 2), ('f16x3', dict(vsr=True), (136, 144), 1),
    ('fp32', dict(deform='basic'), (136, 144), 1), ('f16x3', dict(blocktype='drt_woqp', num_group=4, flow_inter='nearest'), (264, 272), 1)],
    ids=['fp32', 'fp16', 'f16x3-two-contexts', 'f16x3-vsr', 'fp32-dcn', 'f16x3-variants'])
def test_hipgraph_replays_equal_the_eager_clip_beyond_small_frames(prec, extra, hw, n):
    """use_graphs away from the 128x128 case it was built for: persistent kernels, two samples on two streams, the x4 heads, the
    DCN aligner and the r04 constructor variants -- three replays (new input tensors each) equal the eager result bit for bit."""
    cfg = dict(gu.syn.DEFAULT_GENERATOR_CFG, num_blocks=2, **extra)
    sd_np = gu.syn.make_state_dict(cfg, seed=177, par_gain=10.0)
    clip = gu.syn.make_clip(seed=178, n=n, t=3, h=hw[0], w=hw[1], slices='IBBBP', par_classes=3)
    m = build(cfg, sd_np)
    m.precision = prec
    eager = run(m, clip)
    m.use_graphs = True
    for _ in range(3):
        assert torch.equal(run(m, clip), eager)

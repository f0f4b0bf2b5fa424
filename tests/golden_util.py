"""Shared definitions of the golden-vector cases.

Inputs and weights are regenerated from integer seeds (pnp_vcve_amd.synthetic);
only the *reference outputs* are stored under tests/golden/ (written by
oracle/gen_golden.py, which runs the imported reference in the build
container).  Used by tests/ and by the generator script.
"""
import os

import numpy as np

from pnp_vcve_amd import synthetic as syn

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# ---------------------------------------------------------------- generator
# name, cfg overrides, weight seed, clip kwargs.  par_gain / par_scale are
# chosen so that the partition branch is visible (see 'sensitivity' in the
# manifest written by gen_golden.py).
GEN_CASES = [
    dict(name='gen_t7_ibbbp_64x96', cfg={}, wseed=11, par_gain=10.0,
         clip=dict(seed=101, n=1, t=7, h=64, w=96, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_t7_allB_64x64', cfg={}, wseed=12, par_gain=10.0,
         clip=dict(seed=102, n=1, t=7, h=64, w=64, slices='allB', qp_mode='ipb', crf=35)),
    dict(name='gen_t7_allP_64x64', cfg={}, wseed=13, par_gain=10.0,
         clip=dict(seed=103, n=1, t=7, h=64, w=64, slices='allP', qp_mode='ipb', crf=15)),
    dict(name='gen_t8_64x64', cfg={}, wseed=14, par_gain=10.0,
         clip=dict(seed=104, n=1, t=8, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_t8_mirror_64x64', cfg={}, wseed=14, par_gain=10.0, mirror=True,
         clip=dict(seed=105, n=1, t=8, h=64, w=64, slices=[73, 66, 80, 66, 66, 80, 66, 73], qp_mode='qp', crf=25)),
    dict(name='gen_n2_mixed_64x64', cfg={}, wseed=15, par_gain=10.0,
         clip=dict(seed=106, n=2, t=5, h=64, w=64, slices=[[73, 66, 66, 80, 66], [73, 80, 66, 66, 66]],
                   qp_mode='qp', crf=[15, 35])),
    dict(name='gen_nocat_64x64', cfg=dict(with_cat=False), wseed=16, par_gain=10.0,
         clip=dict(seed=107, n=1, t=5, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_noalignkey_64x64', cfg=dict(align_key=False), wseed=17, par_gain=10.0,
         clip=dict(seed=108, n=1, t=5, h=64, w=64, slices='allP', qp_mode='qp', crf=25)),
    dict(name='gen_vsr_64x64', cfg=dict(vsr=True), wseed=18, par_gain=10.0,
         clip=dict(seed=109, n=1, t=2, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_t7_128x128', cfg={}, wseed=19, par_gain=1.0,
         clip=dict(seed=110, n=1, t=7, h=128, w=128, slices='IBBBP', qp_mode='ipb', crf=25)),
    dict(name='gen_parfloat_72x88', cfg={}, wseed=20, par_gain=1.0,
         clip=dict(seed=111, n=1, t=4, h=72, w=88, slices='IBBBP', qp_mode='qp', crf=25, par_scale=1.0)),
    dict(name='gen_two_layer_64x64', cfg=dict(one_layer=False), wseed=21, par_gain=10.0,
         clip=dict(seed=112, n=1, t=3, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_channel_last_64x64', cfg=dict(channel_first=False), wseed=22, par_gain=10.0,
         clip=dict(seed=113, n=1, t=3, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_nose_64x64', cfg=dict(with_se=False), wseed=23, par_gain=10.0,
         clip=dict(seed=114, n=1, t=3, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_nobias_nosoftmax_qp_64x64',
         cfg=dict(with_bias=False, with_se=False, expert_softmax=False, use_base_qp=False, num_experts=4,
                  num_blocks=3),
         wseed=24, par_gain=10.0,
         clip=dict(seed=115, n=1, t=3, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    # r02: mixed constructor switches in one model (each switch alone is covered above)
    dict(name='gen_nocat_twolayer_72x64', cfg=dict(with_cat=False, one_layer=False, align_key=False), wseed=27, par_gain=10.0,
         clip=dict(seed=118, n=1, t=4, h=72, w=64, slices=[73, 66, 80, 66], qp_mode='qp', crf=35)),
    dict(name='gen_chlast_nose_n2_64x64', cfg=dict(channel_first=False, with_se=False, num_blocks=4), wseed=28, par_gain=10.0,
         clip=dict(seed=119, n=2, t=3, h=64, w=64, slices=[[73, 66, 80], [73, 80, 66]], qp_mode='ipb', crf=[15, 25])),
    dict(name='gen_vsr_nocat_e4_64x64', cfg=dict(vsr=True, with_cat=False, num_experts=4, num_blocks=2), wseed=29, par_gain=10.0,
         clip=dict(seed=120, n=1, t=3, h=64, w=64, slices='allP', qp_mode='qp', crf=25)),
    dict(name='gen_t9_two_keys_64x64', cfg=dict(num_blocks=2), wseed=30, par_gain=10.0,
         clip=dict(seed=121, n=1, t=9, h=64, w=64, slices=[73, 66, 66, 80, 66, 66, 66, 80, 66], qp_mode='qp', crf=25)),
    # r04: the x4 heads with every ingredient visible at >= 1e-3 (gen_vsr_*'s par->0 / base*2 moved the output by less
    # than the old 1e-4 gate): partition branches x100, expert routing x30 -> par->0 2.2e-3, base*2 1.2e-3, allkey 3.3e-3, mvs->0 5e-3
    dict(name='gen_vsr_gain_64x64', cfg=dict(vsr=True), wseed=33, par_gain=100.0, caa_gain=30.0,
         clip=dict(seed=122, n=1, t=3, h=64, w=64, slices=[73, 66, 80], qp_mode='qp', crf=25)),
    # r04: the constructor variants no shipped config uses and rounds 1-3 refused (VERDICT r03 "missing" 6): grouped convs
    # (num_group, sr_backbone_utils.py:285-289), blocktype 'drt_woqp' (both 3x3 convs plain nn.Conv2d, :336-384; runs only with
    # one_layer=True in the reference) and flow_inter='nearest' (flow_warp.py:8,47)
    dict(name='gen_group4_64x64', cfg=dict(num_group=4, num_blocks=3), wseed=34, par_gain=10.0,
         clip=dict(seed=123, n=1, t=3, h=64, w=64, slices=[73, 66, 80], qp_mode='qp', crf=25)),
    dict(name='gen_group2_twolayer_chlast_64x64', cfg=dict(num_group=2, one_layer=False, channel_first=False, num_blocks=3), wseed=35,
         par_gain=10.0, clip=dict(seed=124, n=1, t=3, h=64, w=64, slices='allP', qp_mode='ipb', crf=35)),
    dict(name='gen_woqp_64x64', cfg=dict(blocktype='drt_woqp', num_blocks=3), wseed=36, par_gain=10.0,
         clip=dict(seed=125, n=1, t=3, h=64, w=64, slices=[73, 66, 80], qp_mode='qp', crf=25)),
    dict(name='gen_nearest_64x96', cfg=dict(flow_inter='nearest', num_blocks=3), wseed=37, par_gain=10.0,
         clip=dict(seed=126, n=1, t=5, h=64, w=96, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_woqp_group8_nearest_vsr_64x64', cfg=dict(blocktype='drt_woqp', num_group=8, flow_inter='nearest', vsr=True, num_blocks=2),
         wseed=38, par_gain=10.0, clip=dict(seed=127, n=1, t=3, h=64, w=64, slices=[73, 80, 66], qp_mode='qp', crf=15)),
    # sparse_val=True (eval-time sparse evaluation of the 1x1 branches): maps with NON-binary values and overlapping
    # planes, so that "nonzero -> 1/255, later plane wins" is visible (a one-hot/255 map would equal the dense path)
    dict(name='gen_sparse_val_64x64', cfg=dict(sparse_val=True), wseed=25, par_gain=10.0, par_kind='overlap',
         clip=dict(seed=116, n=1, t=3, h=64, w=64, slices='IBBBP', qp_mode='qp', crf=25)),
    dict(name='gen_sparse_val_channel_last_64x72', cfg=dict(sparse_val=True, channel_first=False), wseed=26,
         par_gain=10.0, par_kind='overlap',
         clip=dict(seed=117, n=1, t=3, h=64, w=72, slices='allP', qp_mode='qp', crf=35)),
]


def overlap_par(seed, shape):
    """partition planes for the sparse_val cases: per 4x4 block each plane is zero with probability 1/2, otherwise a
    float in (0, 1] -- planes overlap and are not binary."""
    n, t, c, h, w = shape
    on = syn.randint(seed, 'par_on', (n, t, c, h // 4, w // 4), 0, 1).astype(np.float32)
    val = syn.uniform(seed, 'par_val', (n, t, c, h // 4, w // 4), 0.05, 1.0)
    blk = on * val
    return np.ascontiguousarray(np.repeat(np.repeat(blk, 4, axis=3), 4, axis=4))


def gen_case_inputs(case):
    """-> (cfg, state-dict (numpy), clip dict (numpy))."""
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    cfg.update(case['cfg'])
    sd = syn.make_state_dict(cfg, seed=case['wseed'], par_gain=case.get('par_gain', 1.0), caa_gain=case.get('caa_gain', 1.0))
    clip = syn.make_clip(**case['clip'])
    if case.get('par_kind') == 'overlap':
        clip['partitions'] = overlap_par(case['clip']['seed'], clip['partitions'].shape)
    if case.get('mirror'):
        # mirror-extended sequence: frame i == frame t-1-i (iconvsr.py:396-410)
        t = clip['lq'].shape[1]
        for i in range(t // 2):
            clip['lq'][:, t - 1 - i] = clip['lq'][:, i]
    return cfg, sd, clip


# ---------------------------------------------------------------- flow_warp
WARP_CASES = [
    dict(name='warp_int_1x16x64x96', seed=201, shape=(1, 16, 64, 96), kind='int'),
    dict(name='warp_frac_1x16x64x96', seed=202, shape=(1, 16, 64, 96), kind='frac'),
    dict(name='warp_oob_2x8x64x64', seed=203, shape=(2, 8, 64, 64), kind='oob'),
    dict(name='warp_block_1x64x40x56', seed=204, shape=(1, 64, 40, 56), kind='block'),
    dict(name='warp_zero_1x4x64x64', seed=205, shape=(1, 4, 64, 64), kind='zero'),
    # r04: interpolation='nearest' (flow_inter); quarter-pel block vectors hit the .5 ties of nearbyint, 'oob' leaves the image
    dict(name='warp_nearest_block_1x64x40x56', seed=206, shape=(1, 64, 40, 56), kind='block', mode='nearest'),
    dict(name='warp_nearest_oob_2x8x64x64', seed=207, shape=(2, 8, 64, 64), kind='oob', mode='nearest'),
    dict(name='warp_nearest_frac_1x16x64x96', seed=208, shape=(1, 16, 64, 96), kind='frac', mode='nearest'),
]


def warp_case_inputs(case):
    n, c, h, w = case['shape']
    s = case['seed']
    x = syn.uniform(s, 'x', (n, c, h, w), -1.0, 1.0)
    kind = case['kind']
    if kind == 'int':
        flow = syn.randint(s, 'flow', (n, h, w, 2), -6, 6).astype(np.float32)
    elif kind == 'frac':
        flow = syn.uniform(s, 'flow', (n, h, w, 2), -8.0, 8.0)
    elif kind == 'oob':
        flow = syn.uniform(s, 'flow', (n, h, w, 2), -1.5 * w, 1.5 * w)
    elif kind == 'block':
        blk = syn.randint(s, 'flow', (n, h // 8, w // 8, 2), -32, 32).astype(np.float32) / 4.0
        flow = np.repeat(np.repeat(blk, 8, axis=1), 8, axis=2)
    elif kind == 'zero':
        flow = np.zeros((n, h, w, 2), np.float32)
    else:
        raise ValueError(kind)
    return x, np.ascontiguousarray(flow)


# ---------------------------------------------------------------- CAA predictors
CAA_QPS = [0.0, 15 / 255.0, 25 / 255.0, 35 / 255.0, 51 / 255.0, 66 / 255.0, 73 / 255.0, 80 / 255.0, 1.0]


# ---------------------------------------------------------------- block / branch
BLOCK_CASES = [
    dict(name='block_par255', wseed=31, par_gain=10.0, seed=301, h=32, w=32, par='onehot255'),
    dict(name='block_parzero', wseed=31, par_gain=10.0, seed=302, h=32, w=32, par='zero'),
    dict(name='block_parfloat', wseed=32, par_gain=1.0, seed=303, h=24, w=40, par='float'),
]


def block_case_inputs(case):
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd = syn.make_state_dict(cfg, seed=case['wseed'], par_gain=case['par_gain'])
    s, h, w = case['seed'], case['h'], case['w']
    x = syn.uniform(s, 'x', (1, 64, h, w), -1.0, 1.0)
    if case['par'] == 'zero':
        par = np.zeros((1, 3, h, w), np.float32)
    elif case['par'] == 'float':
        par = syn.uniform(s, 'par', (1, 3, h, w), 0.0, 1.0)
    else:
        cls = syn.randint(s, 'par', (1, h // 8, w // 8), 0, 3)
        blk = np.stack([(cls == j) for j in range(3)], axis=1).astype(np.float32) / np.float32(255.0)
        par = np.repeat(np.repeat(blk, 8, axis=2), 8, axis=3)
    ew = syn.uniform(s, 'ew', (1, 6), 0.0, 1.0)
    ew = (ew / ew.sum(axis=1, keepdims=True)).astype(np.float32)
    gamma = syn.uniform(s, 'gamma', (1, 64), 0.0, 2.0)
    return cfg, sd, x, np.ascontiguousarray(par), ew, gamma


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


# ---------------------------------------------------------------- side-info rasteriser
RASTER_CASES = [
    dict(name='raster_ibbpbp_48x64', seed=401, h=48, w=64, slices='IBBPBP', per_frame=60),
    dict(name='raster_ippp_64x64', seed=402, h=64, w=64, slices='IPPP', per_frame=90),
]


def raster_case_inputs(case):
    """Synthetic per-frame MV records in the on-disk row format of the reference
    (direction, w, h, x_w, y_w, x, y, motion_x, motion_y, scale), incl. blocks that stick out of the frame."""
    s, h, w = case['seed'], case['h'], case['w']
    slices = case['slices']
    rows, frames = [], []
    sizes = [(16, 16), (16, 8), (8, 16), (8, 8)]
    for f, sl in enumerate(slices):
        if sl == 'I':
            continue
        n = case['per_frame']
        sz = syn.randint(s, f'sz{f}', (n,), 0, 3)
        cx = syn.randint(s, f'cx{f}', (n,), -1, w // 4 + 1) * 4
        cy = syn.randint(s, f'cy{f}', (n,), -1, h // 4 + 1) * 4
        mx = syn.randint(s, f'mx{f}', (n,), -64, 64)
        my = syn.randint(s, f'my{f}', (n,), -64, 64)
        dr = syn.randint(s, f'dr{f}', (n,), 0, 1) * 2 - 1
        for i in range(n):
            bw, bh = sizes[int(sz[i])]
            x, y = int(cx[i]), int(cy[i])
            xw, yw = x + int(mx[i]) // 4, y + int(my[i]) // 4
            rows.append([float(dr[i]), bw, bh, xw, yw, x, y, float(mx[i]), float(my[i]), 4.0])
            frames.append(f)
    return np.array(rows, np.float32), np.array(frames, np.int32), slices, h, w


# ---------------------------------------------------------------- PSNR / tensor2img / BasicVSR.evaluate
METRIC_CASES = [
    dict(name='metrics_psnr_40x56', seed=501, t=6, h=40, w=56),
]


def metric_case_inputs(case):
    """(output, gt) float32 (1, T, 3, h, w) clips for the reference's tensor2img + psnr:
    frame 0: plain values in [0, 1]; frame 1: output leaves [0, 1] on both sides (the clamp); frame 2: every output value
    an exact (k + 0.5) / 255 rounding tie, gt exact k / 255; frame 3: output == gt (PSNR inf); frame 4: ties after the clamp
    (-0.5 / 255, 255.5 / 255) mixed with values a hair off a tie; frame 5: a nearly constant pair (tiny MSE)."""
    s, t, h, w = case['seed'], case['t'], case['h'], case['w']
    gt = syn.uniform(s, 'gt', (1, t, 3, h, w), 0.0, 1.0)
    out = gt + syn.uniform(s, 'noise', (1, t, 3, h, w), -0.06, 0.06)
    out[0, 1] = gt[0, 1] * 1.5 - 0.25
    k = np.floor(syn.uniform(s, 'k', (3, h, w), 0.0, 255.0)).astype(np.float32)
    out[0, 2] = (k + np.float32(0.5)) / np.float32(255.0)
    gt[0, 2] = k / np.float32(255.0)
    out[0, 3] = gt[0, 3]
    sel = syn.uniform(s, 'sel', (3, h, w), 0.0, 4.0)
    out[0, 4] = np.where(sel < 1, np.float32(-0.5 / 255.0), np.where(sel < 2, np.float32(255.5 / 255.0),
                         np.where(sel < 3, np.nextafter((k + np.float32(0.5)) / np.float32(255.0), np.float32(2)),
                                  np.nextafter((k + np.float32(0.5)) / np.float32(255.0), np.float32(-1))))).astype(np.float32)
    gt[0, 5] = np.float32(0.5)
    out[0, 5] = np.float32(0.5)
    out[0, 5, 1, 7, 9] = np.float32(0.5 + 1.5 / 255.0)
    return np.ascontiguousarray(out.astype(np.float32)), np.ascontiguousarray(gt.astype(np.float32))

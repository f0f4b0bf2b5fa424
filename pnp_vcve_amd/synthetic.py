"""Deterministic synthetic clips and weights (numpy only, no torch RNG).

Everything here is a pure function of integer seeds through a counter-based
splitmix64 stream, so the GPU box, this container and the golden-vector
generator (oracle/gen_golden.py) all regenerate bit-identical inputs and
weights and only *outputs* have to be committed as fixtures.

Input conventions follow the reference data layer
(/root/reference/mmedit/datasets/pipelines/loading_ipb.py:223-397 and the
RescaleToZeroOne/FramesToTensor steps of configs/HR_davis_LR_128x128.py:109-131):

  lq        (n,T,3,H,W) fp32 in [0,1]
  mvs       (n,T,4,H,W) fp32 pixels, ch0-1 forward (x,y), ch2-3 backward (x,y),
            block-constant over 8x8 codec partitions, quarter-pel
  partitions(n,T,3,H,W) fp32 one-hot over {16x16,16x8,8x8} divided by 255
  slices    (n,T,1,1,1) fp32 ord('I'|'P'|'B')
  QPs       (n,T,1,1,1) fp32 qp/255 (or ord(slice)/255 for the IPB configs)
  base_QPs  (n,T,1,1,1) fp32 crf/255
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(seed: int, count: int) -> np.ndarray:
    """count 64-bit outputs of the splitmix64 stream started at `seed`."""
    with np.errstate(over='ignore'):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def stream_seed(seed: int, name: str) -> int:
    return (_fnv1a(name) ^ ((seed * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def uniform01(seed: int, name: str, shape) -> np.ndarray:
    """U[0,1) with 24 random mantissa bits, as float32 (exactly representable)."""
    count = int(np.prod(shape)) if len(shape) else 1
    z = _splitmix64(stream_seed(seed, name), count)
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u.reshape(shape)


def uniform(seed, name, shape, lo, hi):
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * uniform01(seed, name, shape)).astype(np.float32)


def normal(seed, name, shape, std=1.0):
    count = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, name + '/u1', (count,)).astype(np.float64)
    u2 = uniform01(seed, name + '/u2', (count,)).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    g = r * np.cos(2.0 * np.pi * u2)
    return (g * std).astype(np.float32).reshape(shape)


def randint(seed, name, shape, lo, hi):
    """integers in [lo, hi] inclusive."""
    count = int(np.prod(shape)) if len(shape) else 1
    z = _splitmix64(stream_seed(seed, name), count)
    span = np.uint64(hi - lo + 1)
    return ((z >> np.uint64(11)) % span).astype(np.int64).reshape(shape) + lo


# --------------------------------------------------------------------------
# generator configuration (kwargs of the reference constructor,
# /root/reference/mmedit/models/backbones/sr_backbones/iconvsr_ipb_par.py:18)
# --------------------------------------------------------------------------
DEFAULT_GENERATOR_CFG = dict(
    mid_channels=64, num_blocks=8, padding=3, with_cat=True, use_base_qp=True,
    num_experts=6, expert_softmax=True, init_weight=True, with_bias=True,
    with_se=True, with_par=True, one_layer=True, blocktype='drt',
    channel_first=True, sparse_val=False, align_key=True, vsr=False)


def state_dict_schema(cfg=None):
    """name -> shape for every parameter of the generator, in the reference's
    state-dict naming."""
    c = dict(DEFAULT_GENERATOR_CFG)
    if cfg:
        c.update(cfg)
    mid, nb, E = c['mid_channels'], c['num_blocks'], c['num_experts']
    with_cat = c.get('with_cat', False)
    sch = {}
    sch['BasePredictor.BaseNet.0.weight'] = (mid, 1)
    sch['BasePredictor.BaseNet.0.bias'] = (mid,)
    sch['BasePredictor.BaseNet.2.weight'] = (E, mid)
    sch['BasePredictor.BaseNet.2.bias'] = (E,)
    if c.get('with_bias', False):
        if c.get('with_se', False):
            sch['BiasePredictor.fc.0.weight'] = (mid // 16, 1)
            sch['BiasePredictor.fc.2.weight'] = (mid, mid // 16)
        else:
            sch['BiasePredictor.qf_embed.0.weight'] = (mid, 1)
            sch['BiasePredictor.qf_embed.0.bias'] = (mid,)
            sch['BiasePredictor.to_gamma.0.weight'] = (mid, mid)
            sch['BiasePredictor.to_gamma.0.bias'] = (mid,)
            sch['BiasePredictor.to_beta.0.weight'] = (mid, mid)
            sch['BiasePredictor.to_beta.0.bias'] = (mid,)
    cin = {'backward_resblocks': (2 if with_cat else 1) * mid + 3,
           'forward_resblocks': (3 if with_cat else 2) * mid + 3}
    for br in ('backward_resblocks', 'forward_resblocks'):
        sch[f'{br}.input_conv.0.weight'] = (mid, cin[br], 3, 3)
        sch[f'{br}.input_conv.0.bias'] = (mid,)
        g = c.get('num_group', 1)                    # every conv of a block is grouped (sr_backbone_utils.py:285-289)
        woqp = c.get('blocktype', 'drt') == 'drt_woqp'      # both 3x3 convs plain nn.Conv2d (:343-344, with one_layer)
        for i in range(nb):
            p = f'{br}.main.{i}.'
            if c.get('one_layer', False):
                sch[p + 'conv1.weight'] = (mid, mid // g, 3, 3)
                sch[p + 'conv1.bias'] = (mid,)
            else:
                sch[p + 'conv1.weight'] = (E, mid, mid // g, 3, 3)
                sch[p + 'conv1.bias'] = (E, mid)
            if woqp and c.get('one_layer', False):
                sch[p + 'conv2.weight'] = (mid, mid // g, 3, 3)
                sch[p + 'conv2.bias'] = (mid,)
            else:
                sch[p + 'conv2.weight'] = (E, mid, mid // g, 3, 3)
                sch[p + 'conv2.bias'] = (E, mid)
            for k in ('conv16x16', 'conv16x8', 'conv8x8'):
                sch[p + k + '.weight'] = (mid, mid // g, 1, 1)
    sch['conv_hr.weight'] = (64, 64, 3, 3)
    sch['conv_hr.bias'] = (64,)
    sch['conv_last.weight'] = (3, 64, 3, 3)
    sch['conv_last.bias'] = (3,)
    if c.get('vsr', False):
        sch['upsample1.upsample_conv.weight'] = (mid * 4, mid, 3, 3)
        sch['upsample1.upsample_conv.bias'] = (mid * 4,)
        sch['upsample2.upsample_conv.weight'] = (64 * 4, mid, 3, 3)
        sch['upsample2.upsample_conv.bias'] = (64 * 4,)
    deform = c.get('deform', 'vos')
    if deform in ('basic', 'fvc'):
        sch['deform_align.weight'] = (mid, mid, 3, 3)
        sch['deform_align.bias'] = (mid,)
        sch['deform_align.conv_offset.0.weight'] = (mid, mid + 2, 3, 3)
        sch['deform_align.conv_offset.0.bias'] = (mid,)
        sch['deform_align.conv_offset.2.weight'] = (16 * 9 * 3, mid, 3, 3)
        sch['deform_align.conv_offset.2.bias'] = (16 * 9 * 3,)
    return sch


def make_state_dict(cfg=None, seed=0, par_gain=1.0, caa_gain=1.0):
    """Seeded "trained-like" weights: magnitudes follow the reference's
    initialisers but biases are non-zero and the 1x1
    partition branches are scaled by `par_gain` so that every term of the
    block is numerically visible in a parity check.  `caa_gain` scales the
    weights of the expert-routing predictor (Base_Predictor), i.e. how strongly
    the expert mixture follows the base QP; 1.0 leaves every earlier fixture as it was."""
    sch = state_dict_schema(cfg)
    sd = {}
    for name, shape in sch.items():
        if name.endswith('.bias'):
            sd[name] = uniform(seed, name, shape, -0.05, 0.05)
            continue
        fan_in = int(np.prod(shape[-3:])) if len(shape) >= 4 else shape[-1]
        if '.conv2.weight' in name or ('.conv1.weight' in name and len(shape) == 5):
            b = np.sqrt(6.0 / fan_in)            # kaiming_uniform_, a=0
            sd[name] = uniform(seed, name, shape, -b, b)
        elif any(k in name for k in ('conv16x16', 'conv16x8', 'conv8x8')):
            sd[name] = normal(seed, name, shape, std=par_gain * np.sqrt(2.0 / fan_in))
        elif '.conv1.weight' in name:
            sd[name] = normal(seed, name, shape, std=0.1 * np.sqrt(2.0 / fan_in))
        elif name.startswith('BiasePredictor.fc.'):
            sd[name] = uniform(seed, name, shape, -2.0, 2.0)
        elif 'conv_offset.2' in name:
            sd[name] = normal(seed, name, shape, std=0.02 * np.sqrt(2.0 / fan_in))
        else:                                     # torch default: U(-1/sqrt(fan_in), +)
            b = 1.0 / np.sqrt(fan_in)
            sd[name] = uniform(seed, name, shape, -b, b)
            if caa_gain != 1.0 and name.startswith('BasePredictor.') and name.endswith('.weight'):
                sd[name] = (sd[name] * np.float32(caa_gain)).astype(np.float32)
    return sd


SLICE_PATTERNS = {
    'IBBBPBB': [73, 66, 66, 66, 80, 66, 66],
    'allB': None, 'allP': None,
}


def slice_pattern(name_or_list, t):
    if isinstance(name_or_list, (list, tuple)):
        s = list(name_or_list)
        assert len(s) == t
        return s
    if name_or_list == 'allB':
        return [66] * t
    if name_or_list == 'allP':
        return [80] * t
    if name_or_list == 'IBBBP':     # x264 bframes=3 cadence, I first
        return [73 if i == 0 else (80 if i % 4 == 0 else 66) for i in range(t)]
    raise ValueError(name_or_list)


def make_clip(seed, n=1, t=7, h=128, w=128, slices='IBBBP', qp_mode='qp', crf=25,
              par_scale=1.0 / 255.0, mv_range=32, block=8, par_classes=4):
    """One synthetic batch of clips with the SURVEY.md section 8(d) distributions.

    slices: pattern name / list (same for all samples) or list of n of them.
    qp_mode: 'qp' -> per-frame QP in 20..40 /255 ; 'ipb' -> ord(slice)/255.
    crf: int or list of n ints (base_QPs = crf/255, constant over T).
    par_scale: value of the active one-hot partition plane (reference: 1/255).
    par_classes: 3 = SURVEY 8(d)'s U{0,1,2} (every block of a P/B frame carries a partition record; bench.py);
                 4 = a quarter of the blocks carry none (all-zero) -- the golden fixtures were drawn with this.
    """
    # (a frame that is not a multiple of the block size: the block grid starts at the origin and the frame cuts its last row / column --
    #  what a decoder's cropping does to a coded picture, e.g. 180 rows out of 192)
    bh, bw = (h + block - 1) // block, (w + block - 1) // block
    lq = uniform01(seed, 'lq', (n, t, 3, h, w))
    noise = normal(seed, 'gt_noise', (n, t, 3, h, w), std=0.02)
    gt = np.clip(lq + noise, 0.0, 1.0).astype(np.float32)

    mv_blk = randint(seed, 'mv', (n, t, 4, bh, bw), -mv_range, mv_range).astype(np.float32) / 4.0
    cls = randint(seed, 'par', (n, t, bh, bw), 0, par_classes - 1)       # 3 -> no record (all-zero)
    par_blk = np.zeros((n, t, 3, bh, bw), np.float32)
    for j in range(3):
        par_blk[:, :, j] = (cls == j).astype(np.float32) * np.float32(par_scale)

    if isinstance(slices, str) or all(isinstance(s, (int, np.integer)) for s in slices):
        per_sample = [slice_pattern(slices, t)] * n
    else:
        assert len(slices) == n
        per_sample = [slice_pattern(s, t) for s in slices]
    sl = np.array(per_sample, np.float32).reshape(n, t, 1, 1, 1)

    # I frames carry no motion / partition records (loading_ipb.py:328-369)
    is_i = (sl[:, :, 0, 0, 0] == 73)
    mv_blk[is_i] = 0.0
    par_blk[is_i] = 0.0
    mvs = np.repeat(np.repeat(mv_blk, block, axis=3), block, axis=4)[..., :h, :w]
    par = np.repeat(np.repeat(par_blk, block, axis=3), block, axis=4)[..., :h, :w]

    if qp_mode == 'qp':
        qps = randint(seed, 'qp', (n, t), 20, 40).astype(np.float32) / np.float32(255.0)
    elif qp_mode == 'ipb':
        qps = sl[:, :, 0, 0, 0] / np.float32(255.0)
    else:
        raise ValueError(qp_mode)
    qps = qps.reshape(n, t, 1, 1, 1).astype(np.float32)
    crfs = np.array(crf if isinstance(crf, (list, tuple)) else [crf] * n, np.float32)
    base = (np.repeat(crfs[:, None], t, axis=1) / np.float32(255.0)).reshape(n, t, 1, 1, 1).astype(np.float32)
    return dict(lq=lq, gt=gt, mvs=np.ascontiguousarray(mvs), partitions=np.ascontiguousarray(par),
                slices=sl, QPs=qps, base_QPs=base)


# --------------------------------------------------------------------------
# on-disk clips in the reference's directory layout (for the end-to-end pipeline benchmark and the dataset tests)
# --------------------------------------------------------------------------
def make_mv_records(seed, t, h, w, slices, block=16):
    """Decoder-style MV records of a clip, rows (direction, w, h, x_w, y_w, x, y, motion_x, motion_y, scale) as
    LoadImageFromFileList_ipb reads them (mmedit/datasets/pipelines/loading_ipb.py:328-369): every P / B frame is tiled by
    `block` x `block` partitions split at random into 16x16 / 16x8 / 8x16 / 8x8 records with quarter-pel vectors
    (h * w / 256 ... / 64 records per frame: thousands at 720p).  -> (records float32 (R, 10), frame int32 (R,))."""
    rows, frames = [], []
    gy, gx = (h + block - 1) // block, (w + block - 1) // block
    for f in range(t):
        if slices[f] == 'I':
            continue
        kind = randint(seed, f'kind{f}', (gy, gx), 0, 3)
        mx = randint(seed, f'mx{f}', (gy, gx, 4), -32, 32).astype(np.float32)
        my = randint(seed, f'my{f}', (gy, gx, 4), -32, 32).astype(np.float32)
        dr = (randint(seed, f'dr{f}', (gy, gx, 4), 0, 1) * 2 - 1).astype(np.float32)
        parts = {0: [(0, 0, 16, 16)], 1: [(0, 0, 16, 8), (0, 8, 16, 8)], 2: [(0, 0, 8, 16), (8, 0, 8, 16)],
                 3: [(0, 0, 8, 8), (8, 0, 8, 8), (0, 8, 8, 8), (8, 8, 8, 8)]}
        by, bx = np.mgrid[0:gy, 0:gx]
        for k, plist in parts.items():
            sel = kind == k
            for pi, (ox, oy, bw, bh) in enumerate(plist):
                x, y = (bx[sel] * block + ox).astype(np.float32), (by[sel] * block + oy).astype(np.float32)
                vx, vy = mx[..., pi][sel], my[..., pi][sel]
                n = x.shape[0]
                r = np.stack([dr[..., pi][sel], np.full(n, bw, np.float32), np.full(n, bh, np.float32),
                              x + np.trunc(vx / 4.0), y + np.trunc(vy / 4.0), x, y, vx, vy,
                              np.full(n, 4.0, np.float32)], axis=1)
                rows.append(r.astype(np.float32))
                frames.append(np.full(n, f, np.int32))
    if not rows:
        return np.zeros((0, 10), np.float32), np.zeros((0,), np.int32)
    return np.concatenate(rows), np.concatenate(frames)



def write_clip_tree(root, clips=('000', '011'), t=7, h=128, w=128, crf=25, slices='IBBBP', seed=0):
    """Write `clips` synthetic clips under `root` in the reference's REDS layout and return (lq_folder, gt_folder,
    qp_slice_file): <root>/crfXX/png/<clip>/%08d.png, .../mv/<clip>/%08d.npy, <root>/X4/png/<clip>/%08d.png and the JSON
    QP / slice table (configs/HR_davis_LR_128x128.py:109-131, loading_ipb.py:298-316).  Frames are smooth (fast to encode)."""
    import json
    import os
    from PIL import Image
    cdir = f'crf{crf}'
    if isinstance(slices, str) and slices not in ('IBBBP', 'allB', 'allP'):
        sl = list(slices[:t])                          # explicit letters, e.g. 'IBPB'
    else:
        sl = [chr(int(v)) for v in slice_pattern(slices, t)]
        sl[0] = 'I' if slices == 'IBBBP' else sl[0]
    table = {cdir: {}}
    yy, xx = np.mgrid[0:h, 0:w]
    for ci, clip in enumerate(clips):
        png, mv, gt = (os.path.join(root, cdir, 'png', clip), os.path.join(root, cdir, 'mv', clip), os.path.join(root, 'X4', 'png', clip))
        for d in (png, mv, gt):
            os.makedirs(d, exist_ok=True)
        rec, rf = make_mv_records(seed * 1009 + ci, t, h, w, sl)
        table[cdir][clip] = {}
        for f in range(t):
            base = np.stack([(xx * (2 + ci) + yy * 3 + 17 * f) % 256, (xx + yy * (2 + f)) % 256, (xx * 5 + yy + 40 * ci) % 256], axis=-1)
            img = base.astype(np.uint8)
            Image.fromarray(img).save(os.path.join(gt, f'{f:08d}.png'), compress_level=1)
            Image.fromarray((img // 4 * 4).astype(np.uint8)).save(os.path.join(png, f'{f:08d}.png'), compress_level=1)
            np.save(os.path.join(mv, f'{f:08d}.npy'), rec[rf == f])
            table[cdir][clip][str(f)] = {'slice': sl[f], 'QP': 22 + (f % 12)}
    qp = os.path.join(root, 'qp.json')
    with open(qp, 'w') as fq:
        json.dump(table, fq)
    return os.path.join(root, cdir, 'png'), os.path.join(root, 'X4', 'png'), qp

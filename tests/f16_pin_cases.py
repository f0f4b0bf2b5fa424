"""Cases that pin the fp16-operand path bit for bit across builds.

tools/gen_f16_pins.py ran these on an MI355X with the ROUND-2 build (commit 340801c: fp32 maps rounded by their readers, input
conv as a launch chain, no branch skipping on the fp16 kernels) and stored a SHA-256 of every output in
tests/golden/f16_pins_r02.json; tests/test_gpu_fp16.py recomputes them with the current build.  Only APIs both builds have are
used.  -0.0 is canonicalised to +0.0 before hashing (skipping an all-zero partition branch can flip the sign of a zero sum).
"""
import hashlib

import numpy as np

PIN_CASES = [
    dict(name='default_64x96_t4', cfg={}, wseed=201, clip=dict(seed=301, n=1, t=4, h=64, w=96)),
    dict(name='default_t1_64x64', cfg={}, wseed=202, clip=dict(seed=302, n=1, t=1, h=64, w=64)),
    dict(name='parfloat_72x88', cfg={}, wseed=203, clip=dict(seed=303, n=1, t=4, h=72, w=88, par_scale=1.0)),
    dict(name='nocat_noalign_68x100', cfg=dict(with_cat=False, align_key=False), wseed=204, clip=dict(seed=304, n=1, t=3, h=68, w=100)),
    dict(name='channel_last_two_layer_n2', cfg=dict(channel_first=False, one_layer=False, num_blocks=2), wseed=205,
         clip=dict(seed=305, n=2, t=3, h=64, w=64, crf=[15, 35])),
    dict(name='vsr_64x80', cfg=dict(vsr=True, num_blocks=2), wseed=206, clip=dict(seed=306, n=1, t=2, h=64, w=80)),
    dict(name='lr180_t3', cfg={}, wseed=207, clip=dict(seed=307, n=1, t=3, h=180, w=320)),
    dict(name='pair_tiles_128x512', cfg=dict(num_blocks=2), wseed=208, clip=dict(seed=308, n=1, t=3, h=128, w=512)),
    dict(name='p720_blocks2_t3', cfg=dict(num_blocks=2), wseed=209, clip=dict(seed=309, n=1, t=3, h=720, w=1280)),
    dict(name='p720_default_t2', cfg={}, wseed=210, clip=dict(seed=310, n=1, t=2, h=720, w=1280)),
]


def run_case(case, synthetic, build_backbone, torch):
    """-> fp16-path output (CUDA tensor) of the case with the package the caller imported"""
    cfg = dict(synthetic.DEFAULT_GENERATOR_CFG, **case['cfg'])
    sd = synthetic.make_state_dict(cfg, seed=case['wseed'], par_gain=10.0)
    kw = dict(slices='IBBBP', block=4, par_classes=3, qp_mode='ipb', crf=25)
    kw.update(case['clip'])
    clip = synthetic.make_clip(**kw)
    m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    m.fp16_enabled = True
    a = {k: torch.from_numpy(v).cuda() for k, v in clip.items()}
    with torch.no_grad():
        return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])


def digest(out):
    """SHA-256 of the output with -0.0 canonicalised to +0.0, plus a few values for a readable mismatch report"""
    x = (out + 0.0).float().cpu().numpy()
    flat = x.reshape(-1)
    return dict(sha256=hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest(), shape=list(x.shape),
                mean=float(flat.astype(np.float64).mean()), first=[float(v) for v in flat[:4]],
                last=[float(v) for v in flat[-4:]])

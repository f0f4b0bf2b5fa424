"""ORACLE -- test infrastructure only, never the product path.

A flat CPU (PyTorch fp32) restatement of the PnP-VCVE BAE/CAA forward hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg
may import this module; the product (pnp_vcve_amd) never does and fails loudly
when its HIP library is missing.

Parity pinning: the reference ships no tests or golden vectors (SURVEY.md
section 4), so this restatement is pinned against outputs of the *imported
reference itself* run in the build container (oracle/gen_golden.py ->
tests/golden/*.npz, checked by tests/test_oracle_golden.py).

Every function cites the reference lines it restates; paths are relative to
/root/reference/mmedit/models/.
"""
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# K1  MV-guided bilinear alignment
# --------------------------------------------------------------------------
def flow_warp(x, flow, interpolation='bilinear'):
    """common/flow_warp.py:6-50 (bilinear or nearest, zeros padding, align_corners=True)
    as used by VOSAlignment.forward, backbones/sr_backbones/iconvsr_mv.py:12-18 (interpolation = the
    constructor's flow_inter, iconvsr_ipb.py:16-24).

    x (n,c,h,w); flow (n,h,w,2) in pixels, last dim = (dx, dy).
    Written as an explicit 4-tap gather (not F.grid_sample) so that it is an
    independent restatement; the coordinate arithmetic keeps the reference's
    normalise (flow_warp.py:41-42) / ATen un-normalise round trip.
    """
    if x.shape[-2:] != flow.shape[1:3]:
        raise ValueError(f'The spatial sizes of input ({x.shape[-2:]}) and '
                         f'flow ({flow.shape[1:3]}) are not the same.')
    n, c, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h, dtype=x.dtype), torch.arange(w, dtype=x.dtype), indexing='ij')
    px = gx.unsqueeze(0) + flow[..., 0]
    py = gy.unsqueeze(0) + flow[..., 1]
    nx = 2.0 * px / max(w - 1, 1) - 1.0
    ny = 2.0 * py / max(h - 1, 1) - 1.0
    ix = ((nx + 1) / 2) * (w - 1)          # ATen grid_sampler_unnormalize, align_corners=True
    iy = ((ny + 1) / 2) * (h - 1)
    if interpolation == 'nearest':
        # ATen grid_sampler_2d, GridSamplerInterpolation::Nearest: std::nearbyint of the un-normalised coordinate (ties to
        # even, which is what torch.round does), the pixel if it lies inside the image, else 0
        xn, yn = torch.round(ix), torch.round(iy)
        ok = (xn >= 0) & (xn <= w - 1) & (yn >= 0) & (yn <= h - 1)
        idx = (yn.clamp(0, h - 1) * w + xn.clamp(0, w - 1)).long().reshape(n, 1, h * w).expand(n, c, h * w)
        v = torch.gather(x.reshape(n, c, h * w), 2, idx).reshape(n, c, h, w)
        return v * ok.to(x.dtype).unsqueeze(1)
    if interpolation != 'bilinear':
        raise NotImplementedError(f'flow_warp: interpolation {interpolation!r}')
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    xf = x.reshape(n, c, h * w)

    def tap(xx, yy, wt):
        ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
        idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long().reshape(n, 1, h * w).expand(n, c, h * w)
        v = torch.gather(xf, 2, idx).reshape(n, c, h, w)
        return v * (wt * ok.to(x.dtype)).unsqueeze(1)

    return tap(x0, y0, w_nw) + tap(x1, y0, w_ne) + tap(x0, y1, w_sw) + tap(x1, y1, w_se)


# --------------------------------------------------------------------------
# K12 (optional aligners) modulated deformable conv, restated from the
# published mmcv-full 1.3.13..1.6 semantics (mmcv/ops/csrc/common/
# modulated_deform_conv_*; not vendored in the reference -> PARITY UNPINNED).
# --------------------------------------------------------------------------
def modulated_deform_conv2d(x, offset, mask, weight, bias, deform_groups=16):
    """mmcv.ops.modulated_deform_conv2d as called by backbones/sr_backbones/iconvsr_mv.py:39-41,82-84
    (ModulatedDeformConv2d subclass, :21-30,52-61).  mmcv-full 1.x is NOT vendored in /root/reference: this restates
    its published CUDA semantics (modulated_deform_conv_cuda_kernel.cuh: modulated_deformable_im2col +
    dmcn_im2col_bilinear) -- PARITY UNPINNED against mmcv itself.
    3x3, stride 1, pad 1, dilation 1, groups 1.
    offset (n, dg*2*9, h, w) interleaved (dy, dx) per tap per deform group;
    mask (n, dg*9, h, w) already sigmoid-ed.  Bilinear taps outside the image
    contribute zero (mmcv dmcn_im2col_bilinear: sample is 0 unless
    h_im > -1 && w_im > -1 && h_im < H && w_im < W, with per-corner checks)."""
    n, c, h, w = x.shape
    dg = deform_groups
    cpg = c // dg
    gy, gx = torch.meshgrid(torch.arange(h, dtype=x.dtype), torch.arange(w, dtype=x.dtype), indexing='ij')
    cols = []
    off = offset.reshape(n, dg, 9, 2, h, w)
    msk = mask.reshape(n, dg, 9, h, w)
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg].reshape(n, cpg, h * w)
        taps = []
        for k in range(9):
            ky, kx = k // 3 - 1, k % 3 - 1
            py = gy + ky + off[:, g, k, 0]
            px = gx + kx + off[:, g, k, 1]
            y0 = torch.floor(py)
            x0 = torch.floor(px)
            ly, lx = py - y0, px - x0
            inside = (py > -1) & (px > -1) & (py < h) & (px < w)
            val = 0
            for (yy, xx, wt) in ((y0, x0, (1 - ly) * (1 - lx)), (y0, x0 + 1, (1 - ly) * lx),
                                 (y0 + 1, x0, ly * (1 - lx)), (y0 + 1, x0 + 1, ly * lx)):
                ok = (yy >= 0) & (yy <= h - 1) & (xx >= 0) & (xx <= w - 1) & inside
                idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long().reshape(n, 1, h * w).expand(n, cpg, h * w)
                v = torch.gather(xg, 2, idx).reshape(n, cpg, h, w)
                val = val + v * (wt * ok.to(x.dtype)).unsqueeze(1)
            taps.append(val * msk[:, g, k].unsqueeze(1))
        cols.append(torch.stack(taps, dim=2))            # (n, cpg, 9, h, w)
    col = torch.cat(cols, dim=1).reshape(n, c * 9, h * w)
    out = torch.matmul(weight.reshape(weight.shape[0], -1), col).reshape(n, -1, h, w)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def deform_align(sd, cfg, feat, flow_nchw):
    """iconvsr_ipb.py:19-24 dispatch + iconvsr_mv.py:12-84."""
    mode = cfg.get('deform', 'vos')
    inter = cfg.get('flow_inter', 'bilinear')
    if mode == 'vos':
        return flow_warp(feat, flow_nchw.permute(0, 2, 3, 1), inter)
    p = 'deform_align.'
    if mode == 'basic':                                   # iconvsr_mv.py:68-84
        warped = flow_warp(feat, flow_nchw.permute(0, 2, 3, 1), inter)
        extra = torch.cat([warped, flow_nchw], dim=1)
    elif mode == 'fvc':                                   # iconvsr_mv.py:31-41
        extra = torch.cat([feat, flow_nchw], dim=1)
    else:
        raise TypeError('Not such DCN type')
    o = F.conv2d(extra, sd[p + 'conv_offset.0.weight'], sd[p + 'conv_offset.0.bias'], padding=1)
    o = F.leaky_relu(o, 0.1)
    o = F.conv2d(o, sd[p + 'conv_offset.2.weight'], sd[p + 'conv_offset.2.bias'], padding=1)
    o1, o2, m = torch.chunk(o, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    if mode == 'basic':                                   # :76-77 (tanh-clamped offset is computed but unused)
        offset = offset + flow_nchw.flip(1).repeat(1, offset.size(1) // 2, 1, 1)
    return modulated_deform_conv2d(feat, offset, torch.sigmoid(m), sd[p + 'weight'], sd[p + 'bias'], 16)


# --------------------------------------------------------------------------
# K9  CAA hyper-network
# --------------------------------------------------------------------------
def base_predictor(sd, q, softmax):
    """backbones/sr_backbones/domain_aware.py:172-183.  q (b,t,1,1,1) -> (b,t,E)."""
    b, t = q.shape[:2]
    p = 'BasePredictor.BaseNet.'
    h = F.relu(F.linear(q.reshape(-1, 1), sd[p + '0.weight'], sd[p + '0.bias']))
    o = F.linear(h, sd[p + '2.weight'], sd[p + '2.bias'])
    if softmax:
        o = torch.softmax(o, dim=1)
    return o.view(b, t, -1)


def bias_predictor(sd, cfg, q):
    """SEModule domain_aware.py:210-222 (+Hsigmoid :201-207: relu6(x+3)/3) when
    with_se, else Bias_Predictor :185-199.  Returns (gamma, beta)."""
    b, t = q.shape[:2]
    p = 'BiasePredictor.'
    if cfg.get('with_se', False):
        h = F.relu(F.linear(q.reshape(-1, 1), sd[p + 'fc.0.weight']))
        g = F.relu6(F.linear(h, sd[p + 'fc.2.weight']) + 3.0) / 3.0
        return g.view(b, t, -1), None
    e = F.relu(F.linear(q.reshape(-1, 1), sd[p + 'qf_embed.0.weight'], sd[p + 'qf_embed.0.bias'])).view(b, t, -1)
    g = torch.sigmoid(F.linear(e, sd[p + 'to_gamma.0.weight'], sd[p + 'to_gamma.0.bias']))
    be = torch.tanh(F.linear(e, sd[p + 'to_beta.0.weight'], sd[p + 'to_beta.0.bias']))
    return g, be


# --------------------------------------------------------------------------
# K3/K4  expert-mixture ("dynamic") conv
# --------------------------------------------------------------------------
def dynamic_conv_se(x, ew, weight, bias, gamma, with_se, groups=1):
    """common/sr_backbone_utils.py:193-209 (Dynamic_conv2d_se.forward).
    x (b,64,h,w); ew (b,E); weight (E,64,64/groups,3,3); bias (E,64); gamma (b,64)."""
    b, c, h, w = x.shape
    E = weight.shape[0]
    wagg = torch.mm(ew, weight.view(E, -1)).view(b * c, c // groups, 3, 3)
    bagg = torch.mm(ew, bias).view(-1)
    out = F.conv2d(x.reshape(1, b * c, h, w), wagg, bagg, padding=1, groups=groups * b).view(b, c, h, w)
    if with_se:
        out = out * gamma.unsqueeze(-1).unsqueeze(-1)
    return out


# --------------------------------------------------------------------------
# a13  sparse evaluation of the 1x1 partition branches (sparse_val=True, eval only)
# --------------------------------------------------------------------------
def sparse_conv(sd, prefix, feature, par):
    """common/sr_backbone_utils.py:294-302 (sparse_conv) with mask_roi / mask_roi_back (:262-275) and the index
    lists of backbones/sr_backbones/basicvsr_net.py:456-476,511-514 (generate_indices(par[:, j], 1)).
    par (1,3,1,h,w).  Branch j is evaluated at the pixels where plane j is NONZERO (its value is not used), the
    three results are scattered in the order 16x16, 16x8, 8x8 (a later plane overwrites an earlier one) and the
    map is divided by 255.  Only sample 0 is touched (`feature[0, ...]`): one clip at a time."""
    assert feature.shape[0] == 1, 'sparse_val: the reference indexes sample 0 only'
    dy = torch.zeros_like(feature)
    for j, name in enumerate(('conv16x16', 'conv16x8', 'conv8x8')):
        idx = torch.nonzero(par[:, j].squeeze())           # generate_indices, kernel_size == 1
        h_idx, w_idx = idx[:, 0], idx[:, 1]
        res = torch.mm(sd[prefix + name + '.weight'].view(64, -1), feature[0, :, h_idx, w_idx])
        dy[0, :, h_idx, w_idx] = res
    return dy / 255


# --------------------------------------------------------------------------
# K5/K6  one BAE block
# --------------------------------------------------------------------------
def bae_block(sd, cfg, prefix, x, par, ew, gamma):
    """common/sr_backbone_utils.py:304-333 (ResidualBlockNoBNDynamic_drt.forward) and, for blocktype 'drt_woqp', :366-384
    (ResidualBlockNoBNDynamic_drt_wo_qp.forward: both 3x3 convs are called on the bare map, which only a plain nn.Conv2d
    accepts -- i.e. one_layer=True; with one_layer=False the reference indexes a tensor with 'x' and raises).
    par (b,3,1,h,w).  Every conv of the block has groups = num_group (:285-289)."""
    with_se = cfg.get('with_se', False)
    one_layer = cfg.get('one_layer', False)
    g = cfg.get('num_group', 1)
    woqp = cfg.get('blocktype', 'drt') == 'drt_woqp'
    if woqp and not one_layer:
        raise IndexError("too many indices for tensor of dimension 4")     # what inputs['x'] on a tensor raises (:376)

    def dyres(v):                                          # :310 / :324
        if cfg.get('sparse_val', False):                   # eval-only sparse evaluation, :308-309 / :322-323
            return sparse_conv(sd, prefix, v, par)
        return (F.conv2d(v, sd[prefix + 'conv16x16.weight'], groups=g) * par[:, 0] +
                F.conv2d(v, sd[prefix + 'conv16x8.weight'], groups=g) * par[:, 1] +
                F.conv2d(v, sd[prefix + 'conv8x8.weight'], groups=g) * par[:, 2])

    def conv1(v):
        if one_layer:
            return F.conv2d(v, sd[prefix + 'conv1.weight'], sd[prefix + 'conv1.bias'], padding=1, groups=g)
        return dynamic_conv_se(v, ew, sd[prefix + 'conv1.weight'], sd[prefix + 'conv1.bias'], gamma, with_se, g)

    def conv2(v):
        if woqp:                                           # :343-344: a plain conv too, no expert mix, no gain
            return F.conv2d(v, sd[prefix + 'conv2.weight'], sd[prefix + 'conv2.bias'], padding=1, groups=g)
        return dynamic_conv_se(v, ew, sd[prefix + 'conv2.weight'], sd[prefix + 'conv2.bias'], gamma, with_se, g)

    if cfg.get('channel_first', True):                     # :305-313
        out = F.relu(conv2(x) + dyres(x))
        out = conv1(out)
    else:                                                  # :314-327
        out = F.relu(conv1(x))
        out = conv2(out) + dyres(out)
    return x + out                                         # :329, res_scale = 1


def resblocks(sd, cfg, branch, x, par, ew, gamma):
    """backbones/sr_backbones/basicvsr_net.py:506-519
    (ResidualBlocksWithInputConvDynamic_drt.forward)."""
    b, c, h, w = par.shape
    par5 = par.view(b, c, 1, h, w)
    x = F.leaky_relu(F.conv2d(x, sd[f'{branch}.input_conv.0.weight'], sd[f'{branch}.input_conv.0.bias'], padding=1), 0.1)
    for i in range(cfg['num_blocks']):
        x = bae_block(sd, cfg, f'{branch}.main.{i}.', x, par5, ew, gamma)
    return x


def pixel_shuffle_pack(sd, prefix, x):
    """common/upsample.py:40-51."""
    x = F.conv2d(x, sd[prefix + 'upsample_conv.weight'], sd[prefix + 'upsample_conv.bias'], padding=1)
    return F.pixel_shuffle(x, 2)


# --------------------------------------------------------------------------
# a1  the generator forward
# --------------------------------------------------------------------------
def generator_forward(sd, cfg, lrs, QPs, slices, mvs, base_QPs, par_map):
    """backbones/sr_backbones/iconvsr_ipb_par.py:44-149.

    sd: dict name -> torch.float32 tensor (reference state-dict schema);
    cfg: constructor kwargs.  Returns (n,t,3,H,W) (x4 when cfg['vsr'])."""
    mid = cfg.get('mid_channels', 64)
    with_cat = cfg.get('with_cat', False)
    align_key = cfg.get('align_key', False)
    with_bias = cfg.get('with_bias', False)
    used = base_QPs if cfg.get('use_base_qp', False) else QPs            # :45
    ew_all = base_predictor(sd, used, cfg.get('expert_softmax', False))  # :46
    if with_bias:
        gammas, _ = bias_predictor(sd, cfg, QPs)                         # :47-48
    n, t, c, h, w = lrs.shape
    assert h >= 64 and w >= 64, (
        f'The height and width of inputs should be at least 64, but got {h} and {w}.')
    if h % 4 or w % 4:
        # iconvsr.py:371-394 pads only lrs; mvs/par stay unpadded so the reference
        # raises in flow_warp.py:27-29.  Same error type here.
        raise ValueError('spatial size must be a multiple of 4')
    # iconvsr.py:396-410 mirror detection only switches compute_flow
    # (iconvsr_ipb.py:33-46) to an indexing that selects the same MV maps:
    # flows_backward[-i] == mvs[:, i, 0:2] == flows_forward[i-1].
    flows_forward = mvs[:, 1:, 0:2]
    flows_backward = mvs[:, :t - 1, 2:4]
    key = (slices[:, :, 0, 0, 0] == 73) | (slices[:, :, 0, 0, 0] == 80)  # :60-62
    key = key.clone()
    key[:, -1] = True
    key[:, 0] = True

    outputs = [None] * t
    zeros = lrs.new_zeros(n, mid, h, w)
    key_warp, neighbor = zeros, zeros
    for i in range(t - 1, -1, -1):                                       # :71-100
        lr = lrs[:, i]
        if i < t - 1:
            kws, nbs = [], []
            for b in range(n):
                k = i + 1 + int(torch.where(key[b, i + 1:])[0][0])
                kf = deform_align(sd, cfg, outputs[k][b:b + 1], flows_backward[b:b + 1, i])
                kws.append(kf)
                nbs.append(kf if (align_key and k == i + 1) else outputs[i + 1][b:b + 1])
            key_warp, neighbor = torch.cat(kws), torch.cat(nbs)
        feat = torch.cat([lr, key_warp, neighbor], 1) if with_cat else torch.cat([lr, key_warp], 1)
        gamma = gammas[:, i] if with_bias else None
        outputs[i] = resblocks(sd, cfg, 'backward_resblocks', feat, par_map[:, i], ew_all[:, i], gamma)

    outs = []
    key_warp, neighbor = zeros, zeros
    for i in range(t):                                                   # :103-147
        lr = lrs[:, i]
        if i > 0:
            kws, nbs = [], []
            for b in range(n):
                k = int(torch.where(key[b, :i])[0][-1])
                kf = deform_align(sd, cfg, outputs[k][b:b + 1], flows_forward[b:b + 1, i - 1])
                kws.append(kf)
                nbs.append(kf if (align_key and k == i - 1) else outputs[i - 1][b:b + 1])
            key_warp, neighbor = torch.cat(kws), torch.cat(nbs)
        feat = (torch.cat([lr, key_warp, neighbor, outputs[i]], 1) if with_cat
                else torch.cat([lr, key_warp, outputs[i]], 1))
        gamma = gammas[:, i] if with_bias else None
        fp = resblocks(sd, cfg, 'forward_resblocks', feat, par_map[:, i], ew_all[:, i], gamma)
        outputs[i] = fp
        if cfg.get('vsr', False):                                        # :135-142
            o = F.leaky_relu(pixel_shuffle_pack(sd, 'upsample1.', fp), 0.1)
            o = F.leaky_relu(pixel_shuffle_pack(sd, 'upsample2.', o), 0.1)
            o = F.leaky_relu(F.conv2d(o, sd['conv_hr.weight'], sd['conv_hr.bias'], padding=1), 0.1)
            o = F.conv2d(o, sd['conv_last.weight'], sd['conv_last.bias'], padding=1)
            o = o + F.interpolate(lr, scale_factor=4, mode='bilinear', align_corners=False)
        else:                                                            # :144-146
            o = F.leaky_relu(F.conv2d(fp, sd['conv_hr.weight'], sd['conv_hr.bias'], padding=1), 0.1)
            o = F.conv2d(o, sd['conv_last.weight'], sd['conv_last.bias'], padding=1) + lr
        outs.append(o)
    return torch.stack(outs, dim=1)                                      # :149


# --------------------------------------------------------------------------
# metric (what "PSNR delta vs ref" is measured with)
# --------------------------------------------------------------------------
def tensor2img_uint8(frame):
    """core/misc.py:51-71 for one (3,H,W) RGB frame in [0,1]: clamp, RGB->BGR,
    HWC, *255, round -> uint8."""
    import numpy as np
    f = frame.detach().float().cpu().clamp(0, 1).numpy()
    f = np.transpose(f[[2, 1, 0]], (1, 2, 0))
    return (f * 255.0).round().astype(np.uint8)


def psnr_uint8(img1, img2, crop_border=0):
    """core/evaluation/metrics.py:170-215 (HWC order, no Y conversion)."""
    import numpy as np
    a = img1.astype(np.float32)
    b = img2.astype(np.float32)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return float(20.0 * np.log10(255.0 / np.sqrt(mse)))


def clip_psnr(output, gt, crop_border=0):
    """restorers/basicvsr.py:119-153 (evaluate): per frame PSNR, mean over frames.
    output, gt (1,T,3,H,W) or (T,3,H,W)."""
    if output.dim() == 5:
        output, gt = output[0], gt[0]
    vals = [psnr_uint8(tensor2img_uint8(output[i]), tensor2img_uint8(gt[i]), crop_border)
            for i in range(output.shape[0])]
    return sum(vals) / len(vals)


# --------------------------------------------------------------------------
# bitstream side-info rasteriser (the step right BEFORE the hot path)
# --------------------------------------------------------------------------
def rasterise_side_info(records, rec_frame, slices, h, w):
    """datasets/pipelines/loading_ipb.py:328-369 (LoadImageFromFileList_ipb.__call__ inner loop) followed
    by RescaleToZeroOne on `partitions` (normalization.py:93-99) and FramesToTensor's HWC->CHW
    (formating.py:124-137).

    records (R,10) float32 rows (direction, w, h, x_w, y_w, x, y, motion_x, motion_y, scale) in file order,
    rec_frame (R,) frame index of each row, slices: sequence of 'I'/'P'/'B' per frame.
    Returns mvs (T,4,h,w) float32 [pixels] and partitions (T,3,h,w) float32 in {0, 1/255}.
    Python slice semantics are kept as they are (a block whose start index is negative wraps around and
    usually selects nothing; the far side is clipped)."""
    import numpy as np
    T = len(slices)
    mvs, parts = [], []
    part_ch = {256: 0, 128: 1, 64: 2}
    p_offset = None                                  # unassigned in the reference until the first frame ends
    records = np.asarray(records, np.float32)
    for f in range(T):
        is_b = (slices[f] == 'B')
        mv = np.zeros((h, w, 4), np.float32)
        part = np.zeros((h, w, 3), np.float32)
        for row in records[np.asarray(rec_frame) == f]:
            direction, bw, bh, x_w, y_w, x, y, mx, my, scale = row
            x, y, bw, bh, x_w, y_w = int(x), int(y), int(bw), int(bh), int(x_w), int(y_w)
            mx = mx / scale
            my = my / scale
            ys, xs = slice(y - bh // 2, y + bh // 2), slice(x - bw // 2, x + bw // 2)
            if direction < 0:
                mv[ys, xs, 0] = mx
                mv[ys, xs, 1] = my
            elif direction > 0 and is_b:
                mv[ys, xs, 2] = mx
                mv[ys, xs, 3] = my
            elif direction > 0 and not is_b:
                tgt = mvs[-p_offset]
                yw, xw = slice(y_w - bh // 2, y_w + bh // 2), slice(x_w - bw // 2, x_w + bw // 2)
                tgt[yw, xw, 2] = -mx
                tgt[yw, xw, 3] = -my
            part[ys, xs, part_ch[bw * bh]] = 1
        parts.append(part)
        mvs.append(mv)
        p_offset = (p_offset + 1) if is_b else 1
    mvs = np.stack([m.transpose(2, 0, 1) for m in mvs]).astype(np.float32)
    parts = np.stack([(p.astype(np.float32) / 255.).transpose(2, 0, 1) for p in parts]).astype(np.float32)
    return mvs, parts


# test plumbing (no reference counterpart): numpy state dict -> torch tensors
def to_torch_state(sd_np):
    return {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}

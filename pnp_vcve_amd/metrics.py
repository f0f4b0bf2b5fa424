"""Evaluation metrics with the reference's definitions (host side, numpy).

tensor2img: mmedit/core/misc.py:9-74 ; psnr: mmedit/core/evaluation/metrics.py:170-215 ;
ssim: metrics.py:218-355 (11x11 Gaussian sigma 1.5, 'valid' window).  The reference computes
SSIM with cv2.filter2D; cv2 is not available here, the same window is applied with a separable
'valid' correlation in float64 (SSIM parity is unpinned against cv2 itself -- SURVEY.md section 8c --
but checked against an independent scipy.ndimage restatement of the reference formula,
tests/test_host_logic.py::test_ssim_against_independent_scipy_restatement).

Reference quirk kept: with crop_border != 0 the reference slices `img[cb:-cb, cb:-cb, None]`
(metrics.py:343-345), which turns an HWC image into (H', W', 1, 3); its channel loop then runs once
over `img[..., 0]`, i.e. SSIM is computed on channel 0 (B of the BGR image) only.  PSNR is a mean over
all elements and is unaffected.  The shipped configs use crop_border=0.
"""
import numpy as np
import torch


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1)):
    """(1,3,H,W) / (3,H,W) RGB tensor in [0,1] -> (H,W,3) BGR uint8 (rounded)."""
    if not torch.is_tensor(tensor):
        raise TypeError(f'tensor expected, got {type(tensor)}')
    t = tensor.squeeze(0).squeeze(0).float().detach().cpu().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 3:
        img = np.transpose(t.numpy()[[2, 1, 0], :, :], (1, 2, 0))
    elif t.dim() == 2:
        img = t.numpy()
    else:
        raise ValueError(f'Only support 3D or 2D tensor here. But received with dimension: {t.dim()}')
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def psnr(img1, img2, crop_border=0, input_order='HWC', convert_to=None):
    assert img1.shape == img2.shape, f'Image shapes are different: {img1.shape}, {img2.shape}.'
    if convert_to is not None:
        raise NotImplementedError('convert_to is not used by the shipped configs')
    if input_order == 'CHW':
        img1, img2 = img1.transpose(1, 2, 0), img2.transpose(1, 2, 0)
    a, b = img1.astype(np.float32), img2.astype(np.float32)
    if crop_border != 0:
        a = a[crop_border:-crop_border, crop_border:-crop_border, None]
        b = b[crop_border:-crop_border, crop_border:-crop_border, None]
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 20. * np.log10(255. / np.sqrt(mse))


def _gauss_valid(x, k):
    # separable 'valid' correlation with the symmetric 1-D kernel k along both axes
    n = len(k)
    h, w = x.shape
    tmp = np.zeros((h - n + 1, w), np.float64)
    for i in range(n):
        tmp += k[i] * x[i:i + h - n + 1, :]
    out = np.zeros((h - n + 1, w - n + 1), np.float64)
    for i in range(n):
        out += k[i] * tmp[:, i:i + w - n + 1]
    return out


def _ssim_channel(a, b):
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = a.astype(np.float64), b.astype(np.float64)
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    mu1, mu2 = _gauss_valid(a, g), _gauss_valid(b, g)
    s1 = _gauss_valid(a * a, g) - mu1 ** 2
    s2 = _gauss_valid(b * b, g) - mu2 ** 2
    s12 = _gauss_valid(a * b, g) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    return m.mean()


def ssim(img1, img2, crop_border=0, input_order='HWC', convert_to=None):
    assert img1.shape == img2.shape, f'Image shapes are different: {img1.shape}, {img2.shape}.'
    if convert_to is not None:
        raise NotImplementedError('convert_to is not used by the shipped configs')
    if input_order == 'CHW':
        img1, img2 = img1.transpose(1, 2, 0), img2.transpose(1, 2, 0)
    if img1.ndim == 2:
        img1, img2 = img1[..., None], img2[..., None]
    if crop_border != 0:       # metrics.py:343-350: the `None` index leaves only channel 0 in the loop (see module docstring)
        img1 = img1[crop_border:-crop_border, crop_border:-crop_border, :1]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, :1]
    return float(np.mean([_ssim_channel(img1[..., i], img2[..., i]) for i in range(img1.shape[2])]))


ALLOWED_METRICS = {'PSNR': psnr, 'SSIM': ssim}

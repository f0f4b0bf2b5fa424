"""Test loops (mmedit/apis/test.py:13-126) for the hot path: one process per GPU, each runs its
shard of clips through model(test_mode=True, **data); metrics are gathered with one all-gather."""
import torch

from .datasets import collate
from .dist import gather_clip_metrics, get_dist_info, shard_indices


def _to_device(data, device):
    """Host -> device; when the clip carries raw decoder MV records (CompressedClipFolderDataset) the dense
    motion / partition maps are painted on the GPU (pnp_rasterise_side_info_f32) instead of on the host."""
    out = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in data.items()}
    if 'mv_records' in out:
        from .ops import rasterise_side_info
        rec, rf = out.pop('mv_records'), out.pop('rec_frame')
        t, h, w = out['lq'].shape[1], out['lq'].shape[-2], out['lq'].shape[-1]
        sl = [float(v) for v in data['slices'].reshape(-1)[:t]]
        mvs, par = rasterise_side_info(rec.float(), rf.int(), sl, h, w)
        out['mvs'], out['partitions'] = mvs.unsqueeze(0), par.unsqueeze(0)
    return out


def single_gpu_test(model, dataset, save_image=False, save_path=None, device='cuda'):
    model.eval()
    results = []
    for i in range(len(dataset)):
        data = _to_device(collate([dataset[i]]), device)
        with torch.no_grad():
            results.append(model(test_mode=True, save_image=save_image, save_path=save_path, **data))
    return results


def multi_gpu_test(model, dataset, save_image=False, save_path=None, device='cuda', metrics=('PSNR', 'SSIM')):
    """Returns, on every rank, the ordered per-clip results [{'eval_result': {...}}, ...]."""
    model.eval()
    rank, world = get_dist_info()
    mine = shard_indices(len(dataset), rank, world)
    local = []
    for i in mine:
        data = _to_device(collate([dataset[i]]), device)
        with torch.no_grad():
            res = model(test_mode=True, save_image=save_image, save_path=save_path, **data)
        fps = data['lq'].shape[1] / model.last_forward_seconds if getattr(model, 'last_forward_seconds', None) else 0.0
        local.append([float(res['eval_result'].get(m, float('nan'))) for m in metrics] + [fps])
    table = gather_clip_metrics(local, len(dataset), device=device if world > 1 else None)
    return [dict(eval_result={m: float(table[i, j]) for j, m in enumerate(metrics)}, frames_per_s=float(table[i, -1]))
            for i in range(len(dataset))]

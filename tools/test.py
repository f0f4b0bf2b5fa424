#!/usr/bin/env python
"""Evaluation driver with the reference's CLI (tools/test.py:18-61 of ZeldaM1/PnP-VCVE):
    python tools/test.py CONFIG CHECKPOINT [--launcher pytorch] [--save-path DIR] [--cfg-options k=v ...]
CHECKPOINT may be 'none' (seeded random weights: released checkpoints are not available offline)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pnp_vcve_amd  # noqa: E402,F401
from pnp_vcve_amd import restorer  # noqa: E402,F401
from pnp_vcve_amd.apis import multi_gpu_test  # noqa: E402
from pnp_vcve_amd.checkpoint import load_checkpoint  # noqa: E402
from pnp_vcve_amd.config import Config, parse_cfg_options  # noqa: E402
from pnp_vcve_amd.datasets import build_dataset  # noqa: E402
from pnp_vcve_amd.dist import get_dist_info, init_dist  # noqa: E402
from pnp_vcve_amd.registry import build_model  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='PnP-VCVE hot-path tester (MI355X)')
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--out')
    p.add_argument('--gpu-collect', action='store_true', help='accepted for CLI parity; collection is always on-GPU')
    p.add_argument('--testdir_lr', default=None)
    p.add_argument('--testdir_gt', default=None)
    p.add_argument('--save-path', default=None, type=str)
    p.add_argument('--tmpdir')
    p.add_argument('--cfg-options', nargs='+', default=None)
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    p.add_argument('--fp16', action='store_true',
                   help='fp16 MFMA operands for the 64-channel convs (same as `fp16 = dict()` in the config); default fp32')
    p.add_argument('--precision', choices=['fp32', 'fp16', 'f16x3'], default=None,
                   help="arithmetic of the 64-channel convs: fp32 (exact, default), fp16 (= --fp16), f16x3 (split fp16: fp32-level "
                        "results from three fp16 MFMAs per product, ~2.2x the fp32 rate); also `precision = '...'` in the config")
    p.add_argument('--clips-in-flight', type=int, default=2, choices=[1, 2],
                   help='2 (default): two clips of equal shape go through the generator as one batch, interleaved on two streams '
                        '(+5 % at 720p, bit-identical, a second workspace); 1: strictly one clip per forward like the reference')
    p.add_argument('--local_rank', type=int, default=0)
    a = p.parse_args()
    if 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = str(a.local_rank)
    return a


def main():
    args = parse_args()
    cfg = Config.fromfile(args.config)
    if args.cfg_options:
        cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    # One arithmetic per run.  `--fp16` / `fp16 = dict(...)` in the config (the mmcv idiom) IS precision 'fp16'; a command line
    # switch overrides the config; two switches that disagree are an error instead of the last one silently winning.
    cli = {p for p in (args.precision, 'fp16' if args.fp16 else None) if p is not None}
    if len(cli) > 1:
        raise SystemExit(f'tools/test.py: --fp16 contradicts --precision {args.precision}')
    conf = {p for p in (cfg.get('precision', None), 'fp16' if cfg.get('fp16', None) is not None else None) if p is not None}
    if not cli and len(conf) > 1:
        raise SystemExit(f"tools/test.py: the config sets fp16 = dict(...) and precision = {cfg.get('precision')!r}")
    precision = next(iter(cli or conf), None)
    if args.launcher != 'none':
        init_dist(args.launcher, **cfg.dist_params)
    rank, world = get_dist_info()
    if args.seed is not None:
        torch.manual_seed(args.seed)
    if args.testdir_lr is not None:                     # tools/test.py:98-103 of the reference
        cfg.merge_from_dict({'data.test.lq_folder': args.testdir_lr})
        print('-------------------- test LR dir :', args.testdir_lr)
    if args.testdir_gt is not None:
        cfg.merge_from_dict({'data.test.gt_folder': args.testdir_gt})
        print('-------------------- test GT dir :', args.testdir_gt)
    dataset = build_dataset(cfg.data.test)
    model = build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    if args.checkpoint.lower() != 'none':
        load_checkpoint(model, args.checkpoint, map_location='cpu')
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    model = model.to(dev)          # no DDP wrap: inference replicas share nothing
    if precision == 'fp16':
        from pnp_vcve_amd.restorer import wrap_fp16_model
        wrap_fp16_model(model)
    elif precision is not None:
        model.precision = precision
    if args.save_path is not None:      # PNG encode off the critical path (pnp_vcve_amd/io_async.py)
        from pnp_vcve_amd.io_async import FrameWriter
        model.frame_writer = FrameWriter(max_workers=4)
    try:
        outputs = multi_gpu_test(model, dataset, save_image=args.save_path is not None, save_path=args.save_path,
                                 device=dev, metrics=tuple(cfg.test_cfg['metrics']), clips_in_flight=args.clips_in_flight)
    finally:
        if getattr(model, 'frame_writer', None) is not None:
            model.frame_writer.close()
            model.frame_writer = None
    if rank == 0:
        stats = dataset.evaluate(outputs)
        for k, v in stats.items():
            print(f'Eval-{k}: {v}')
        print('{:.4f}/{:.4f}'.format(float(stats.get('PSNR', float('nan'))), float(stats.get('SSIM', float('nan')))))
        fps = [o['frames_per_s'] for o in outputs]
        import torch.distributed as dist
        backend = dist.get_backend() if dist.is_available() and dist.is_initialized() else 'none'
        print(f'clips {len(outputs)}  world {world}  backend {backend}  mean frames/s per clip {sum(fps) / len(fps):.2f}')
        if args.out:
            torch.save(outputs, args.out)


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Headline benchmark: enhanced frames/s of the BAE/CAA forward hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 720p|lr180|128]

One "step" = one forward of the generator over one synthetic 7-frame clip per GPU (inputs resident
in HBM).  N > 1: one process per GPU (torch.distributed.run), clips sharded one per rank (weak
scaling, replicas only -- the model has no cross-GPU tensors), a barrier + synchronize on both sides
of the timed region, MAX over ranks, and one RCCL all-gather of (PSNR, frames/s) per rank.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

WORKLOADS = {'720p': (720, 1280), 'lr180': (180, 320), '128': (128, 128)}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s achievable)


def committed_pmc_traffic(precision='fp32'):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc.json, written by
    tools/profile_gpu.sh for this same command at 720p): (2 x FETCH_SIZE + WRITE_SIZE) KiB, the gfx950
    correction of MI355X_MICROARCH.md.  None when no profile has been committed."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json'))
                   if ('fp16' in os.path.basename(f)) == (precision == 'fp16'))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def make_inputs(seed, t, h, w, dev, n=1):
    from pnp_vcve_amd import synthetic as syn
    clip = syn.make_clip(seed=seed, n=n, t=t, h=h, w=w, slices='IBBBP', qp_mode='qp', crf=25 if n == 1 else [25] * n,
                         block=8 if h % 8 == 0 else 4)
    return clip, {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}


def gpu_psnr(out, gt):
    """reference PSNR definition (core/misc.py:51-71 + core/evaluation/metrics.py:200-215), mean over frames,
    from the on-device statistic kernel (pnp_psnr_sse_f32)."""
    from pnp_vcve_amd.ops import psnr_frames
    return float(psnr_frames(out, gt).mean())


def cpu_baseline(sd_np, cfg, h, w):
    """The oracle (CPU restatement of the reference, oracle/cpu_ref.py) timed on the host cores on a
    bounded sample of the same workload: a 2-frame clip at the full frame size (per-frame cost does
    not depend on T; ~40 s at 720p on the EPYC hosts).  The thread count is calibrated first (8/16/32 on a 128x128 clip): on the
    2x64-core EPYC hosts of the MI355X boxes oneDNN is fastest at 16 threads on these 64-channel
    convs and 10x slower at 128+."""
    from oracle import cpu_ref
    from pnp_vcve_amd import synthetic as syn
    sd = cpu_ref.to_torch_state(sd_np)

    def run(clip):
        a = {k: torch.from_numpy(v) for k, v in clip.items()}
        with torch.no_grad():
            t0 = time.perf_counter()
            o = cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'],
                                          a['partitions'])
            return o, time.perf_counter() - t0

    cal = syn.make_clip(seed=1, n=1, t=2, h=128, w=128)
    best, best_nt = None, None
    for nt in (8, 16, 32):
        if nt > (os.cpu_count() or 1):
            break
        torch.set_num_threads(nt)
        run(cal)
        _, dt = run(cal)
        if best is None or dt < best:
            best, best_nt = dt, nt
    torch.set_num_threads(best_nt or 1)
    clip = syn.make_clip(seed=4242, n=1, t=2, h=h, w=w, slices='IBBBP', qp_mode='qp', crf=25,
                         block=8 if h % 8 == 0 else 4)
    ref, dt = run(clip)
    return clip, ref, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='720p', choices=sorted(WORKLOADS))
    ap.add_argument('--frames', type=int, default=7)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'fp16'],
                    help="fp16 = BASELINE configs[4]'s opt-in 'fp16 MFMA convs' (fp16 operands, fp32 accumulate and "
                         "feature maps); the headline metric is the default fp32")
    ap.add_argument('--vsr', action='store_true', help='x4 SR heads (generator vsr=True): output is 4h x 4w')
    ap.add_argument('--clips', type=int, default=1,
                    help='clips per GPU per step (one batch; small frames run them concurrently, DESIGN.md section 4)')
    ap.add_argument('--graphs', action='store_true',
                    help='replay each clip as one hipGraph (generator.use_graphs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip per-kernel HIP-event timing')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    backend = os.environ.get('PNP_DIST_BACKEND', 'nccl')     # 'nccl' is RCCL on ROCm; 'gloo' only for 1-GPU dry runs
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cdev = dev if backend == 'nccl' else torch.device('cpu')   # where the tiny collective payloads live

    from pnp_vcve_amd import synthetic as syn
    from pnp_vcve_amd.registry import build_backbone
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    cfg['vsr'] = bool(args.vsr)
    sd_np = syn.make_state_dict(cfg, seed=2025)
    m = build_backbone(dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.to(dev).eval()
    m.fp16_enabled = args.precision == 'fp16'
    m.use_graphs = bool(args.graphs)

    h, w = WORKLOADS[args.workload]
    T = args.frames
    clip, a = make_inputs(1000 + rank, T, h, w, dev, args.clips)   # clip `rank` of the synthetic set (sampler rule: idx[rank::world])

    def step():
        with torch.no_grad():
            return m(a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    # Two HIP events per launch cost ~1 % of a 720p step but 50-90 % of a 128x128 one (700 launches of a few us):
    # below 720p the per-kernel timing runs as a separate pass of the same steps right after the timed region.
    events_inside = not args.no_kernel_events and args.workload == '720p'
    if events_inside:
        m.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    et = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
    elapsed_max = float(et.item())
    if not args.no_kernel_events and not events_inside:
        m.profile(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    prof = None if args.no_kernel_events else m.profile_read()
    m.profile(False)

    # per-rank metrics, gathered with one small collective (PSNR, frames/s): mmedit/apis/test.py:211-233
    gt = a['gt']
    if args.vsr:        # synthetic HR ground truth: the LR one, nearest-upsampled (only feeds the gathered metric)
        gt = gt.repeat_interleave(4, -1).repeat_interleave(4, -2).contiguous()
    psnr = gpu_psnr(out, gt)
    mine = torch.tensor([psnr, args.steps * T * args.clips / elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        allm = torch.stack(allm).cpu().numpy()
    else:
        allm = mine.cpu().numpy()[None]

    if rank == 0:
        frames = world * args.steps * T * args.clips
        res = {
            'metric': 'enhanced frames/sec (1280x720, 7-frame window)' if args.workload == '720p'
                      else f'enhanced frames/sec ({w}x{h}, {T}-frame window)',
            'value': frames / elapsed_max, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed_max / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'fp32' else 'f16 MFMA operands, f32 accumulate / feature maps (opt-in)',
            'data': 'synthetic',
            'config': {'workload': f'{args.clips} x {T}x3x{h}x{w} clip per GPU per step '
                                   f'({dict(**{"720p": "BASELINE configs[2] shape", "128": "BASELINE configs[0-1] shape", "lr180": "BASELINE configs[4] LR shape"})[args.workload]}), '
                                   f'full BAE+CAA forward, config HR_davis_LR_128x128 generator, seeded random weights',
                       'vsr_x4_heads': bool(args.vsr), 'hip_graphs': bool(args.graphs),
                       'parallelism': f'clip-sharded replicas x{world}', 'frames_per_step_per_gpu': T * args.clips},
            'kernel_events': ('none' if args.no_kernel_events else 'inside the timed region' if events_inside
                              else 'separate pass of the same steps after the timed region'),
            'psnr_per_rank': [float(x) for x in allm[:, 0]],
            'frames_per_s_per_rank': [float(x) for x in allm[:, 1]],
        }
        pmc, pmc_src = committed_pmc_traffic(args.precision) if args.workload == '720p' and not args.vsr else ({}, None)
        if prof is not None:
            cb = prof['conv_block']
            ci = prof['conv_input']
            ch = prof['conv_head']
            wp = prof['mv_warp']
            conv_ms = cb['ms'] + ci['ms'] + ch['ms']
            conv_fl = cb['work'] + ci['work'] + ch['work']
            ach = cb['work'] / (cb['ms'] * 1e-3) / 1e12 if cb['ms'] > 0 else 0.0
            # The persistent kernel skips a 1x1 partition branch on tiles where its plane is all zero (bit-identical).
            # `achieved` stays the reference's dense (algorithmic) FLOP count over the measured time; `executed` discounts
            # the skipped branch chunks so that the matrix pipe's real utilisation is visible next to it.
            from pnp_vcve_amd.ops import par_tile_flags
            fl = torch.stack([par_tile_flags(a['partitions'][0, i]) for i in range(T)])
            branches = sum(((fl >> j) & 1).float().mean().item() for j in range(3))        # needed branches per tile
            run = torch.clamp(sum(((fl >> j) & 1) for j in range(3)), min=1).float().mean().item()   # chunks really run
            nb = 2 * cfg['num_blocks']
            dense = nb * (2 * 576 + 192) + 576
            skipped_frac = nb * 64 * (3 - run) / dense if h * w >= 1024 * 128 else 0.0
            executed = ach * (1 - skipped_frac)
            res['roofline'] = {'kernel': 'conv3x3_persist_kernel<PAR> (the 64->64 BAE-block convs + conv_hr; fp32 MFMA 32x32x2; '
                                         'conv3x3_mfma_kernel below 1024 tiles)',
                               'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / PEAK_F32_MFMA_TFLOPS,
                               'executed_TFLOPs': executed, 'executed_frac': executed / PEAK_F32_MFMA_TFLOPS,
                               'partition_branches_needed_per_tile': branches, 'partition_branch_chunks_run_per_tile': run,
                               'traffic': (pmc.get('conv3x3_persist_kernel', pmc.get('conv3x3_mfma_kernel<4,1,2,2>', {}))).get('hbm_bytes_per_launch'),
                               'traffic_source': pmc_src,
                               'launches': cb['launches'], 'avg_launch_us': 1e3 * cb['ms'] / max(cb['launches'], 1),
                               'all_convs_TFLOPs': conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
                               'device_ms_per_step': {'conv_block': cb['ms'] / args.steps, 'conv_input': ci['ms'] / args.steps,
                                                      'conv_head': ch['ms'] / args.steps, 'mv_warp': wp['ms'] / args.steps}}
            if args.precision == 'fp16':
                # At the fp16 matrix rate the block convs are HBM-bound: price them in bytes.  Per frame the kind holds
                # 16 front halves (read x, write o, 3 partition planes), 16 back halves (read o, read x, write) and conv_hr.
                nb = 2 * cfg['num_blocks']
                bytes_frame = h * w * (nb * (512 + 12) + nb * 768 + (16 if args.vsr else 1) * 512)
                gbs = bytes_frame * T * args.steps / (cb['ms'] * 1e-3) / 1e9 if cb['ms'] > 0 else 0.0
                f16k = [v for k, v in pmc.items() if 'conv3x3_f16_kernel' in k]      # launch-weighted mean over the variants
                f16_traffic = (sum(v['hbm_bytes_per_launch'] * v['launches'] for v in f16k) / sum(v['launches'] for v in f16k)
                               if f16k else None)
                res['roofline'] = {'kernel': 'conv3x3_f16_kernel<PAR,LR4> (64->64 BAE-block convs + conv_hr; fp16 MFMA 32x32x16, '
                                             'weights resident in LDS, persistent strips)',
                                   'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                   'frac': gbs / PEAK_HBM_GBS, 'traffic': f16_traffic, 'traffic_source': pmc_src,
                                   'launches': cb['launches'],
                                   'avg_launch_us': 1e3 * cb['ms'] / max(cb['launches'], 1),
                                   'matrix_TFLOPs': ach, 'matrix_peak_TFLOPs': 2500.0,
                                   'device_ms_per_step': res['roofline']['device_ms_per_step']}
            if wp['launches']:
                gbs = wp['work'] / (wp['ms'] * 1e-3) / 1e9
                res['roofline_mv_warp'] = {'kernel': 'mv_warp_nhwc_kernel (MV-guided bilinear alignment)', 'bound': 'hbm',
                                           'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                           'frac': gbs / PEAK_HBM_GBS,
                                           'traffic': pmc.get('mv_warp_nhwc_kernel', {}).get('hbm_bytes_per_launch'),
                                           'traffic_source': pmc_src, 'launches': wp['launches'],
                                           'avg_launch_us': 1e3 * wp['ms'] / wp['launches'],
                                           'algorithmic_bytes_per_launch': wp['work'] / wp['launches']}
        if world == 1 and not args.no_cpu_baseline:
            cclip, ref, dt = cpu_baseline(sd_np, cfg, h, w)
            ca = {k: torch.from_numpy(v).to(dev) for k, v in cclip.items()}
            with torch.no_grad():
                got = m(ca['lq'], ca['QPs'], ca['slices'], ca['mvs'], ca['base_QPs'], ca['partitions']).cpu()
            from oracle import cpu_ref
            gt = torch.from_numpy(cclip['gt'])
            res['cpu_baseline'] = {
                'value': 2 / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                'sample': f'oracle/cpu_ref.py (PyTorch-CPU fp32 restatement of the reference, pinned by tests/golden) on '
                          f'one 2x3x{h}x{w} clip (2 of the 7 frames, same frame size) = {dt:.1f} s; threads calibrated '
                          f'over 8/16/32 on this host ({os.cpu_count()} logical CPUs)',
                'sample_seconds': dt}
            res['parity'] = {'sample': f'2x3x{h}x{w}', 'max_abs_diff_vs_cpu': float((got - ref).abs().max()),
                             'psnr_delta_db': cpu_ref.clip_psnr(got, gt) - cpu_ref.clip_psnr(ref, gt),
                             'gate': 1e-3}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Headline benchmark: enhanced frames/s of the BAE/CAA forward hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 720p|lr180|128] [--precision fp32|fp16] ...

One "step" = one forward of the generator over one synthetic 7-frame clip per GPU (inputs resident in HBM).

N > 1: one process per GPU.  Started by the driver through torch.distributed.run (RANK / WORLD_SIZE in the env) the
script is a rank; started plainly as `python bench.py --gpus N` it LAUNCHES the N ranks itself -- fresh child
processes through `python -m torch.distributed.run`, created before this process has imported torch or touched the
GPU -- relays rank 0's single JSON line and exits with the children's status (the reference's launcher:
tools/dist_test.sh:11-22).  Clips are sharded one per rank (weak scaling, replicas only: the model has no cross-GPU
tensors), a barrier + synchronize on both sides of the timed region, MAX over ranks, and one RCCL all-gather of
(PSNR, frames/s) per rank (mmedit/apis/test.py:211-233).  Rank 0 prints ONE JSON line.

stdout carries exactly ONE strict-JSON line of at most 4 KB and nothing else (the driver keeps the last 8 KB of stdout; round 3's
21 KB line was cut in half and never parsed): file descriptor 1 is pointed at stderr for the run -- RCCL and gloo print to
stdout too -- and the line is written to the saved descriptor at the end.  At N = 1 with the default workload the other BASELINE.json workloads (7x3x128x128 fp32, 1 and
8 clips; 180x320 fp16 with and without the x4 heads; the opt-in precisions at 720p; the end-to-end loop) are measured in the same
process after the headline, each with its own timed region, roofline and kernel-event mode; they go to `bench_secondary.json`
beside this file (and to stderr), and the line carries only `north_star_128`, a compact summary of the 7x3x128x128 entries.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {'720p': (720, 1280), 'lr180': (180, 320), '128': (128, 128)}
WORKLOAD_NOTE = {'720p': 'BASELINE configs[2] shape', '128': 'BASELINE configs[0-1] shape',
                 'lr180': 'BASELINE configs[4] LR shape'}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16/bf16
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s achievable)
GEN_TYPE = 'IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par'


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='720p', choices=sorted(WORKLOADS))
    ap.add_argument('--frames', type=int, default=7)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'fp16', 'f16x3'],
                    help="fp16 = BASELINE configs[4]'s opt-in 'fp16 MFMA convs' (fp16 operands, fp32 accumulate and "
                         "feature maps); the headline metric is the default fp32")
    ap.add_argument('--vsr', action='store_true', help='x4 SR heads (generator vsr=True): output is 4h x 4w')
    ap.add_argument('--deform', default='vos', choices=['vos', 'basic', 'fvc'],
                    help="alignment: 'vos' = MV bilinear warp (the shipped configs); 'basic'/'fvc' = the modulated "
                         "deformable aligners (iconvsr_mv.py:21-84), reported as roofline_dcn")
    ap.add_argument('--clips', type=int, default=1,
                    help='clips per GPU per step (one batch; small frames run them concurrently, DESIGN.md section 4)')
    ap.add_argument('--graphs', action='store_true', help='replay each clip as one hipGraph (generator.use_graphs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip per-kernel HIP-event timing')
    ap.add_argument('--no-secondary', action='store_true', help='skip the other BASELINE workloads after the headline')
    ap.add_argument('--f16-mirrors', type=int, default=None, choices=[0, 1, 2],
                    help='fp16 path A/B: 0 no fp16 mirrors (r02 schedule), 1 frame slots + aligned key frame (default), 2 also the '
                         'running map inside a branch (PNP_OPT_F16_CHAIN_MIRRORS)')
    ap.add_argument('--winograd', type=int, default=None, choices=[0, 1, 2],
                    help='PNP_OPT_WINOGRAD of the fp32 path (include/pnpvcve.h): 0 direct kernels, 1 Winograd F(2x2,3x3) -- quadrant-unit '
                         'kernels on frames of up to 128 16x16 tiles, persistent tile kernels above, 2 the tile kernels at every frame '
                         'size (default: the library default)')
    ap.add_argument('--tile-queue', type=int, default=None, choices=[0, 1],
                    help='split-fp16 path A/B: 0 = every block walks a static share of the tiles, 1 (default) = per-XCD tile queue '
                         '(PNP_OPT_TILE_QUEUE)')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torch.distributed environment starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """Start N ranks of this script as fresh child processes (torch.distributed.run, rendezvous on 127.0.0.1) and
    relay rank 0's JSON line.  Runs BEFORE this process imports torch: the parent never initialises the GPU, and no
    process that did is ever replaced by another program."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                       # rank 0's JSON line goes to stdout; everything else to stderr
        s = ln.strip()
        if s.startswith('{') and '"metric"' in s:
            line = s
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print('bench.py: the ranks exited without a result line', file=sys.stderr)
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------
def committed_pmc_traffic(tag):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_<tag>pmc.json, written by
    tools/profile_gpu.sh for this same command): (2 x FETCH_SIZE + WRITE_SIZE), the gfx950 correction of
    MI355X_MICROARCH.md.  The newest round wins.  ({}, None) when no profile has been committed."""
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json'))
             if re.fullmatch(rf'r\d+_{tag}pmc\.json', os.path.basename(f))]
    if not files:
        return {}, None
    f = sorted(files)[-1]
    with open(f) as fh:
        return json.load(fh), os.path.relpath(f, ROOT)


def _wino_flags(name):
    """(PAR, RES, MS, FO) of a summarised `conv3x3_wino_kernel<...>` name -- or, for `conv3x3_wino_gated_kernel<RES>` (round 6: a front
    half behind the device-side gate as ONE launch, fold-only or branch body), (None, RES, False, None); None for anything else"""
    import re
    m = re.search(r'conv3x3_wino_kernel<\s*(true|false)\s*,\s*(true|false)\s*,\s*(true|false)\s*,\s*(true|false)\s*>', name)
    if m:
        return tuple(x == 'true' for x in m.groups())
    m = re.search(r'conv3x3_wino_gated_kernel<\s*(true|false)\s*>', name)
    return (None, m.group(1) == 'true', False, None) if m else None


def _launch_weighted_traffic(pmc, prefix, min_bytes=0.0):
    """launch-weighted mean HBM bytes per launch over the kernel variants whose summarised name starts with `prefix`.

    `conv3x3_wino_kernel<PAR,RES,MS,FO>` is several kernels under one symbol: the roofline of the block convs counts exactly the
    single-source instantiations (MS = false: plain = conv_hr, RES = back halves, FO / PAR = front halves, + the gated front-half
    kernels) -- the multi-source one is the input conv -- and among those only launches that worked: round 5 launched a front half
    twice behind a device-side gate and one of the two returned after reading the frame's partition word (a few KB; `min_bytes` drops
    those variants)."""
    ks = []
    for k, v in pmc.items():
        if 'hbm_bytes_per_launch' not in v:
            continue
        if prefix == 'conv3x3_wino_kernel':
            f = _wino_flags(k)
            if f is None or f[2]:
                continue
        elif not (k.startswith(prefix) or '::' + prefix in k):
            continue
        if v['hbm_bytes_per_launch'] < min_bytes:
            continue
        ks.append(v)
    if not ks:
        return None
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks) / sum(v['launches'] for v in ks)


def block_conv_algorithmic_bytes(h, w, num_blocks):
    """algorithmic HBM bytes per launch of the fp32 block convs (launch-weighted over a frame's 2 x num_blocks halves of both
    sweeps + conv_hr), fp32 pixel-major maps of 256 B per pixel: front half = read x + 3 partition planes + write o = 524 B / px,
    back half = read o + residual x + write = 768 B / px, conv_hr = read + write = 512 B / px"""
    nb = 2 * num_blocks
    return (nb * 524 + nb * 768 + 512) / (2 * nb + 1) * h * w


MAX_LINE_BYTES = 4096
SECONDARY_FILE = os.path.join(ROOT, 'bench_secondary.json')


def strict(obj, digits=6):
    """JSON-safe copy: non-finite floats -> None (json.dumps(allow_nan=False) would raise), floats to `digits` significant digits."""
    import math
    if isinstance(obj, float):
        if not math.isfinite(obj):
            return None
        if obj.is_integer() and abs(obj) < 2.0 ** 53:
            return int(obj)                       # byte / launch counts stay exact
        return float(f'{obj:.{digits}g}') if digits else obj
    if isinstance(obj, dict):
        return {k: strict(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [strict(v, digits) for v in obj]
    return obj


# what the single stdout line keeps of a roofline object (the full objects, with their prose, go to bench_secondary.json)
ROOFLINE_KEEP = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launches', 'avg_launch_us', 'frac_wall',
                 'algorithmic_frac', 'algorithmic_TFLOPs', 'winograd', 'frac_dense_par', 'hbm_frac', 'algorithmic_bytes_per_launch',
                 'traffic_ratio', 'traffic_source')


def bounded_line(res):
    """the ONE stdout line: strict JSON, <= MAX_LINE_BYTES.  Rooflines are cut to ROOFLINE_KEEP with the kernel named by its
    symbol only; if the line is still too long, optional blocks are dropped in a fixed order (never value / roofline / cpu_baseline)."""
    line_obj = dict(res)
    for k in ('roofline', 'roofline_mv_warp', 'roofline_dcn'):
        if k in line_obj:
            r = line_obj[k]
            c = {q: r[q] for q in ROOFLINE_KEEP if q in r}
            c['kernel'] = r['kernel'].split(' (')[0]
            line_obj[k] = c
    line_obj = strict(line_obj)
    for drop in (None, 'frames_per_s_per_rank', 'psnr_per_rank', 'launches_per_frame', 'kernel_events', 'opt_in_720p', 'north_star_128', 'dist',
                 'parity', 'roofline_dcn', 'roofline_mv_warp'):
        if drop is not None:
            line_obj.pop(drop, None)
        line = json.dumps(line_obj, allow_nan=False, separators=(',', ':'))
        if len(line.encode()) <= MAX_LINE_BYTES:
            return line
    raise RuntimeError(f'bench line is {len(line)} bytes (> {MAX_LINE_BYTES}) even without its optional blocks')


def make_inputs(seed, t, h, w, dev, n=1, crfs=None):
    """lq ~ U[0,1), quarter-pel block MVs, partition class ~ U{0,1,2} per 8x8 block (one-hot
    / 255, none on I frames), IBBBP cadence, QP 20..40, base_QP = crf / 255."""
    import torch
    from pnp_vcve_amd import synthetic as syn
    crf = (crfs or [25] * n)
    clip = syn.make_clip(seed=seed, n=n, t=t, h=h, w=w, slices='IBBBP', qp_mode='qp', crf=crf[0] if n == 1 else list(crf),
                         block=8, par_classes=3)
    return clip, {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}


def gpu_psnr(out, gt):
    """reference PSNR definition (core/misc.py:51-71 + core/evaluation/metrics.py:200-215), mean over frames,
    from the on-device statistic kernel (pnp_psnr_sse_f32)."""
    from pnp_vcve_amd.ops import psnr_frames
    return float(psnr_frames(out, gt).mean())


def cpu_baseline(sd_np, cfg, h, w):
    """The oracle (CPU restatement of the reference, oracle/cpu_ref.py) timed on the host cores on a bounded sample
    of the same workload: a 2-frame clip at the full frame size (per-frame conv cost does not depend on T; both frames
    are sequence ends, so the sample runs 2 of the clip's 12 alignment calls -- the warp is 0.7 % of the CPU time;
    ~40 s at 720p on the EPYC hosts).  The thread count is calibrated first (8/16/32 on a 128x128 clip): on the
    2x64-core EPYC hosts of the MI355X boxes oneDNN is fastest at 16 threads on these 64-channel convs and 10x slower
    at 128+."""
    import torch
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
                         block=8, par_classes=3)
    ref, dt = run(clip)
    return clip, ref, dt


def cpu_baseline_128(T, runs=5, thread_counts=(8, 16, 32)):
    """BASELINE configs[0]: the oracle on one synthetic T x 3 x 128 x 128 clip on the host cores, 1 warm-up + median of `runs`,
    at the calibrated thread count -- the CPU number that stands beside the 128x128 GPU throughput entries.  Also the bounded
    CPU leg of an N > 1 line (rank 0 alone, after the timed region: ~5 s)."""
    import statistics
    import torch
    from oracle import cpu_ref
    from pnp_vcve_amd import synthetic as syn
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd = cpu_ref.to_torch_state(syn.make_state_dict(cfg, seed=2025))
    clip = syn.make_clip(seed=1000, n=1, t=T, h=128, w=128, slices='IBBBP', qp_mode='qp', crf=25, block=8, par_classes=3)
    a = {k: torch.from_numpy(v) for k, v in clip.items()}

    def run():
        with torch.no_grad():
            t0 = time.perf_counter()
            cpu_ref.generator_forward(sd, cfg, a['lq'], a['QPs'], a['slices'], a['mvs'], a['base_QPs'], a['partitions'])
            return time.perf_counter() - t0

    best, best_nt = None, 1
    for nt in thread_counts:
        if nt > (os.cpu_count() or 1) and best is not None:
            break
        nt = min(nt, os.cpu_count() or 1)
        torch.set_num_threads(nt)
        run()
        dt = run()
        if best is None or dt < best:
            best, best_nt = dt, nt
    torch.set_num_threads(best_nt)
    run()
    ts = sorted(run() for _ in range(runs))
    med = statistics.median(ts)
    return {'value': T / med, 'unit': 'frames/s', 'cores': best_nt, 'threads': best_nt, 'host_logical_cpus': os.cpu_count(),
            'kind': 'port', 'seconds_per_clip_median': med, 'seconds_per_clip_runs': ts,
            'sample': f'oracle/cpu_ref.py on one {T}x3x128x128 clip (BASELINE configs[0]), 1 warm-up + median of {runs}; threads '
                      f'calibrated over {"/".join(str(n) for n in thread_counts)}'}


F16_MIRRORS = None      # --f16-mirrors
TILE_QUEUE = None       # --tile-queue
WINOGRAD = None         # --winograd
DTYPE_TEXT = {'fp32': 'f32',
              'fp16': 'f16 MFMA operands, f32 accumulate / feature maps (opt-in; whole-clip max-abs vs fp32 up to 2e-2, PSNR delta '
                      '< 1e-3 dB)',
              'f16x3': 'split f16 (each operand hi + lo/2048, three f16 MFMAs per product), f32 accumulate / feature maps (opt-in; '
                       'whole-clip max-abs vs the reference < 1e-4, inside the 1e-3 gate)'}


def build_model(cfg, sd_np, dev, precision, graphs=False):
    import torch
    from pnp_vcve_amd.registry import build_backbone
    m = build_backbone(dict(type=GEN_TYPE, **cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.to(dev).eval()
    m.precision = precision
    m.use_graphs = bool(graphs)
    if F16_MIRRORS is not None:
        from pnp_vcve_amd import _native
        m.set_option(_native.OPT_F16_MIRRORS, 1 if F16_MIRRORS >= 1 else 0)
        m.set_option(_native.OPT_F16_CHAIN_MIRRORS, 1 if F16_MIRRORS >= 2 else 0)
    if WINOGRAD is not None:
        from pnp_vcve_amd import _native
        m.set_option(_native.OPT_WINOGRAD, WINOGRAD)
    if TILE_QUEUE is not None:
        from pnp_vcve_amd import _native
        m.set_option(_native.OPT_TILE_QUEUE, TILE_QUEUE)
    return m


def rooflines(m, prof, cfg, a, T, h, w, steps, precision, vsr, pmc, pmc_src, dense_prof=None, ms_per_step=None, clips=1):
    """roofline objects from the per-launch HIP-event records of pnp_generator_profile (device ms, launches,
    algorithmic work per kind).

    Every roofline carries `frac_wall`: the path's algorithmic work per step in the roofline's unit (all conv FLOPs, or the
    block convs' algorithmic bytes) over the step's WALL time -- comparable across entries.  With one clip per step launches run
    back to back on one stream and `achieved` / `frac` are per-launch figures (work / sum of HIP-event launch durations).  With
    several clips per step the clips run on side streams and their launches OVERLAP: the sum of launch durations exceeds the wall
    time and a per-launch figure undersells the chip, so there `achieved` / `frac` ARE the wall-clock figures and the per-launch
    ones are kept as `per_launch_achieved` / `per_launch_frac`."""
    import torch
    from pnp_vcve_amd.ops import par_tile_flags
    res = {}
    cb, ci, ch, wp, dc = (prof[k] for k in ('conv_block', 'conv_input', 'conv_head', 'mv_warp', 'dcn'))
    conv_ms = cb['ms'] + ci['ms'] + ch['ms']
    conv_fl = cb['work'] + ci['work'] + ch['work']
    ach = cb['work'] / (cb['ms'] * 1e-3) / 1e12 if cb['ms'] > 0 else 0.0
    dev_ms = {'conv_block': cb['ms'] / steps, 'conv_input': ci['ms'] / steps, 'conv_head': ch['ms'] / steps,
              'mv_warp': wp['ms'] / steps, 'dcn': dc['ms'] / steps}
    nb = 2 * cfg['num_blocks']
    big = h * w >= 1024 * 128
    if precision == 'fp32':
        # The conv kernels skip a 1x1 partition branch on tiles where its plane is all zero (bit-identical).  `ach` is the
        # reference's dense (algorithmic) FLOP count over the measured time; `executed` discounts the skipped branch chunks:
        # the matrix pipe's real utilisation; `frac_dense_par` is the same kernels timed on a dense float partition map (all
        # three branches live on every tile: nothing skipped, executed == algorithmic).
        fl = torch.stack([par_tile_flags(a['partitions'][0, i]) for i in range(T)])
        branches = sum(((fl >> j) & 1).float().mean().item() for j in range(3))        # needed branches per tile
        run = torch.clamp(sum(((fl >> j) & 1) for j in range(3)), min=1).float().mean().item()   # chunks really run
        dense = nb * (2 * 576 + 192) + 576
        skipped_frac = nb * 64 * (3 - run) / dense
        executed = ach * (1 - skipped_frac)
        kern = ('conv3x3_persist_kernel<PAR> (the 64->64 BAE-block convs + conv_hr; fp32 MFMA 32x32x2, persistent strips)'
                if big else 'conv3x3_mfma_kernel<2,2,1,2> (the 64->64 BAE-block convs + conv_hr; fp32 MFMA 32x32x2, 4x16 tiles)')
        from pnp_vcve_amd import _native
        form = _native.wino_kernel_form(h, w, m.get_option(_native.OPT_WINOGRAD))      # the header's rule, stated once
        wino, units = form is not None, form == 'units'
        if wino:
            # Winograd F(2x2,3x3) (csrc/conv_wino.hip): 16 transform positions per 2x2 output pixels instead of 36 taps -> the 3x3 part
            # executes 256/576 of the direct form's matrix FLOPs; the 1x1 branches run per 8x8-pixel quadrant (one wave), each wave
            # only the planes that are nonzero on its pixels
            par = a['partitions'][0]                                    # (T,3,h,w)
            blk = torch.nn.functional.max_pool2d((par != 0).float(), 8, ceil_mode=True)     # (T,3,h/8,w/8): plane live on the block
            branches = blk.sum(1).mean().item()                         # planes live per wave quadrant
            # a quadrant with ONE live plane that is constant on it needs no branch MFMAs at all: the plane is folded into the B
            # fragments of four positions (64 FMAs per step; conv_wino.hip, pv_finish)
            hi = torch.nn.functional.max_pool2d(par, 8, ceil_mode=True)
            lo = -torch.nn.functional.max_pool2d(-par, 8, ceil_mode=True)
            const = ((hi == lo) | (blk == 0)).all(1)                    # every plane constant (or dead) on the block
            folded = const & (blk.sum(1) == 1)
            run = (blk.sum(1) * (~folded).float()).mean().item()        # branches run as MFMAs per wave quadrant
            skipped_frac = 1 - (nb * (2 * 256 + 64 * run) + 256) / dense
            executed = ach * (1 - skipped_frac)
            kern = ('conv3x3_wino_kernel<PAR,RES> (the 64->64 BAE-block convs + conv_hr as Winograd F(2x2,3x3): fp32 MFMA 16x16x4, 16x16-pixel '
                    'block tiles on 256 persistent blocks, K-outer with in-place halo refill)')
            if units:
                kern = ('conv3x3_wino_quad_kernel<PAR,RES> (the 64->64 BAE-block convs + conv_hr as Winograd F(2x2,3x3): fp32 MFMA 16x16x4, one '
                        'block per 8x8-pixel quadrant unit, the four waves split the output channels)')
        # roofline.achieved / frac price the FLOPs the kernel EXECUTES (what the matrix pipe really did per second);
        # algorithmic_* is the reference's dense count over the same time (what a user gets per second)
        r = {'kernel': kern, 'bound': 'mfma', 'achieved': executed, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
             'frac': executed / PEAK_F32_MFMA_TFLOPS, 'algorithmic_TFLOPs': ach,
             'algorithmic_frac': ach / PEAK_F32_MFMA_TFLOPS,
             'definition': ('achieved = matrix FLOPs the kernel EXECUTES (Winograd F(2x2,3x3): 16 position products per 2x2 pixels instead of 36 '
                            'tap products, + the 1x1 branches each wave really runs; = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512) / HIP-event launch '
                            'time -- the matrix pipe\'s utilisation; algorithmic_* = the reference\'s direct-conv count 2*K*64*H*W over the same '
                            'time (what a user gets per second: it may exceed the peak, the algorithm does fewer multiplies)') if wino else
                           ('achieved = executed FLOPs (dense reference count minus the 64-deep 1x1 branch chunks skipped on tiles '
                            'whose partition plane is all zero; = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512) / HIP-event launch time; '
                            'algorithmic_* = the dense reference count 2*K*64*H*W over the same time'),
             'winograd': bool(wino),
             'partition_branches_needed_per_tile': branches, 'partition_branch_chunks_run_per_tile': run,
             'traffic': _launch_weighted_traffic(pmc, ('conv3x3_wino_quad_kernel' if units else 'conv3x3_wino_kernel') if wino
                                                 else 'conv3x3_persist_kernel' if big else 'conv3x3_mfma_kernel<2,2,1,2>',
                                                 min_bytes=0.01 * 512 * h * w),      # (a gated return reads a few KB)
             'algorithmic_bytes_per_launch': block_conv_algorithmic_bytes(h, w, cfg['num_blocks']),
             'traffic_source': pmc_src,
             'launches': cb['launches'], 'avg_launch_us': 1e3 * cb['ms'] / max(cb['launches'], 1),
             'all_convs_TFLOPs': conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
             'device_ms_per_step': dev_ms}
        r['traffic_ratio'] = (r['traffic'] / r['algorithmic_bytes_per_launch']) if r['traffic'] else None
        if dense_prof is not None and dense_prof['conv_block']['ms'] > 0:
            d = dense_prof['conv_block']
            dach = d['work'] / (d['ms'] * 1e-3) / 1e12
            r['dense_par_TFLOPs'] = dach
            r['frac_dense_par'] = dach / PEAK_F32_MFMA_TFLOPS
        res['roofline'] = r
    elif precision == 'f16x3':
        # Split fp16: three fp16 MFMAs per product (executed = 3 x the dense algorithmic count, branch skipping discounted as
        # in fp32), fp32 maps in HBM (front 12 + 256 + 256, back 256 + 256 + 256 bytes per pixel and block).  Both roofs are
        # close (0.20 ns/px of matrix work at 2.5 PFLOP/s vs 0.16 ns/px of HBM traffic); the matrix pipe is the larger one.
        fl = torch.stack([par_tile_flags(a['partitions'][0, i]) for i in range(T)])
        run = sum(((fl >> j) & 1) for j in range(3)).float().mean().item()
        dense = nb * (2 * 576 + 192) + 576
        executed = 3 * ach * (1 - nb * 64 * (3 - run) / dense)
        per_block = 1292
        bytes_frame = h * w * (nb * per_block + 512)
        gbs = bytes_frame * T * steps * a['lq'].shape[0] / (cb['ms'] * 1e-3) / 1e9 if cb['ms'] > 0 else 0.0
        res['roofline'] = {
            'kernel': 'conv3x3_f16x3_kernel<PAR> (64->64 BAE-block convs + conv_hr; operands split hi + lo/2048, three fp16 MFMA '
                      '16x16x32 per product, fp32 accumulate, fp32 maps; one 8x16 tile per 4-wave block, weight chunks through '
                      'a 3-slot LDS ring, fragment reads dealt into the MFMA gaps, 2 blocks per CU)',
            'bound': 'mfma', 'achieved': executed, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': executed / PEAK_F16_MFMA_TFLOPS, 'algorithmic_TFLOPs': ach,
            'fp32_equivalent_frac_of_fp32_peak': ach / PEAK_F32_MFMA_TFLOPS,
            'definition': 'achieved = executed fp16 FLOPs (3 x the dense reference count, minus the 1x1 branch chunks skipped on '
                          'tiles whose partition plane is all zero) / HIP-event launch time; algorithmic_TFLOPs = the dense '
                          'reference count over the same time',
            'partition_branch_chunks_run_per_tile': run,
            'hbm_GBs': gbs, 'hbm_frac': gbs / PEAK_HBM_GBS, 'algorithmic_bytes_per_pixel_per_block': per_block,
            'traffic': _launch_weighted_traffic(pmc, 'conv3x3_f16x3_kernel'), 'traffic_source': pmc_src,
            'launches': cb['launches'], 'avg_launch_us': 1e3 * cb['ms'] / max(cb['launches'], 1),
            'all_convs_TFLOPs': conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
            'device_ms_per_step': dev_ms}
    else:
        # At the fp16 matrix rate the block convs are HBM-bound: price them in algorithmic bytes.  Per frame: nb BAE
        # blocks (two launches: front 256 + 12 [3 partition planes] + 128 [fp16 map], back 128 + 256 [residual] + 256 = 1036)
        # + conv_hr (read 256, write 128 fp16; x16 pixels behind
        # the x4 heads).
        per_block = 1036
        bytes_frame = h * w * (nb * per_block + (16 if vsr else 1) * 384)
        gbs = bytes_frame * T * steps * a['lq'].shape[0] / (cb['ms'] * 1e-3) / 1e9 if cb['ms'] > 0 else 0.0
        f16_traffic = _launch_weighted_traffic(pmc, 'conv3x3_f16_kernel' if big else 'conv3x3_f16_small_kernel')
        res['roofline'] = {
            'kernel': ('conv3x3_f16_kernel<PAR,LR4,SRC16,OUT16> (64->64 BAE-block convs + conv_hr; fp16 MFMA 32x32x16, weights resident '
                       'in LDS, persistent strips, two groups in anti-phase)' if big else
                       'conv3x3_f16_small_kernel<PAR,SRC16,OUT16,G> (64->64 BAE-block convs + conv_hr on frames under 1024 tiles; fp16 MFMA '
                       '32x32x16, one tile per 4-wave group, weight chunks streamed through a 3-slot LDS ring, 3 blocks per CU)'),
            'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
            'algorithmic_bytes_per_pixel_per_block': per_block,
            'traffic': f16_traffic, 'traffic_source': pmc_src, 'launches': cb['launches'],
            'avg_launch_us': 1e3 * cb['ms'] / max(cb['launches'], 1),
            'matrix_TFLOPs': ach, 'matrix_peak_TFLOPs': PEAK_F16_MFMA_TFLOPS, 'device_ms_per_step': dev_ms,
            'note': 'priced against HBM because that is the smaller of the two roofs at the fp16 matrix rate; what actually bounds it '
                    '(DESIGN.md 3.4 item 5, profiles/r03_ub_mfma_issue.txt): a wave issues in order and each 1-KiB fragment read costs '
                    'it 16 cycles of issue, so one wave per SIMD at 1.5 reads per MFMA runs 60 cycles per MFMA (two waves: 44.7 per '
                    'SIMD); the matrix phase at that rate is balanced against the memory phase of the partner group'}
    r = res['roofline']
    if ms_per_step:
        if r['bound'] == 'mfma':
            mult = 3.0 if precision == 'f16x3' else 1.0     # split fp16 executes three fp16 MFMAs per product
            wall = mult * conv_fl / steps / (ms_per_step * 1e-3) / 1e12
        else:
            wall = bytes_frame * T * a['lq'].shape[0] / (ms_per_step * 1e-3) / 1e9
        r['achieved_wall'] = wall
        r['frac_wall'] = wall / r['peak']
        if clips > 1:
            r['per_launch_achieved'], r['per_launch_frac'] = r['achieved'], r['frac']
            r['achieved'], r['frac'] = wall, wall / r['peak']
            r['definition'] = ('clips run concurrently on side streams, launches overlap: achieved / frac = algorithmic work per step '
                               '(all convs' + (' x 3 MFMAs per product' if precision == 'f16x3' else '') + ') / wall time per step; '
                               'per_launch_* = work / sum of HIP-event launch durations (undersells overlapping launches)')
    if wp['launches']:
        gbs = wp['work'] / (wp['ms'] * 1e-3) / 1e9
        res['roofline_mv_warp'] = {'kernel': 'mv_warp_nhwc64_kernel (MV-guided bilinear alignment)', 'bound': 'hbm',
                                   'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
                                   'traffic': _launch_weighted_traffic(pmc, 'mv_warp_nhwc'),
                                   'traffic_source': pmc_src, 'launches': wp['launches'],
                                   'avg_launch_us': 1e3 * wp['ms'] / wp['launches'],
                                   'algorithmic_bytes_per_launch': wp['work'] / wp['launches']}
    if dc['launches']:
        gbs = dc['work'] / (dc['ms'] * 1e-3) / 1e9
        res['roofline_dcn'] = {'kernel': 'dcn_window_kernel<%s> (modulated deformable alignment: per-tap bilinear gather of 16 groups '
                                         'from per-half LDS windows + 64x576 MFMA contraction, %s)'
                                         % (('true', 'fp16 MFMA operands') if precision == 'fp16' else ('false', 'exact fp32 MFMA')), 'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS,
                               'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
                               'traffic': _launch_weighted_traffic(pmc, 'dcn_window_kernel'), 'traffic_source': pmc_src,
                               'launches': dc['launches'], 'avg_launch_us': 1e3 * dc['ms'] / dc['launches'],
                               'algorithmic_bytes_per_launch': dc['work'] / dc['launches'],
                               'algorithmic_bytes_per_pixel': 2240}
    return res


def timed_steps(step, steps, dist=None, cdev=None):
    """The timed region of the contract: barrier + synchronize, EXACTLY `steps` steps, synchronize + barrier; returns (last
    output, this rank's seconds, MAX over ranks).  The collectives run over whatever backend the process group has (RCCL on the
    GPU boxes; gloo in the CPU test of the 8-rank path, where there is no device to synchronize)."""
    import torch
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max = elapsed
    if dist is not None:
        et = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed_max = float(et.item())
    return out, elapsed, elapsed_max


def measure(dev, sd_np, cfg, *, workload, precision, vsr, clips, graphs, steps, warmup, T, rank=0, world=1,
            kernel_events=True, dense_par=False, dist=None, cdev=None, crfs=None):
    """Build the model for (cfg, precision), make the clip, time `steps` forwards (barrier + synchronize on both sides,
    MAX over ranks) and take the per-kernel HIP-event records.  Returns (result dict, model, device inputs)."""
    import torch
    h, w = WORKLOADS[workload]
    m = build_model(cfg, sd_np, dev, precision, graphs)
    clip, a = make_inputs(1000 + rank, T, h, w, dev, clips, crfs)   # clip `rank` of the synthetic set (sampler rule: idx[rank::world])

    def step(inp=a):
        with torch.no_grad():
            return m(inp['lq'], inp['QPs'], inp['slices'], inp['mvs'], inp['base_QPs'], inp['partitions'])

    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    # Two HIP events per launch cost ~1 % of a 720p step but 50-90 % of a 128x128 one (hundreds of launches of a few
    # us): below 720p the per-kernel timing runs as a separate pass of the same steps right after the timed region.
    events_inside = kernel_events and workload == '720p'
    if events_inside:
        m.profile(True)
    out, elapsed, elapsed_max = timed_steps(step, steps, dist, cdev)
    ms_events_pass = None
    if kernel_events and not events_inside:
        m.profile(True)
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms_events_pass = 1e3 * (time.perf_counter() - te) / steps      # the wall time the per-launch durations below add up inside
    prof = m.profile_read() if kernel_events else None
    m.profile(False)
    dense_prof = None
    if dense_par and kernel_events:
        from pnp_vcve_amd import synthetic as syn
        dn = dict(a)
        dn['partitions'] = torch.from_numpy(syn.uniform(77, 'dense_par', tuple(a['partitions'].shape), 0.05, 1.0) / 255.0).to(dev)
        step(dn)
        torch.cuda.synchronize()
        m.profile(True)
        step(dn)
        torch.cuda.synchronize()
        dense_prof = m.profile_read()
        m.profile(False)
    gt = a['gt']
    if vsr:        # synthetic HR ground truth: the LR one, nearest-upsampled (only feeds the gathered metric)
        gt = gt.repeat_interleave(4, -1).repeat_interleave(4, -2).contiguous()
    psnr = gpu_psnr(out, gt)
    frames_rank = steps * T * clips
    res = {'value': world * frames_rank / elapsed_max, 'ms_per_step': 1e3 * elapsed_max / steps,
           'elapsed_rank': elapsed, 'psnr_rank': psnr, 'frames_per_s_rank': frames_rank / elapsed,
           'kernel_events': ('none' if not kernel_events else 'inside the timed region' if events_inside
                             else 'separate pass of the same steps after the timed region'),
           'ms_per_step_events_pass': ms_events_pass,
           'launches_per_frame': (sum(v['launches'] for v in prof.values()) / (steps * T * clips)) if prof else None}
    if prof is not None:
        ptag = {'fp32': '', 'fp16': 'fp16_', 'f16x3': 'f16x3_'}[precision]
        tag = ptag if workload == '720p' and not vsr and cfg.get('deform', 'vos') == 'vos' \
            else f'{workload}_' + ptag + ('vsr_' if vsr else '') + ('' if cfg.get('deform', 'vos') == 'vos' else cfg['deform'] + '_')
        pmc, pmc_src = committed_pmc_traffic(tag)
        res.update(rooflines(m, prof, cfg, a, T, h, w, steps, precision, vsr, pmc, pmc_src, dense_prof,
                             ms_per_step=res['ms_per_step'], clips=clips))
    return res, m, a


def workload_text(clips, T, h, w, workload, cfg, precision):
    return (f'{clips} x {T}x3x{h}x{w} clip per GPU per step ({WORKLOAD_NOTE[workload]}), full BAE+CAA forward, config '
            f'HR_davis_LR_128x128 generator, seeded random weights; side info per SURVEY 8(d): partition class ~ U{{0,1,2}} per '
            f'8x8 block (one-hot/255, none on the I frame), quarter-pel block MVs, IBBBP cadence')


def main(argv=None, measure_fn=None):
    """`measure_fn` replaces measure() in the CPU test of the rank path (tests/test_host_logic.py: 8 gloo ranks, no GPU); the
    script itself always runs the real one and refuses to start without a GPU."""
    args = parse_args(argv)
    global F16_MIRRORS, TILE_QUEUE, WINOGRAD
    WINOGRAD = args.winograd
    F16_MIRRORS = args.f16_mirrors
    TILE_QUEUE = args.tile_queue
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))

    # stdout belongs to the ONE result line.  Libraries print there too (RCCL writes a version banner to the C-level stdout, which
    # is flushed at exit, i.e. BEHIND the line; gloo prints its connection lines): file descriptor 1 is pointed at stderr for
    # the whole run and the line is written to the saved descriptor at the very end.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np   # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('PNP_DIST_BACKEND', 'nccl')     # 'nccl' is RCCL on ROCm; 'gloo' only for 1-GPU dry runs
    ndev = max(torch.cuda.device_count(), 1)                  # counting devices does not initialise the GPU
    local = local % ndev
    dev = torch.device('cuda', local)
    # A torch.distributed environment (RANK / WORLD_SIZE from torch.distributed.run) means "be a rank": the process group
    # is created and every collective below runs over it even at WORLD_SIZE=1, so the RCCL path (init with device_id,
    # barrier, all_reduce(MAX), all_gather on device tensors, destroy) is the same code at N = 1 and N = 8.
    grouped = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if measure_fn is None:
        if not torch.cuda.is_available():
            raise RuntimeError('bench.py needs an MI355X: no GPU is visible (there is no CPU path to measure)')
        measure_fn = measure
        torch.cuda.set_device(dev)
    if grouped:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == 'nccl' else torch.device('cpu')   # where the tiny collective payloads live

    from pnp_vcve_amd import synthetic as syn
    cfg = dict(syn.DEFAULT_GENERATOR_CFG)
    cfg['vsr'] = bool(args.vsr)
    if args.deform != 'vos':
        cfg['deform'] = args.deform
    sd_np = syn.make_state_dict(cfg, seed=2025)
    h, w = WORKLOADS[args.workload]
    T = args.frames
    headline = args.workload == '720p' and args.precision == 'fp32' and not args.vsr and args.deform == 'vos'

    r, m, a = measure_fn(dev, sd_np, cfg, workload=args.workload, precision=args.precision, vsr=args.vsr, clips=args.clips,
                      graphs=args.graphs, steps=args.steps, warmup=args.warmup, T=T, rank=rank, world=world,
                      kernel_events=not args.no_kernel_events, dense_par=headline and world == 1,
                      dist=dist if grouped else None, cdev=cdev)

    # per-rank metrics, gathered with one small collective (PSNR, frames/s): mmedit/apis/test.py:211-233
    mine = torch.tensor([r['psnr_rank'], r['frames_per_s_rank']], dtype=torch.float64, device=cdev)
    if grouped:
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        allm = torch.stack(allm).cpu().numpy()
    else:
        allm = mine.cpu().numpy()[None]

    if rank == 0:
        res = {
            'metric': 'enhanced frames/sec (1280x720, 7-frame window)' if args.workload == '720p'
                      else f'enhanced frames/sec ({w}x{h}, {T}-frame window)',
            'value': r['value'], 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': r['ms_per_step'], 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPE_TEXT[args.precision],
            'data': 'synthetic',
            'config': {'workload': workload_text(args.clips, T, h, w, args.workload, cfg, args.precision),
                       'vsr_x4_heads': bool(args.vsr), 'hip_graphs': bool(args.graphs), 'deform': args.deform,
                       'parallelism': f'clip-sharded replicas x{world}', 'frames_per_step_per_gpu': T * args.clips},
            'kernel_events': r['kernel_events'], 'launches_per_frame': r['launches_per_frame'],
            # which collectives really ran: with a torch.distributed environment the barrier / all_reduce(MAX) / all_gather
            # above went over this backend ('nccl' = RCCL) -- at WORLD_SIZE=1 too
            'dist': {'process_group': grouped, 'backend': dist.get_backend() if grouped else None,
                     'rccl_version': ('.'.join(str(v) for v in torch.cuda.nccl.version())
                                      if grouped and backend == 'nccl' else None),
                     'collectives': ['barrier', 'all_reduce(MAX)', 'all_gather', 'barrier'] if grouped else []},
            'psnr_per_rank': [float(x) for x in allm[:, 0]],
            'frames_per_s_per_rank': [float(x) for x in allm[:, 1]],
        }
        for k in ('roofline', 'roofline_mv_warp', 'roofline_dcn'):
            if k in r:
                res[k] = r[k]
        if world == 1 and not args.no_cpu_baseline:
            cclip, ref, dt = cpu_baseline(sd_np, cfg, h, w)
            ca = {k: torch.from_numpy(v).to(dev) for k, v in cclip.items()}
            with torch.no_grad():
                got = m(ca['lq'], ca['QPs'], ca['slices'], ca['mvs'], ca['base_QPs'], ca['partitions']).cpu()
            from oracle import cpu_ref
            gt = torch.from_numpy(cclip['gt'])
            if args.vsr:
                gt = gt.repeat_interleave(4, -1).repeat_interleave(4, -2)
            res['cpu_baseline'] = {
                'value': 2 / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'threads': torch.get_num_threads(),
                'host_logical_cpus': os.cpu_count(), 'kind': 'port',
                'sample': f'oracle/cpu_ref.py (PyTorch-CPU fp32 restatement of the reference, pinned by tests/golden) on '
                          f'one 2x3x{h}x{w} clip (2 of the 7 frames, same frame size; both frames are sequence ends, so the '
                          f'sample holds 2 of the clip\'s 12 alignment calls -- 0.7 % of the CPU time) = {dt:.1f} s; threads '
                          f'calibrated over 8/16/32 on this host ({os.cpu_count()} logical CPUs)',
                'sample_seconds': dt}
            res['parity'] = {'sample': f'2x3x{h}x{w}', 'max_abs_diff_vs_cpu': float((got - ref).abs().max()),
                             'psnr_delta_db': cpu_ref.clip_psnr(got, gt) - cpu_ref.clip_psnr(ref, gt),
                             'gate': 1e-3}
        elif world > 1 and not args.no_cpu_baseline:
            # an N > 1 line still carries the CPU leg: rank 0 alone times the oracle on the 128x128 clip of BASELINE
            # configs[0] AFTER the timed region (the other ranks wait in the final barrier below); ~5 s, never at the frame
            # size of the GPU workload (that sample is the N = 1 line's)
            cb = cpu_baseline_128(T, runs=3)
            cb['sample'] += '; timed on rank 0 after the timed region of an N > 1 run (the N = 1 line times the workload\'s own frame size)'
            res['cpu_baseline'] = cb
        del m, a
        torch.cuda.empty_cache()
        full = None
        if world == 1 and headline and not args.no_secondary and not args.no_kernel_events:
            # the other BASELINE workloads: measured after the headline, written beside this file and to stderr -- never on stdout
            try:
                sec = secondary_workloads(dev, T, args.no_cpu_baseline)
                res['north_star_128'] = north_star_128(sec)
                res['opt_in_720p'] = opt_in_720p(sec)
                res['secondary_file'] = os.path.basename(SECONDARY_FILE)
                full = dict(res, secondary=sec)
            except Exception as e:                      # the headline line must still be printed
                print(f'bench.py: secondary workloads failed: {e!r}', file=sys.stderr, flush=True)
                res['north_star_128'] = None
        if full is not None:
            text = json.dumps(strict(full, digits=0), allow_nan=False, indent=1)
            try:
                with open(SECONDARY_FILE, 'w') as fh:
                    fh.write(text + '\n')
            except OSError as e:
                print(f'bench.py: cannot write {SECONDARY_FILE}: {e}', file=sys.stderr)
            print(text, file=sys.stderr, flush=True)
        os.write(line_fd, (bounded_line(res) + '\n').encode())
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    os.close(line_fd)


def north_star_128(sec):
    """compact summary of the 7x3x128x128 entries of `secondary` for the stdout line (north_star: "throughput on synthetic
    7x3x128x128 clips ... alongside the reference CPU path timed on the host cores (core count stated)")."""
    by = {(e.get('workload'), e.get('precision'), e.get('clips_per_step'), e.get('hip_graphs')): e for e in sec if 'workload' in e}
    one, eight, graph = by.get(('128', 'fp32', 1, False)), by.get(('128', 'fp32', 8, False)), by.get(('128', 'fp32', 8, True))
    x3 = by.get(('128', 'f16x3', 1, False))
    out = {'unit': 'frames/s', 'dtype': 'f32'}
    if one:
        out['clips_1'] = one['value']
        out['frac_per_launch'] = one['roofline']['frac']                    # executed FLOPs (Winograd: 256/576 of the 3x3 part) per launch time
        out['algorithmic_frac_per_launch'] = one['roofline'].get('algorithmic_frac')
        out['winograd'] = one['roofline'].get('winograd')
        out['frac_wall'] = one['roofline'].get('frac_wall')                 # algorithmic FLOPs over wall time
        cb = one.get('cpu_baseline')
        if cb:
            out['cpu'] = cb['value']
            out['cpu_cores'] = cb['cores']
            out['speedup_vs_cpu'] = one['value'] / cb['value']
    if eight:
        out['clips_8'] = eight['value']
        out['frac_wall_clips_8'] = eight['roofline'].get('frac_wall')
    if graph:
        out['clips_8_hipgraph'] = graph['value']
    if x3:
        out['clips_1_f16x3'] = x3['value']
    return out


def opt_in_720p(sec):
    """compact summary of the opt-in arithmetics at the headline shape (the headline `value` itself is exact fp32)"""
    out = {'unit': 'frames/s'}
    for e in sec:
        if e.get('workload') == '720p' and e.get('precision') in ('fp32', 'f16x3') and e.get('clips_per_step') == 2:
            out[e['precision'] + '_two_clips_interleaved'] = e['value']
        if e.get('workload') == '720p' and e.get('precision') in ('fp16', 'f16x3') and e.get('clips_per_step') == 1:
            out[e['precision']] = e['value']
            out[e['precision'] + '_frac'] = e['roofline']['frac']
            out[e['precision'] + '_bound'] = e['roofline']['bound']
    return out


def pipeline_e2e(dev, T, clips=10, workers=16, precision='fp32'):
    """The whole evaluation loop of tools/test.py on an on-disk tree in the reference's REDS layout (restorers/basicvsr.py:155-231,
    apis/test.py:100-119): PNG + MV-record decode on a loader thread and H2D on a side stream one clip ahead (ClipPrefetcher),
    MV / partition maps painted on the GPU (pnp_rasterise_side_info_f32), the fp32 generator, PSNR + SSIM on the device, enhanced
    frames to uint8 on the device and PNG-encoded on a thread pool (FrameWriter).  Reports frames/s over the whole loop including
    the final PNG drain, and where the main thread waited."""
    import shutil
    import tempfile
    import torch
    from pnp_vcve_amd import restorer, synthetic as syn   # noqa: F401
    from pnp_vcve_amd.apis import ClipPrefetcher
    from pnp_vcve_amd.datasets import build_dataset
    from pnp_vcve_amd.io_async import FrameWriter
    from pnp_vcve_amd.registry import build_model
    h, w = WORKLOADS['720p']
    root = tempfile.mkdtemp(prefix='pnp_e2e_')
    try:
        t_tree = time.perf_counter()
        lq, gt, qp = syn.write_clip_tree(os.path.join(root, 'data'), clips=[f'{i:03d}' for i in range(clips + 1)], t=T, h=h, w=w)
        t_tree = time.perf_counter() - t_tree
        ds = build_dataset(dict(type='SRREDSMultipleGTCompressDataset', lq_folder=lq, gt_folder=gt, num_input_frames=100,
                                pipeline=[dict(type='LoadImageFromFileList_ipb', qp_slice_file=qp)], scale=1,
                                val_partition='REDS4', test_mode=True))
        cfg = dict(syn.DEFAULT_GENERATOR_CFG)
        model = build_model(dict(type='BasicVSR', generator=dict(type=GEN_TYPE, **cfg), pixel_loss=dict(type='CharbonnierLoss')),
                            train_cfg=None, test_cfg=dict(metrics=['PSNR', 'SSIM'], crop_border=0))
        sd_np = syn.make_state_dict(cfg, seed=2025)
        model.generator.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
        model = model.to(dev).eval()
        model.precision = precision
        out_dir = os.path.join(root, 'out')
        with FrameWriter(max_workers=workers) as writer:
            model.frame_writer = writer
            data = next(iter(ClipPrefetcher(ds, [0], dev)))   # clip 0 is the warm-up (first-use allocations, kernel attributes)
            with torch.no_grad():
                model(test_mode=True, save_image=True, save_path=out_dir, **data)
            torch.cuda.synchronize()
            writer.drain()                                    # the warm-up clip's PNG encodes must not overlap the timed loop
            # the timed clips' loader starts INSIDE the timed window: the first clip's decode + H2D is fully exposed, as it is
            # for the first clip of a real run
            t0 = time.perf_counter()
            # two clips in flight like tools/test.py's default: a pair goes through the generator as one batch (two streams: each
            # clip's launches fill the other's tails), metrics / PNG submission stay clip by clip (apis.multi_gpu_test)
            from pnp_vcve_amd.apis import GENERATOR_INPUTS, _pairable
            it = iter(ClipPrefetcher(ds, range(1, len(ds)), dev, depth=2))
            stall, fwd, psnr, n = [], 0.0, [], 0

            def take():
                a = time.perf_counter()
                d = next(it, None)
                stall.append(time.perf_counter() - a)
                return d

            def finish(d, out=None):
                nonlocal n
                with torch.no_grad():
                    kw = {} if out is None else {'precomputed_output': out}
                    res = model(test_mode=True, save_image=True, save_path=out_dir, **kw, **d)
                psnr.append(res['eval_result']['PSNR'])
                n += 1
                return 0.0 if out is not None else model.last_forward_seconds

            while True:
                d0 = take()
                if d0 is None:
                    break
                d1 = take()
                if d1 is not None and _pairable(d0, d1):
                    with torch.no_grad():
                        torch.cuda.synchronize()
                        a = time.perf_counter()
                        out = model.generator(*[torch.cat([d0[k], d1[k]]) for k in GENERATOR_INPUTS])
                        torch.cuda.synchronize()
                        fwd += time.perf_counter() - a
                    finish(d0, out[0:1])
                    finish(d1, out[1:2])
                else:
                    fwd += finish(d0)
                    if d1 is not None:
                        fwd += finish(d1)
                    else:
                        break
            t_loop = time.perf_counter() - t0
            a = time.perf_counter()
        drain = time.perf_counter() - a                        # FrameWriter.close(): the PNG encodes still queued
        total = time.perf_counter() - t0
        model.frame_writer = None
        pngs = sum(len(f) for _, _, f in os.walk(out_dir))
        return {'name': f'end-to-end tools/test.py loop on an on-disk REDS-layout tree: {n} clips x {T}x3x{h}x{w} {precision} (PNG + MV records from '
                        f'disk -> GPU rasteriser -> generator -> PSNR + SSIM on the device -> async PNG write-back)',
                'metric': f'enhanced frames/sec ({w}x{h}, {T}-frame window), whole pipeline', 'value': n * T / total, 'unit': 'frames/s',
                'frames_per_s_before_the_final_png_drain': n * T / t_loop,
                'clips': n, 'frames': n * T, 'seconds_total': total, 'seconds_generator_forward': fwd,
                'seconds_main_thread_waiting_for_loader_h2d': sum(stall),
                'seconds_main_thread_waiting_for_loader_h2d_per_clip': stall,
                'seconds_png_drain_after_last_clip': drain,
                'seconds_metrics_and_uint8_d2h_and_submit': t_loop - fwd - sum(stall),
                'dtype': DTYPE_TEXT[precision], 'png_workers': workers, 'pngs_written': pngs, 'psnr': float(sum(psnr) / max(len(psnr), 1)),
                'seconds_writing_the_synthetic_tree_untimed': t_tree,
                'note': 'clip 0 is an untimed warm-up on its own loader; the timed clips\' loader is created inside the timed window (the first '
                        'clip\'s decode + H2D is fully exposed); after that it stages two clips ahead, so decode + H2D are visible only '
                        'where they exceed the previous pair\'s GPU time; two clips in flight (one generator call per pair, two streams)'}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def secondary_workloads(dev, T, no_cpu_baseline=False):
    """The other workloads BASELINE.json names, each with its own timed region (same barrier-free N = 1 protocol:
    warm-up, synchronize, K steps, synchronize), roofline and kernel-event mode."""
    import torch
    from pnp_vcve_amd import synthetic as syn
    out = []
    specs = [
        dict(name='7x3x128x128 fp32, 1 clip (north_star / configs[0-1])', workload='128', precision='fp32', vsr=False, clips=1,
             steps=20, warmup=3),
        dict(name='7x3x128x128 fp32, 8 clips per step (8 concurrent contexts)', workload='128', precision='fp32', vsr=False,
             clips=8, steps=5, warmup=2),
        dict(name='7x3x128x128 fp32, 8 clips per step, each clip replayed as one hipGraph', workload='128', precision='fp32',
             vsr=False, clips=8, steps=5, warmup=2, graphs=True, kernel_events=False),
        dict(name='7x3x720x1280 fp16 MFMA convs (the headline shape with the opt-in fp16 operands)', workload='720p', precision='fp16',
             vsr=False, clips=1, steps=3, warmup=1),
        dict(name='7x3x720x1280 split-fp16 convs (the headline shape; opt-in PNP_PREC_F16X3: fp32-level results, inside the 1e-3 gate, '
                  'from the fp16 matrix pipe)', workload='720p', precision='f16x3', vsr=False, clips=1, steps=3, warmup=1),
        dict(name='7x3x180x320 fp16 MFMA convs, mixed crf15/25/35 batch of 3 (configs[4], vsr=False as the config ships)',
             workload='lr180', precision='fp16', vsr=False, clips=3, steps=5, warmup=2, crfs=[15, 25, 35]),
        dict(name='7x3x180x320 -> 720x1280 fp16 MFMA convs, x4 heads, mixed crf15/25/35 batch of 3 (configs[4] as described)',
             workload='lr180', precision='fp16', vsr=True, clips=3, steps=5, warmup=2, crfs=[15, 25, 35]),
        dict(name='7x3x128x128 split-fp16 convs, 1 clip (north_star shape; opt-in PNP_PREC_F16X3, fp32-level results)', workload='128',
             precision='f16x3', vsr=False, clips=1, steps=20, warmup=3),
        dict(name='7x3x720x1280 fp32, 2 clips per step interleaved on two streams (each fills the partial last round of tiles and the '
                  'dispatch gaps of the other\'s conv launches; bit-identical to one at a time)', workload='720p', precision='fp32',
             vsr=False, clips=2, steps=3, warmup=1),
        dict(name='7x3x720x1280 split-fp16 convs, 2 clips per step interleaved on two streams (as above; each context has its own tile '
                  'queue)', workload='720p', precision='f16x3', vsr=False, clips=2, steps=3, warmup=1),
    ]
    specs += [
        dict(name='100x3x720x1280 fp32, one clip: the reference configs\' real clip length (configs/HR_davis_LR_128x128.py:202 '
                  'num_input_frames=100; 24.5 GB workspace, 100 live frame slots)', workload='720p', precision='fp32', vsr=False, clips=1,
             steps=2, warmup=1, frames=100),
        dict(name='7x3x720x1280 fp32 with PNP_OPT_WINOGRAD = 0: the direct implicit-GEMM kernels of rounds 1-4 (exact fp32 MFMA 32x32x2), '
                  'same session as the headline', workload='720p', precision='fp32', vsr=False, clips=1, steps=3, warmup=1, winograd=0),
        dict(name='7x3x180x320 fp32, 1 clip (240 16x16 tiles: the mid-size range, persistent Winograd tile kernels on less than one '
                  'round of tiles)', workload='lr180', precision='fp32', vsr=False, clips=1, steps=5, warmup=2),
        dict(name='7x3x180x320 fp32 with PNP_OPT_WINOGRAD = 0, same session', workload='lr180', precision='fp32', vsr=False, clips=1,
             steps=5, warmup=2, winograd=0),
    ]
    cpu128 = None if no_cpu_baseline else cpu_baseline_128(T)
    for sp in specs:
        cfg = dict(syn.DEFAULT_GENERATOR_CFG)
        cfg['vsr'] = sp['vsr']
        sd_np = syn.make_state_dict(cfg, seed=2025)
        h, w = WORKLOADS[sp['workload']]
        global WINOGRAD
        keep_w, Tsp = WINOGRAD, sp.get('frames', T)
        # workspace of the entry (T + 3 maps of 256 B per pixel, inputs, outputs, the allocator's slack): skipped, not failed, where it cannot fit
        need = (Tsp + 3) * h * w * 256 * sp['clips'] * (2 if sp['precision'] != 'fp32' else 1) * 1.3 + 2e9
        free = torch.cuda.mem_get_info()[0]
        if need > free:
            out.append({'name': sp['name'], 'workload': sp['workload'], 'precision': sp['precision'], 'value': None, 'unit': 'frames/s',
                        'frames_per_clip': Tsp, 'clips_per_step': sp['clips'], 'vsr_x4_heads': sp['vsr'], 'hip_graphs': sp.get('graphs', False),
                        'skipped': f'needs ~{need / 1e9:.1f} GB of device memory, {free / 1e9:.1f} GB free'})
            continue
        if 'winograd' in sp:
            WINOGRAD = sp['winograd']
        try:
            r, m, a = measure(dev, sd_np, cfg, workload=sp['workload'], precision=sp['precision'], vsr=sp['vsr'],
                              clips=sp['clips'], graphs=sp.get('graphs', False), steps=sp['steps'], warmup=sp['warmup'], T=Tsp,
                              crfs=sp.get('crfs'), kernel_events=sp.get('kernel_events', True))
        finally:
            WINOGRAD = keep_w
        e = {'name': sp['name'], 'workload': sp['workload'], 'precision': sp['precision'],
             'metric': f'enhanced frames/sec ({w}x{h}, {Tsp}-frame window)', 'value': r['value'], 'frames_per_clip': Tsp,
             'unit': 'frames/s', 'ms_per_step': r['ms_per_step'], 'steps': sp['steps'], 'warmup': sp['warmup'],
             'dtype': DTYPE_TEXT[sp['precision']],
             'clips_per_step': sp['clips'], 'vsr_x4_heads': sp['vsr'], 'hip_graphs': sp.get('graphs', False),
             'kernel_events': r['kernel_events'], 'ms_per_step_events_pass': r['ms_per_step_events_pass'],
             'launches_per_frame': r['launches_per_frame'], 'psnr': r['psnr_rank']}
        for k in ('roofline', 'roofline_mv_warp'):
            if k in r:
                e[k] = r[k]
        if sp['workload'] == '128' and sp['precision'] == 'fp32' and cpu128 is not None:
            e['cpu_baseline'] = cpu128          # north_star: "alongside the reference CPU path timed on the host cores"
            e['speedup_vs_cpu'] = r['value'] / cpu128['value']
        out.append(e)
        del m, a
        torch.cuda.empty_cache()
    # the loop twice: a process's FIRST pass fills the caching allocators inside the timed window (pinned host staging buffers, the
    # device tensors of the clips in flight, the second workspace context: hipHostMalloc / hipMalloc beside running kernels) -- 58.5
    # frames/s against 66.6 for every later pass (tools/e2e_probe.py, r06); an evaluation run has hundreds of clips, so the entry's value
    # is the steady pass and the first one is kept beside it
    first = pipeline_e2e(dev, T)
    torch.cuda.empty_cache()
    e2e = pipeline_e2e(dev, T)
    e2e['first_pass_of_the_process'] = {k: first[k] for k in ('value', 'seconds_total', 'seconds_generator_forward')}
    e2e['note'] += ('; `value` is the SECOND pass of the loop in this process (allocators warm: the steady state of a long evaluation), '
                    'first_pass_of_the_process the cold one')
    out.append(e2e)
    torch.cuda.empty_cache()
    return out


if __name__ == '__main__':
    main()

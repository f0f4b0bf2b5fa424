"""GPU: the BasicVSR wrapper and the test driver, driven by the config keys of the reference."""
import os
import subprocess
import sys

import numpy as np
import pytest
from bench_util import bench_line
import torch

from oracle import cpu_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_basicvsr_forward_test_matches_oracle_metrics():
    import pnp_vcve_amd  # noqa: F401
    from pnp_vcve_amd import restorer, synthetic as syn  # noqa: F401
    from pnp_vcve_amd.config import Config
    from pnp_vcve_amd.registry import build_model
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'HR_davis_LR_128x128.py'))
    model = build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    gcfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd_np = syn.make_state_dict(gcfg, seed=9, par_gain=10.0)
    model.generator.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    model = model.cuda().eval()
    clip = syn.make_clip(seed=91, n=1, t=4, h=64, w=80)
    data = {k: torch.from_numpy(v).cuda() for k, v in clip.items()}
    res = model(test_mode=True, lq=data['lq'], gt=data['gt'], QPs=data['QPs'], slices=data['slices'], mvs=data['mvs'],
                base_QPs=data['base_QPs'], partitions=data['partitions'])
    assert set(res['eval_result']) == {'PSNR', 'SSIM'}
    c = {k: torch.from_numpy(v) for k, v in clip.items()}
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), gcfg, c['lq'], c['QPs'], c['slices'], c['mvs'],
                                        c['base_QPs'], c['partitions'])
    assert abs(res['eval_result']['PSNR'] - cpu_ref.clip_psnr(ref, c['gt'])) < 0.01
    # without metrics the wrapper returns CPU tensors (basicvsr.py:199-202)
    model.test_cfg = None
    res = model(test_mode=True, lq=data['lq'], gt=data['gt'], QPs=data['QPs'], slices=data['slices'], mvs=data['mvs'],
                base_QPs=data['base_QPs'], partitions=data['partitions'])
    assert not res['output'].is_cuda and float((res['output'] - ref).abs().max()) < 1e-4
    with pytest.raises(NotImplementedError):
        model(lq=data['lq'], gt=data['gt'])          # training path is out of scope


def test_test_driver_cli(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'test.py'),
                          os.path.join(ROOT, 'configs', 'HR_davis_LR_128x128.py'), 'none',
                          '--cfg-options', 'data.test.num_clips=2', 'data.test.num_input_frames=3',
                          'data.test.height=64', 'data.test.width=64', '--save-path', str(tmp_path)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'Eval-PSNR' in out.stdout and 'Eval-SSIM' in out.stdout
    assert os.path.exists(os.path.join(str(tmp_path), '000', '00000000.png'))


def test_test_driver_cli_fp16_switch_matches_fp32_psnr():
    """`--fp16` (mmcv's wrap_fp16_model idiom) drives the fp16-operand kernels; PSNR within north_star's 1e-3 dB."""
    import re
    vals = []
    for extra in ([], ['--fp16']):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'test.py'),
                              os.path.join(ROOT, 'configs', 'HR_davis_LR_128x128_IPB.py'), 'none',
                              '--cfg-options', 'data.test.num_clips=2', 'data.test.num_input_frames=5',
                              'data.test.height=128', 'data.test.width=128', '--seed', '0'] + extra,   # same random init
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        vals.append(float(re.search(r'Eval-PSNR: ([0-9.]+)', out.stdout).group(1)))
    assert vals[0] != vals[1] and abs(vals[0] - vals[1]) < 1e-3, vals


def test_folder_dataset_end_to_end_with_gpu_rasteriser(tmp_path):
    """dist_test-style run on an on-disk clip tree in the reference's layout: frames + MV records from disk,
    dense maps painted on the GPU, generator, on-device PSNR -- against the oracle fed with the oracle's maps."""
    import pnp_vcve_amd  # noqa: F401
    from pnp_vcve_amd import restorer, synthetic as syn  # noqa: F401
    from pnp_vcve_amd.apis import multi_gpu_test
    from pnp_vcve_amd.config import Config
    from pnp_vcve_amd.datasets import build_dataset
    from pnp_vcve_amd.registry import build_model
    from test_host_logic import _write_clip_tree
    lq, gt, qp, slices = _write_clip_tree(str(tmp_path), clips=('000',), t=4)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'REDS_folder_example.py'))
    cfg.merge_from_dict({'data.test.lq_folder': lq, 'data.test.gt_folder': gt})
    cfg.data.test.pipeline[1]['qp_slice_file'] = qp
    ds = build_dataset(cfg.data.test)
    model = build_model(cfg.model, train_cfg=None, test_cfg=dict(metrics=['PSNR'], crop_border=0))
    gcfg = dict(syn.DEFAULT_GENERATOR_CFG)
    sd_np = syn.make_state_dict(gcfg, seed=12, par_gain=10.0)
    model.generator.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    model = model.cuda().eval()
    out = multi_gpu_test(model, ds, device='cuda', metrics=('PSNR',))
    item = ds[0]
    mvs, par = cpu_ref.rasterise_side_info(item['mv_records'].numpy(), item['rec_frame'].numpy(), slices, 64, 64)
    with torch.no_grad():
        ref = cpu_ref.generator_forward(cpu_ref.to_torch_state(sd_np), gcfg, item['lq'][None], item['QPs'][None],
                                        item['slices'][None], torch.from_numpy(mvs)[None], item['base_QPs'][None],
                                        torch.from_numpy(par)[None])
    assert abs(out[0]['eval_result']['PSNR'] - cpu_ref.clip_psnr(ref, item['gt'][None])) < 0.01


def test_dist_test_driver_two_ranks_equals_one_rank():
    """tools/dist_test.sh with 2 ranks (both on cuda:0, gloo for the metric all-gather -- RCCL needs one GPU per rank):
    clip sharding, per-rank GPU forward, gather and re-interleave give exactly the single-process numbers."""
    import re
    common = ['--seed', '0', '--cfg-options', 'data.test.num_clips=5', 'data.test.num_input_frames=3',
              'data.test.height=64', 'data.test.width=64']
    cfgp = os.path.join(ROOT, 'configs', 'HR_davis_LR_128x128_IPB.py')
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'test.py'), cfgp, 'none'] + common,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    env = dict(os.environ, PORT='29533')
    two = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'dist_test.sh'), cfgp, 'none', '2'] + common +
                         ['dist_params.backend=gloo'], capture_output=True, text=True, timeout=600, env=env)
    assert two.returncode == 0, two.stdout + two.stderr
    get = lambda s, k: re.search(rf'Eval-{k}: ([0-9.]+)', s).group(1)      # noqa: E731
    assert 'world 2' in two.stdout and 'world 1' in one.stdout
    assert get(one.stdout, 'PSNR') == get(two.stdout, 'PSNR') and get(one.stdout, 'SSIM') == get(two.stdout, 'SSIM')


def test_bench_emits_the_contract_line():
    """bench.py on a tiny workload: ONE JSON line with the driver's keys, the roofline / cpu_baseline / parity objects."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', '128', '--frames', '3',
                          '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    d = bench_line(out.stdout)
    for k in ('roofline', 'cpu_baseline', 'parity'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['dtype'] == 'f32' and d['vs_baseline'] is None
    assert d['value'] > 0 and abs(d['value'] - 2 * 3 / (d['ms_per_step'] * 2e-3)) < 1e-4 * d['value']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-5
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0
    assert d['parity']['max_abs_diff_vs_cpu'] < d['parity']['gate']
    assert 'workload' in d['config'] and 'model' not in d['config']


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` (the driver's plain form, no torch.distributed environment) starts two ranks itself
    (tools/dist_test.sh:11-22's job), gathers per-rank metrics and prints ONE line with n_gpus == 2.  Both ranks share
    cuda:0 here and the tiny collectives go over gloo (RCCL needs one GPU per rank)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['PNP_DIST_BACKEND'] = 'gloo'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', '128', '--frames', '3',
                          '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    d = bench_line(out.stdout)
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'clip-sharded replicas x2'
    assert len(d['frames_per_s_per_rank']) == 2 and len(d['psnr_per_rank']) == 2
    assert d['psnr_per_rank'][0] != d['psnr_per_rank'][1]            # rank r ran clip r of the synthetic set
    # an N > 1 line carries the CPU leg too: the oracle on the 128x128 clip, rank 0, outside the timed region
    assert 'roofline' in d and d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0 and '128x128' in d['cpu_baseline']['sample']
    # whole-job rate = all ranks' frames over the slowest rank's time
    assert abs(d['value'] - 2 * 2 * 3 / (d['ms_per_step'] * 2e-3)) < 1e-4 * d['value']


def test_bench_headline_line_is_short_and_the_secondary_workloads_go_to_a_side_file():
    """the default (720p fp32) run: stdout is ONE strict-JSON line under 4 KB with roofline / cpu-free blocks (round 3's 21 KB line
    was truncated by the driver and never parsed); the other BASELINE workloads are measured too but land in bench_secondary.json
    beside bench.py (kept short: 1 step, no CPU baseline)."""
    import json
    side = os.path.join(ROOT, 'bench_secondary.json')
    if os.path.exists(side):
        os.remove(side)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    d = bench_line(out.stdout)
    assert d['metric'].startswith('enhanced frames/sec (1280x720') and d['dtype'] == 'f32'
    r = d['roofline']
    # frac prices EXECUTED FLOPs; algorithmic_frac (the dense reference count) can only be larger; on a dense partition map
    # nothing is skipped, so the dense-map figure is an executed figure too
    assert r['bound'] == 'mfma' and r['kernel'].startswith('conv3x3_wino_kernel') and r['winograd'] is True
    # (Winograd: the algorithmic count is the direct conv's, 2.25x what the 3x3 part executes -- above the matrix peak is the point)
    assert 0 < r['frac'] < 1 < r['algorithmic_frac'] and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-5
    assert 0 < r['frac_dense_par'] <= r['algorithmic_frac'] * 1.02 and 0 < r['frac_wall'] <= r['algorithmic_frac'] * 1.02
    assert d['roofline_mv_warp']['bound'] == 'hbm' and d['roofline_mv_warp']['frac'] > 0
    assert 'U{0,1,2}' in d['config']['workload']
    ns = d['north_star_128']
    assert ns['clips_1'] > 0 and ns['clips_8'] > 0 and ns['clips_8_hipgraph'] > 0 and 0 < ns['frac_per_launch'] < 1
    assert 'cpu' not in ns                                      # --no-cpu-baseline covers the 128x128 CPU leg too
    assert d['secondary_file'] == 'bench_secondary.json'
    oi = d['opt_in_720p']                                       # the opt-in arithmetics at the headline shape, compact
    # structure only: no rate is compared with another rate in the parity suite (tests/test_gpu_zz_perf.py holds those)
    assert oi['fp32_two_clips_interleaved'] > 0 and oi['f16x3_two_clips_interleaved'] > 0
    assert oi['f16x3'] > 0 and oi['fp16'] > 0 and oi['f16x3_bound'] == 'mfma' and oi['fp16_bound'] == 'hbm'
    with open(side) as fh:
        full = json.load(fh, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert full['value'] == d['value'] or abs(full['value'] - d['value']) < 1e-4 * d['value']
    assert 'definition' in full['roofline'] and 'device_ms_per_step' in full['roofline']      # the prose lives here, not on stdout
    sec = full['secondary']
    assert len(sec) == 15 and sec[8]['clips_per_step'] == 2 and sec[8]['workload'] == '720p' and sec[8]['roofline']['per_launch_frac'] > 0
    assert sec[9]['clips_per_step'] == 2 and sec[9]['precision'] == 'f16x3'
    # the reference configs' real clip length, and the direct kernels of rounds 1-4 in the same session
    assert sec[10]['workload'] == '720p' and '100x3x720x1280' in sec[10]['name'] and sec[10]['value'] > 0
    assert 'PNP_OPT_WINOGRAD = 0' in sec[11]['name'] and sec[11]['value'] > 0
    # mid-size frames (240 tiles): the tile kernels and the direct ones, same session
    assert sec[12]['workload'] == 'lr180' and sec[13]['workload'] == 'lr180' and sec[12]['value'] > 0 and sec[13]['value'] > 0
    e2e = sec[14]                                   # the whole tools/test.py loop on an on-disk tree
    assert e2e['pngs_written'] == (e2e['clips'] + 1) * 7 and e2e['value'] > 0 and 20 < e2e['psnr'] < 60
    assert e2e['seconds_total'] >= e2e['seconds_generator_forward'] > 0
    for e in sec[:14]:
        if e.get('skipped'):                  # an entry whose workspace does not fit the device's free memory (the 100-frame clip: 24.5 GB)
            continue
        assert e['value'] > 0
        if not e['hip_graphs']:               # per-kernel events are not taken inside a graph replay
            rf = e['roofline']
            assert rf['frac'] > 0 and rf['frac_wall'] > 0 and e['launches_per_frame'] > 0
            dom = max(rf['device_ms_per_step'].values())
            if e['clips_per_step'] == 1:
                # one clip = one stream: launches run back to back, so the device time of any kind fits inside the wall time of
                # the pass the events were taken in (below 720p a separate pass: two events per few-us launch slow it down)
                assert dom <= (e['ms_per_step_events_pass'] or e['ms_per_step']) * 1.02, e['name']
                assert 'per_launch_frac' not in rf
            else:
                # concurrent clips: achieved / frac ARE the wall-clock figures, reproducible from the entry's own numbers
                assert abs(rf['frac'] - rf['frac_wall']) < 1e-12 and rf['per_launch_frac'] > 0
                assert abs(rf['achieved'] - rf['achieved_wall']) < 1e-9
    assert sec[0]['roofline']['bound'] == 'mfma' and sec[3]['roofline']['bound'] == 'hbm' and sec[5]['roofline']['bound'] == 'hbm'
    assert sec[2]['hip_graphs'] is True and sec[6]['vsr_x4_heads'] is True
    assert sec[3]['value'] > 0                     # fp16 operands at the headline shape
    # split fp16 (fp32-level results, three fp16 MFMAs per product): priced on the matrix pipe, executed = 3 x algorithmic
    x3 = sec[4]
    assert x3['dtype'].startswith('split f16') and x3['roofline']['bound'] == 'mfma' and x3['value'] > 0
    assert x3['roofline']['achieved'] <= 3 * x3['roofline']['algorithmic_TFLOPs'] * 1.001
    assert abs(x3['psnr'] - d['psnr_per_rank'][0]) < 1e-3      # same clip, same weights: the fp32 headline's PSNR
    assert sec[7]['dtype'].startswith('split f16') and sec[7]['roofline']['bound'] == 'mfma' and abs(sec[7]['psnr'] - sec[0]['psnr']) < 1e-3
    assert all('cpu_baseline' not in e for e in sec)          # --no-cpu-baseline covers the secondary entries too
    assert abs(ns['clips_1'] - sec[0]['value']) < 1e-4 * ns['clips_1'] and abs(ns['clips_8'] - sec[1]['value']) < 1e-4 * ns['clips_8']


def test_evaluate_refuses_a_batch_instead_of_scoring_sample_zero():
    """the reference evaluates with samples_per_gpu=1; a batch would silently drop samples from the metric"""
    from pnp_vcve_amd.registry import build_model
    from pnp_vcve_amd import synthetic as syn
    gcfg = dict(syn.DEFAULT_GENERATOR_CFG, num_blocks=1)
    model = build_model(dict(type='BasicVSR', generator=dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par',
                                                              **gcfg), pixel_loss=dict(type='CharbonnierLoss')),
                        train_cfg=None, test_cfg=dict(metrics=['PSNR', 'SSIM'], crop_border=0)).cuda().eval()
    x = torch.rand(2, 3, 3, 64, 64, device='cuda')
    with pytest.raises(ValueError):
        model.evaluate(x, x.clone())
    one = model.evaluate(x[:1], x[:1].clone())
    assert one['PSNR'] == float('inf') and abs(one['SSIM'] - 1.0) < 1e-12


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_bench_deform_basic_reports_the_dcn_roofline(precision):
    """bench.py --deform basic: the modulated-deformable aligner is timed as its own kind and priced on 2240 B per pixel"""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', '128', '--frames', '3', '--steps', '2',
                          '--warmup', '1', '--deform', 'basic', '--precision', precision, '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    d = bench_line(out.stdout)
    r = d['roofline_dcn']
    assert d['config']['deform'] == 'basic' and r['bound'] == 'hbm' and r['launches'] == 2 * 4       # 4 alignments per 3-frame clip
    assert abs(r['algorithmic_bytes_per_launch'] - 2240 * 128 * 128) < 1 and r['frac'] > 0
    assert r['kernel'] == ('dcn_window_kernel<true>' if precision == 'fp16' else 'dcn_window_kernel<false>')     # <true> = fp16 MFMA operands


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank_env(world=1, rank=0, **extra):
    env = {k: v for k, v in os.environ.items() if k not in ('PNP_DIST_BACKEND',)}
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(extra)
    return env


def test_bench_rank_path_runs_over_rccl_at_world_size_1():
    """bench.py as a rank of a torch.distributed job with the DEFAULT backend ('nccl' = RCCL): init_process_group with
    device_id -> barrier -> timed steps -> barrier -> all_reduce(MAX) of the elapsed time -> all_gather of the per-rank
    metrics (device tensors) -> barrier -> destroy_process_group, all in a fresh child process.  The same code runs at
    WORLD_SIZE=8 on the driver's node (tools/dist_test.sh:11-22, mmedit/apis/test.py:211-233)."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', '128', '--frames', '3', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, env=_rank_env())
    assert out.returncode == 0, out.stdout + out.stderr
    d = bench_line(out.stdout)
    assert d['n_gpus'] == 1 and d['dist']['process_group'] is True and d['dist']['backend'] == 'nccl'
    assert d['dist']['rccl_version'] and d['dist']['collectives'] == ['barrier', 'all_reduce(MAX)', 'all_gather', 'barrier']
    assert len(d['frames_per_s_per_rank']) == 1 and d['frames_per_s_per_rank'][0] > 0
    assert abs(d['value'] - 2 * 3 / (d['ms_per_step'] * 2e-3)) < 1e-4 * d['value']
    # without the environment the same command creates no process group
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', '128', '--frames', '3', '--steps', '1',
                          '--warmup', '1', '--no-cpu-baseline', '--no-kernel-events'], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    d = bench_line(out.stdout)
    assert d['dist'] == {'process_group': False, 'backend': None, 'rccl_version': None, 'collectives': []}


def test_gather_clip_metrics_over_rccl_at_world_size_1(tmp_path):
    """pnp_vcve_amd.dist: init_dist('pytorch', backend='nccl') + gather_clip_metrics on device tensors through a real
    RCCL all_gather (world size 1), barrier, destroy -- in a child process."""
    script = tmp_path / 'rccl_gather.py'
    script.write_text(
        "import sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from pnp_vcve_amd.dist import init_dist, gather_clip_metrics, get_dist_info, shard_indices\n"
        "init_dist('pytorch', backend='nccl')\n"
        "assert dist.get_backend() == 'nccl' and get_dist_info() == (0, 1)\n"
        "idx = shard_indices(3, 0, 1)\n"
        "local = [[30.0 + i, 0.9, 45.0] for i in idx]\n"
        "calls = []\n"
        "orig = dist.all_gather\n"
        "def spy(parts, t, *a, **k):\n"
        "    calls.append(t.device.type)\n"
        "    return orig(parts, t, *a, **k)\n"
        "dist.all_gather = spy\n"
        "tab = gather_clip_metrics(local, 3, device=torch.device('cuda', 0))\n"
        "assert calls == ['cuda'], calls\n"
        "assert tab.shape == (3, 3) and tab.dtype == torch.float64 and not tab.is_cuda\n"
        "assert tab[:, 0].tolist() == [30.0, 31.0, 32.0]\n"
        "x = torch.ones(4, device='cuda'); dist.all_reduce(x, op=dist.ReduceOp.MAX); dist.barrier()\n"
        "torch.cuda.synchronize(); dist.destroy_process_group(); print('RCCL_OK', torch.cuda.nccl.version())\n")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=_rank_env())
    assert out.returncode == 0 and 'RCCL_OK' in out.stdout, out.stdout + out.stderr


def test_dist_test_driver_one_rank_over_rccl():
    """tools/dist_test.sh CONFIG none 1 with the config's own dist_params (backend='nccl'): the reference's launcher line,
    RCCL process group, clip sharding, GPU forward, metric all_gather on device tensors; equals the non-distributed run."""
    import re
    common = ['--seed', '0', '--cfg-options', 'data.test.num_clips=3', 'data.test.num_input_frames=3',
              'data.test.height=64', 'data.test.width=64']
    cfgp = os.path.join(ROOT, 'configs', 'HR_davis_LR_128x128_IPB.py')
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'test.py'), cfgp, 'none'] + common,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    env = dict(os.environ, PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    rc = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'dist_test.sh'), cfgp, 'none', '1'] + common,
                        capture_output=True, text=True, timeout=600, env=env)
    assert rc.returncode == 0, rc.stdout + rc.stderr
    get = lambda s, k: re.search(rf'Eval-{k}: ([0-9.]+)', s).group(1)      # noqa: E731
    assert 'backend nccl' in rc.stdout
    assert get(one.stdout, 'PSNR') == get(rc.stdout, 'PSNR') and get(one.stdout, 'SSIM') == get(rc.stdout, 'SSIM')


def test_tools_test_precision_switch():
    """tools/test.py --precision f16x3 (split fp16) scores like the default fp32 run to the printed digits' last place or so; --precision
    fp16 is the --fp16 switch."""
    import re
    common = ['--seed', '0', '--cfg-options', 'data.test.num_clips=2', 'data.test.num_input_frames=3',
              'data.test.height=64', 'data.test.width=64']
    cfgp = os.path.join(ROOT, 'configs', 'HR_davis_LR_128x128_IPB.py')
    res = {}
    for flags in ((), ('--precision', 'f16x3'), ('--precision', 'fp16'), ('--fp16',)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'test.py'), cfgp, 'none'] + common + list(flags),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        res[flags] = float(re.search(r'Eval-PSNR: ([0-9.]+)', out.stdout).group(1))
    assert abs(res[()] - res[('--precision', 'f16x3')]) < 1e-4
    assert res[('--precision', 'fp16')] == res[('--fp16',)] and abs(res[()] - res[('--fp16',)]) < 5e-2


def test_bench_gpus_8_launches_eight_ranks():
    """`python bench.py --gpus 8` -- the driver's scaling run -- on the one GPU of this box: eight rank processes from the
    launcher (all on cuda:0; gloo for the tiny collectives because RCCL wants one GPU per rank), eight per-rank entries,
    whole-job arithmetic, and a failing rank fails the launcher."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['PNP_DIST_BACKEND'] = 'gloo'
    env['OMP_NUM_THREADS'] = '4'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--workload', '128', '--frames', '3',
                          '--steps', '2', '--warmup', '1', '--no-kernel-events'], capture_output=True, text=True, timeout=1500,
                         env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    d = bench_line(out.stdout)
    assert d['n_gpus'] == 8 and d['config']['parallelism'] == 'clip-sharded replicas x8' and d['scaling'] == 'weak'
    assert len(d['frames_per_s_per_rank']) == 8 and len(set(d['psnr_per_rank'])) == 8      # rank r ran clip r
    assert d['dist']['process_group'] is True and d['dist']['backend'] == 'gloo'
    assert abs(d['value'] - 8 * 2 * 3 / (d['ms_per_step'] * 2e-3)) < 1e-4 * d['value']
    assert d['cpu_baseline']['cores'] >= 1 and 'secondary' not in d
    # rc propagation: an argument the ranks reject makes every rank exit non-zero -> the launcher does too
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', '128', '--frames', '0'],
                         capture_output=True, text=True, timeout=600, env=env)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith('{')]


def test_prefetcher_uint8_upload_is_bit_identical_to_the_float_pipeline(tmp_path):
    """ClipPrefetcher takes the frames of an on-disk clip to the GPU as uint8 and applies RescaleToZeroOne + FramesToTensor
    (/255, HWC -> CHW) there: every tensor must equal what the dataset's own float pipeline (the reference's) produces, bit
    for bit, and the parallel file decode must keep the frame order."""
    from pnp_vcve_amd import synthetic as syn
    from pnp_vcve_amd.apis import ClipPrefetcher
    from pnp_vcve_amd.datasets import build_dataset
    lq, gt, qp = syn.write_clip_tree(str(tmp_path / 'data'), clips=['000', '011', '015'], t=5, h=72, w=104, seed=3)
    ds = build_dataset(dict(type='SRREDSMultipleGTCompressDataset', lq_folder=lq, gt_folder=gt, num_input_frames=100,
                            pipeline=[dict(type='LoadImageFromFileList_ipb', qp_slice_file=qp)], scale=1, val_partition='REDS4',
                            test_mode=True))
    assert len(ds) == 3
    order = [2, 0, 1]
    for i, data in zip(order, ClipPrefetcher(ds, order, 'cuda')):
        ref = ds[i]
        torch.cuda.synchronize()
        assert 'lq_u8' not in data and data['lq'].dtype == torch.float32 and data['lq'].is_contiguous()
        for k in ('lq', 'gt', 'slices', 'QPs', 'base_QPs'):
            assert data[k].shape == (1,) + tuple(ref[k].shape), k
            assert torch.equal(data[k][0].cpu(), ref[k]), (i, k)
        # the raw MV records were painted into dense maps on the device: the same maps as through the float path
        from pnp_vcve_amd.apis import _to_device
        from pnp_vcve_amd.datasets import collate
        via_float = _to_device(collate([ref]), torch.device('cuda'))
        assert 'mv_records' not in data and torch.equal(data['mvs'], via_float['mvs'])
        assert torch.equal(data['partitions'], via_float['partitions']) and torch.equal(data['lq'], via_float['lq'])
        assert data['meta'][0]['key'] == ref['meta']['key']
    # frames differ from clip to clip and from frame to frame (the equality above is not vacuous)
    assert not torch.equal(ds[0]['lq'][0], ds[0]['lq'][1]) and not torch.equal(ds[0]['lq'], ds[1]['lq'])


def test_tools_test_two_clips_in_flight_scores_exactly_like_one():
    """tools/test.py --clips-in-flight 2 (the default): pairs of clips go through the generator as one batch (two streams), the third
    of three clips alone; PSNR / SSIM must equal the strict one-clip-per-forward run digit for digit."""
    import re
    common = ['--seed', '0', '--cfg-options', 'data.test.num_clips=3', 'data.test.num_input_frames=3',
              'data.test.height=64', 'data.test.width=64']
    cfgp = os.path.join(ROOT, 'configs', 'HR_davis_LR_128x128_IPB.py')
    got = {}
    for cif in ('1', '2'):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'test.py'), cfgp, 'none', '--clips-in-flight', cif] + common,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        got[cif] = (re.search(r'Eval-PSNR: ([0-9.]+)', out.stdout).group(1), re.search(r'Eval-SSIM: ([0-9.]+)', out.stdout).group(1))
    assert got['1'] == got['2'], got


def test_multi_gpu_test_sparse_val_model_is_not_paired():
    """a sparse_val generator refuses n != 1 in eval mode; with the default clips_in_flight=2 the loop must fall back
    to clip by clip instead of aborting on the first pair of equal-shaped clips."""
    from pnp_vcve_amd import synthetic as syn
    from pnp_vcve_amd.apis import multi_gpu_test
    from pnp_vcve_amd.datasets import SyntheticCompressedClipDataset
    from pnp_vcve_amd.registry import build_model
    gcfg = dict(syn.DEFAULT_GENERATOR_CFG, num_blocks=2, sparse_val=True)
    model = build_model(dict(type='BasicVSR', generator=dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **gcfg),
                             pixel_loss=dict(type='CharbonnierLoss')), train_cfg=None,
                        test_cfg=dict(metrics=['PSNR'], crop_border=0)).cuda().eval()
    ds = SyntheticCompressedClipDataset(num_clips=4, num_input_frames=3, height=64, width=64)
    one = multi_gpu_test(model, ds, device='cuda', metrics=('PSNR',), clips_in_flight=1)
    two = multi_gpu_test(model, ds, device='cuda', metrics=('PSNR',), clips_in_flight=2)
    assert [r['eval_result'] for r in one] == [r['eval_result'] for r in two] and len(two) == 4


def test_multi_gpu_test_pairs_equal_shapes_only():
    """apis.multi_gpu_test(clips_in_flight=2) on the real model: per-clip metrics equal the one-at-a-time loop exactly, in clip order,
    for an odd number of clips; clips of different shapes fall back to one at a time."""
    from pnp_vcve_amd import synthetic as syn
    from pnp_vcve_amd.apis import multi_gpu_test, _pairable
    from pnp_vcve_amd.datasets import SyntheticCompressedClipDataset
    from pnp_vcve_amd.registry import build_model
    gcfg = dict(syn.DEFAULT_GENERATOR_CFG, num_blocks=2)
    model = build_model(dict(type='BasicVSR', generator=dict(type='IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par', **gcfg),
                             pixel_loss=dict(type='CharbonnierLoss')), train_cfg=None,
                        test_cfg=dict(metrics=['PSNR', 'SSIM'], crop_border=0)).cuda().eval()
    ds = SyntheticCompressedClipDataset(num_clips=5, num_input_frames=3, height=64, width=80)
    one = multi_gpu_test(model, ds, device='cuda', clips_in_flight=1)
    two = multi_gpu_test(model, ds, device='cuda', clips_in_flight=2)
    assert len(one) == len(two) == 5
    for a, b in zip(one, two):
        assert a['eval_result'] == b['eval_result']
    assert len({r['eval_result']['PSNR'] for r in one}) == 5          # five different clips: the order check is not vacuous
    x = {k: torch.zeros(1, 3, 3, 64, 64) for k in ('lq', 'QPs', 'slices', 'mvs', 'base_QPs', 'partitions')}
    y = dict(x, lq=torch.zeros(1, 3, 3, 64, 72))
    assert _pairable(x, x) and not _pairable(x, y) and not _pairable(x, {k: v for k, v in x.items() if k != 'mvs'})

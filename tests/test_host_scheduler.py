"""CPU: the clip scheduler (csrc/generator.hip's host C++: workspace carving, key-frame selection, expert dedup, event pool,
side streams) compiled with a plain host compiler under AddressSanitizer + UBSan and run against recording launchers
(tests/host/sched_stub.cpp; HIP replaced by csrc/host_stub/hip_stub.h).  GPU ASan is not available on this pool, so this is
where the host side of the library meets a sanitizer."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'host', 'sched_stub.cpp')


def _pattern(p, t):
    """tests/host/sched_stub.cpp::pattern"""
    s = [66.0] * t
    if p == 'IBBBP':
        s = [73.0 if i == 0 else (80.0 if i % 4 == 0 else 66.0) for i in range(t)]
    elif p == 'allP':
        s = [73.0 if i == 0 else 80.0 for i in range(t)]
    elif p == 'allB':
        s[0] = 73.0
    else:
        for i in range(min(t, len(p))):
            s[i] = float(ord(p[i]))
    return s


# name -> (n, t, contexts, slice patterns per sample, distinct routing values per sample or None = 1)
EXPECT = {
    'ibbbp_t7': (1, 7, 1, ['IBBBP']), 'allB_t7': (1, 7, 1, ['allB']), 'allP_t7': (1, 7, 1, ['allP']),
    'n2_mixed_t5': (2, 5, 1, ['IBBPB', 'IPBBB']), 't1': (1, 1, 1, ['I']), 't100': (1, 100, 1, ['IBBBP']),
    'n8_ctx8_twice': (8, 3, 8, ['IBBBP', 'allP']), 'n5_ctx3': (5, 3, 3, ['IBBBP']), 'profiled_twice': (1, 4, 1, ['IBBBP']),
    'vsr_t2': (1, 2, 1, ['IBBBP']), 'basic_t3': (1, 3, 1, ['IBBBP']), 'nocat_noalign_t4': (1, 4, 1, ['IBPB']),
    'channel_last_two_layer_t3': (1, 3, 1, ['IBBBP']), 'sparse_val_t3': (1, 3, 1, ['IBBBP']), 'qp_routed_t6': (1, 6, 1, ['IBBBP']),
    'p720_t2': (1, 2, 1, ['IBBBP']),
}


@pytest.fixture(scope='module')
def stub_run(tmp_path_factory):
    cxx = shutil.which('g++') or shutil.which('clang++') or '/opt/rocm/lib/llvm/bin/clang++'
    exe = str(tmp_path_factory.mktemp('sched') / 'sched_stub')
    cmd = [cxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-DPNP_HOST_STUB',
           '-Wno-attributes', '-x', 'c++', SRC, '-o', exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    docs = {}
    for ln in r.stdout.splitlines():
        if ln.startswith('{'):
            d = json.loads(ln)
            docs[d['name']] = d
    return exe, r, docs, env


def test_scheduler_is_clean_under_asan_and_ubsan(stub_run):
    exe, r, docs, env = stub_run
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr and 'LeakSanitizer' not in r.stderr, r.stderr[-4000:]
    assert len(docs) == 3 * len(EXPECT) + 4 + 6
    for name, d in docs.items():
        assert d['pack_rc'] == 0 and d['forward_rc'] == 0 and d['errors'] == [], (name, d['errors'])
        # no event or stream outlives its generator; the workspace handed over is exactly contexts x the advertised size
        assert d['live_events_after_destroy'] == 0 and d['live_streams_after_destroy'] == 0, name
        base = name.split('_', 1)[1]
        for pre in ('nomirrors_', 'chainmirrors_', 'wino1_', 'wino2_'):
            base = base[len(pre):] if base.startswith(pre) else base
        assert d['workspace_bytes'] == EXPECT[base][2] * d['context_bytes'] and d['context_bytes'] % 256 == 0


def test_the_harness_catches_a_workspace_that_is_too_small(stub_run):
    """self-test: 256 bytes less workspace than advertised must be reported (the last carved region runs into the red zone)"""
    exe, _, _, env = stub_run
    r = subprocess.run([exe, 'f32_ibbbp_t7', 'f16_ibbbp_t7'], capture_output=True, text=True, timeout=300,
                       env=dict(env, PNP_STUB_SHRINK_WS='256'))
    docs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode != 0 and len(docs) == 2
    for d in docs:
        assert any('leaves its buffer' in e for e in d['errors']), d['errors']


@pytest.mark.parametrize('prec', ['f32', 'f16', 'x3'])
def test_key_frame_selection_matches_the_reference_rule(stub_run, prec):
    """iconvsr_ipb_par.py:60-62,81,116: key = slice in {I, P}, both ends forced; the backward sweep aligns the nearest key frame
    AFTER frame i, the forward sweep the nearest one BEFORE it -- read off the recorded warp launches (source slot, flow plane)."""
    _, _, docs, _ = stub_run
    for base, (n, t, ctx, pats) in EXPECT.items():
        d = docs[f'{prec}_{base}']
        exp_frame, exp_plane, exp_key, exp_ctx = [], [], [], []
        for b in range(n):
            sl = _pattern(pats[b % len(pats)], t)
            key = [v in (73.0, 80.0) for v in sl]
            key[0] = key[-1] = True
            for i in range(t - 2, -1, -1):
                exp_frame.append(i)
                exp_plane.append(2)
                exp_key.append(next(k for k in range(i + 1, t) if key[k]))
                exp_ctx.append(b % ctx)
            for i in range(1, t):
                exp_frame.append(i)
                exp_plane.append(0)
                exp_key.append(next(k for k in range(i - 1, -1, -1) if key[k]))
                exp_ctx.append(b % ctx)
        reps = 2 if base.endswith('twice') else 1      # the records of the LAST forward are reported
        assert reps and d['warp_frame'] == exp_frame and d['warp_flow_plane'] == exp_plane, base
        assert d['warp_key_slot'] == exp_key and d['warp_context'] == exp_ctx, (base, d['warp_key_slot'], exp_key)
        assert d['dcn_calls'] == (len(exp_key) if base == 'basic_t3' else 0)


@pytest.mark.parametrize('prec', ['f32', 'f16', 'x3'])
def test_expert_mixtures_are_made_once_per_distinct_routing_value(stub_run, prec):
    """Dynamic_conv2d_se's mm(attention, weight) (sr_backbone_utils.py:198-202) is hoisted to once per distinct routing input of
    a clip: base_QP is constant over a clip (1 mixture per sample), QP routing (use_base_qp=False) varies per frame; every
    partition-branch conv of frame i must use the mixture made for its routing value."""
    _, _, docs, _ = stub_run
    for base, (n, t, ctx, pats) in EXPECT.items():
        d = docs[f'{prec}_{base}']
        if base == 'qp_routed_t6':
            qps = [(20 + (i * 7) % 20) for i in range(t)]
            first = {}
            for i, q in enumerate(qps):
                first.setdefault(q, len(first))
            assert d['mix_slot'] == list(range(len(first)))
            assert all(m == first[qps[f]] for f, m in zip(d['par_conv_frame'], d['par_conv_mixture']))
        else:
            assert d['mix_slot'] == [0] * n, (base, d['mix_slot'])
            assert set(d['par_conv_mixture']) == {0}
        nb = 2 if base.startswith(('vsr', 'channel_last')) is False else 2
        assert nb and len(d['par_conv_frame']) > 0 and set(d['par_conv_frame']) == set(range(t))


def test_event_pool_and_side_streams_are_reused(stub_run):
    _, _, docs, _ = stub_run
    for prec in ('f32', 'f16', 'x3'):
        d = docs[f'{prec}_profiled_twice']
        assert d['event_pool_after_forward'][0] > 0 and d['event_pool_after_forward'][0] == d['event_pool_after_forward'][1]
        assert docs[f'{prec}_ibbbp_t7']['event_pool_after_forward'] == [0]           # no events without profiling
        d = docs[f'{prec}_n8_ctx8_twice']
        assert d['streams_created_after_forward'] == [8, 8]                          # the second forward creates none
        side = sorted(set(d['launch_stream']))
        assert len(side) == 8 and 0 not in side                                      # 8 samples on 8 side streams, none on the caller's
        # fork: every side stream waits for an event recorded on the caller's stream before its first launch;
        # join: the caller's stream waits for one event per side stream
        fork = [(s, on) for s, on in zip(d['wait_stream'], d['wait_event_recorded_on']) if s != 0]
        join = [(s, on) for s, on in zip(d['wait_stream'], d['wait_event_recorded_on']) if s == 0]
        assert sorted(s for s, _ in fork) == side and all(on == 0 for _, on in fork)
        assert sorted(on for _, on in join) == side
        d = docs[f'{prec}_n5_ctx3']
        assert d['streams_created_after_forward'] == [3] and len(set(d['launch_stream']) - {0}) == 3     # stream 0: pnp_generator_pack


def test_fp16_mirrors_are_scheduled_consistently(stub_run):
    """PNP_OPT_F16_MIRRORS: the warp writes fp16, every input conv with a 64-channel source reads ALL of them as fp16 maps in one
    launch and writes the mirror of its output; the stub's written-range bookkeeping (errors == []) already proved that no mirror
    is read before its producer ran -- here the masks show the mirrors are really in use, and absent with the option off."""
    _, _, docs, _ = stub_run
    d, off, chain = docs['f16_ibbbp_t7'], docs['f16_nomirrors_ibbbp_t7'], docs['f16_chainmirrors_ibbbp_t7']
    assert set(d['warp_f16']) == {1} and set(chain['warp_f16']) == {1} and set(off['warp_f16']) == {0}
    for doc, in_branch in ((d, False), (chain, True)):
        multi = [(ns, m) for ns, m, f in zip(doc['conv_nsrc'], doc['conv_map_mask'], doc['conv_f16_path']) if ns > 1]
        assert multi and all((m & 0xF) == sum(1 << s for s in range(1, ns)) for ns, m in multi)    # every wide source through its mirror
        assert all(bool(m & 32) == in_branch for _, m in multi)          # the input conv mirrors its output only for the in-branch chain
        single = [m for ns, m in zip(doc['conv_nsrc'], doc['conv_map_mask']) if ns == 1]
        # default: one fp32 + mirror output per branch and frame (its last block: the frame slot); chain: every back half
        assert sum(1 for m in single if m & 32) == (2 * 7 * 8 if in_branch else 2 * 7)
    assert not any(m & (32 | 0xE) for m in off['conv_map_mask'])
    assert d['launches_first_forward'] == off['launches_first_forward'] == chain['launches_first_forward']     # same launches, different maps
    for nm in ('f16_chainmirrors_channel_last_two_layer_t3', 'f16_chainmirrors_p720_t2'):
        assert docs[nm]['errors'] == [] and docs[nm]['forward_rc'] == 0
    assert set(docs['f32_ibbbp_t7']['conv_map_mask']) == {0} and set(docs['f32_ibbbp_t7']['conv_f16_path']) == {0}
    assert set(docs['f16_basic_t3']['warp_f16']) == {0}                              # DCN aligners keep the r02 schedule


def test_winograd_option_routes_the_single_source_64_channel_convs(stub_run):
    """PNP_OPT_WINOGRAD (fp32 only): both halves of every BAE block and conv_hr carry a Winograd image and take conv_wino.hip; input
    convs and the RGB / pixel-shuffle heads never do; option 1 = quadrant units up to 128 16x16 tiles, the tile kernels above, option 2 = tile kernels everywhere; the stub's
    range bookkeeping (errors == []) proved that every image -- the per-frame ones of the expert-mixed convs included -- was written
    before it was read, in the workspace the scheduler advertised."""
    _, _, docs, _ = stub_run
    d, ref = docs['f32_wino2_ibbbp_t7'], docs['f32_ibbbp_t7']
    assert d['errors'] == [] and d['conv_nsrc'] == ref['conv_nsrc'] and set(ref['conv_wino']) == {0}
    # t = 7, 8 blocks: 2 sweeps x 7 frames x 16 block halves + 7 conv_hr
    assert sum(d['conv_wino']) == 2 * 7 * 16 + 7
    assert all(w == 0 for w, ns in zip(d['conv_wino'], d['conv_nsrc']) if ns > 1)
    # the input convs with at least one 64-channel source take the multi-source form: every frame of both sweeps but the last
    # frame's backward one (the frame alone)
    assert sum(d['conv_wino_ms']) == 2 * 7 - 1 and all(ns > 1 for m, ns in zip(d['conv_wino_ms'], d['conv_nsrc']) if m)
    assert set(ref['conv_wino_ms']) == {0} and set(docs['f16_wino2_ibbbp_t7']['conv_wino_ms']) == {0}
    # one launch per branch and frame makes the 8 images of the expert-mixed convs, one per clip the per-frame "any partition record"
    # words (the I frames' front halves are gated on them inside launch_conv3x3_wino): 15 more launches than the direct schedule
    assert d['launches_first_forward'] == ref['launches_first_forward'] + 15
    # auto mode (the default): a small frame (24 tiles) takes the quadrant-unit kernels on the same convs; 720p takes the tile kernels,
    # units nowhere
    a = docs['f32_wino1_ibbbp_t7']
    assert a['errors'] == [] and a['conv_wino'] == d['conv_wino'] and a['conv_wino_ms'] == d['conv_wino_ms']
    assert a['conv_wino_units'] == [x | y for x, y in zip(d['conv_wino'], d['conv_wino_ms'])]      # both kinds as quadrant units
    assert set(d['conv_wino_units']) == {0} and set(docs['f32_wino1_p720_t2']['conv_wino_units']) == {0}
    p = docs['f32_wino1_p720_t2']
    assert p['errors'] == [] and sum(p['conv_wino']) == 2 * 2 * 16 + 2
    assert set(docs['f16_wino2_ibbbp_t7']['conv_wino']) == {0}                   # the option is an fp32-path switch
    c = docs['f32_wino2_channel_last_two_layer_t3']                             # both convs of a block expert-mixed, branches + residual in one launch
    assert c['errors'] == [] and sum(c['conv_wino']) == 2 * 3 * 16 + 3
    v = docs['f32_wino2_vsr_t2']                                                  # x4 head: conv_hr at 4h x 4w takes it, the pixel-shuffle convs do not
    assert v['errors'] == [] and sum(v['conv_wino']) == 2 * 2 * 16 + 2


def test_split_fp16_schedule_uses_the_split_kernel_where_it_applies(stub_run):
    """PNP_PREC_F16X3: every NHWC64 conv with a 64-channel source (input convs, both halves of every BAE block, conv_hr) is handed
    to the split kernel with both weight images (the stub checked hi and lo were written before being read); the RGB-only input
    conv, the pixel-shuffle / RGB heads and the DCN offset convs stay on the fp32 kernels; no fp16 maps anywhere."""
    _, _, docs, _ = stub_run
    for name in ('x3_ibbbp_t7', 'x3_vsr_t2', 'x3_basic_t3', 'x3_p720_t2'):
        d, f = docs[name], docs['f32_' + name[3:]]
        assert d['errors'] == [] and set(d['conv_map_mask']) == {0} and set(d['warp_f16']) == {0}
        assert d['conv_nsrc'] == f['conv_nsrc']                      # the same convs in the same order as the fp32 schedule
        assert set(d['conv_f16_path']) == {0, 2} and set(f['conv_f16_path']) == {0}
        # the convs with partition branches (BAE front halves) all run split
        assert d['par_conv_frame'] == f['par_conv_frame']
    d = docs['x3_ibbbp_t7']
    n_split = sum(1 for p in d['conv_f16_path'] if p == 2)
    # t = 7, 8 blocks: per frame and sweep 1 input conv + 16 block halves, except the last frame's backward input conv (RGB only);
    # + conv_hr per frame; conv_last is an RGB head
    assert n_split == 2 * 7 * 17 - 1 + 7 and len(d['conv_f16_path']) - n_split == 1 + 7

"""What every test of bench.py's stdout holds it to: ONE strict-JSON line, short enough for the driver's 8 KB stdout tail."""
import json

MAX_LINE_BYTES = 4096


def _no_constants(name):
    raise ValueError(f'non-strict JSON constant {name!r} in the bench line')


def bench_line(stdout, clean=True):
    """the single result line of a bench.py run, parsed strictly (NaN / Infinity are refused).  It must be the LAST non-blank
    line of stdout (the driver reads the tail); with `clean` nothing else may be on stdout at all (gloo prints its own
    "[Gloo] Rank ..." connection lines there, so the multi-rank gloo tests pass clean=False)."""
    nonblank = [ln for ln in stdout.splitlines() if ln.strip()]
    lines = [ln for ln in nonblank if ln.startswith('{')]
    assert len(lines) == 1 and nonblank[-1] == lines[0], f'stdout must end with the one JSON line:\n{stdout[-2000:]}'
    if clean:
        assert len(nonblank) == 1, f'stdout must hold nothing but the JSON line:\n{stdout[-2000:]}'
    else:
        assert all(ln == lines[0] or 'Gloo' in ln or 'peer ranks' in ln for ln in nonblank), stdout[-2000:]
    assert len(lines[0].encode()) <= MAX_LINE_BYTES, f'bench line is {len(lines[0])} bytes'
    d = json.loads(lines[0], parse_constant=_no_constants)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config'):
        assert k in d, k
    assert 'secondary' not in d
    return d

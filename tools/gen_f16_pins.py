#!/usr/bin/env python
"""Writes the SHA-256 pins of tests/f16_pin_cases.py with the package found under --root (default: this tree).

    python tools/gen_f16_pins.py --root .r02ref --out gpurun_out/f16_pins_r02.json      (on an MI355X)

.r02ref is `git archive 340801c` (the round-2 tree) built in place; the result is committed as
tests/golden/f16_pins_r02.json and checked by tests/test_gpu_fp16.py::test_f16_path_is_bit_identical_to_the_round_2_build.
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--root', default=HERE)
ap.add_argument('--out', required=True)
args = ap.parse_args()
root = os.path.abspath(args.root)
sys.path.insert(0, os.path.join(HERE, 'tests'))
sys.path.insert(0, root)                      # the package under test wins
import torch  # noqa: E402

import pnp_vcve_amd  # noqa: E402
from pnp_vcve_amd import synthetic  # noqa: E402
from pnp_vcve_amd.registry import build_backbone  # noqa: E402
import f16_pin_cases as pc  # noqa: E402

assert os.path.abspath(pnp_vcve_amd.__file__).startswith(root), (pnp_vcve_amd.__file__, root)
res = {'package': os.path.relpath(os.path.dirname(pnp_vcve_amd.__file__), HERE), 'device': torch.cuda.get_device_name(0), 'cases': {}}
for case in pc.PIN_CASES:
    out = pc.run_case(case, synthetic, build_backbone, torch)
    again = pc.run_case(case, synthetic, build_backbone, torch)
    d = pc.digest(out)
    assert d['sha256'] == pc.digest(again)['sha256'], case['name']      # run-to-run deterministic in the first place
    res['cases'][case['name']] = d
    print(case['name'], d['sha256'][:16], d['shape'], flush=True)
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
with open(args.out, 'w') as f:
    json.dump(res, f, indent=1)

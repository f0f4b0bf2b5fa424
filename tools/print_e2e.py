#!/usr/bin/env python
"""print the end-to-end entry of bench.py's side file:  python tools/print_e2e.py bench_secondary.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
e = d['secondary'][-1]
print({k: v for k, v in e.items() if k not in ('name', 'note')})

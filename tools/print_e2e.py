#!/usr/bin/env python
"""print the end-to-end secondary entry of a bench.py JSON line:  python tools/print_e2e.py bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d['secondary'][-1]
print({k: v for k, v in e.items() if k not in ('name', 'note')})

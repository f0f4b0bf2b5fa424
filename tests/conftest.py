import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    # the MI355X hosts have 256 logical CPUs: oneDNN on the oracle's 64-channel convs is fastest at ~16
    # threads and 10x slower when oversubscribed (see bench.py cpu_baseline calibration)
    import os
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))

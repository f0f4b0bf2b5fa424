#!/usr/bin/env python
"""Command line of pnp_vcve_amd/isa_hazards.py: disassembles the gfx950 code objects of a library (or reads an assembly listing) and
reports every store of more than 64 bits whose data registers the next vector-ALU instruction writes -- the hazard LLVM leaves
unpadded for the SGPR-soffset store form (see that module's header; build_native.py pads it in the listing, so a built library is clean).

    python tools/lint_store_hazard.py [library.so | object.o | listing.s ...]       (default: pnp_vcve_amd/lib/libpnpvcve_hip.so)

exits 1 and prints every site when one is found."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_vcve_amd.isa_hazards import lint_library  # noqa: E402

if __name__ == '__main__':
    paths = sys.argv[1:] or [os.path.join(ROOT, 'pnp_vcve_amd', 'lib', 'libpnpvcve_hip.so')]
    bad = 0
    for p in paths:
        found, n = lint_library(p)
        print(f'{p}: {n} code object(s), {len(found)} store(s) of more than 64 bits with a vector-ALU write of their data registers right behind')
        for kernel, no, st, nx, where, _ in found:
            print(f'   {where} {kernel} (line {no}):\n      {st}\n      {nx}')
        bad += len(found)
    sys.exit(1 if bad else 0)

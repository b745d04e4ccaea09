#!/usr/bin/env python3
"""The GPU fuzz mix (tests/fuzz_cases.py) under the SIMT emulator, no GPU: kernel LOGIC of the chosen kinds on random inputs.
   Usage: python tools/fuzz_emu.py <seconds> <first seed> [kinds, e.g. 0,2,6]   (kinds: see fuzz_cases.KINDS; SpGEMM = 0, 1, 2, 6)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fuzz_cases as fz
from emu import emu_backend
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kinds = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
n_ok, per_kind, last = fz.run(emu_backend.backend(), budget, seed0, kinds=kinds)
print("fuzz (emulator): %d cases in %.0f s (seeds %d..%d); kinds %s; per kind %s" % (n_ok, budget, seed0, last, kinds, per_kind), flush=True)

#!/usr/bin/env python3
"""Randomised parity sweep on the GPU against the CPU oracle, run by hand through gpurun for long budgets
(`python tools/fuzz_gpu.py SECONDS [SEED0]`); the cases live in tests/fuzz_cases.py, of which the driver-run GPU suite takes a
60-second slice (tests/test_gpu_parity.py::test_fuzz_slice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases
import parity_cases as pc
be = pc.kk.torch_backend()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_ok, per_kind, last = fuzz_cases.run(be, budget, seed0)
print("fuzz: %d cases passed in %.0f s (seeds %d..%d); per kind (%s): %s" % (n_ok, budget, seed0, last, ", ".join(fuzz_cases.KINDS), per_kind))

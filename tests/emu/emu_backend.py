"""TEST INFRASTRUCTURE: a numpy Backend bound to tests/emu/libkkamd_emu.so -- the same kernel sources
as the product, compiled by g++ against the SIMT emulator (kk_emu.h).  Lets `-m "not gpu"` tests check
kernel logic (tiling, carries, hash tables, scans, sorts) on a machine without a GPU.  It is NOT a CPU
fallback of the product: nothing under kokkos-kernels_amd/ can load it."""
import ctypes as C
import os
import subprocess

import numpy as np

import kk_loader

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libkkamd_emu.so")
_BACKEND = None


def build():
    subprocess.check_call(["make", "-C", HERE, "-s"])
    return SO


def backend():
    global _BACKEND
    if _BACKEND is None:
        kk = kk_loader.load()
        build()
        lib = kk._capi.bind(C.CDLL(SO))
        _BACKEND = kk.Backend(lib, lambda n, dt: np.zeros(int(n), dtype=dt),
                              lambda a: None if a is None else a.ctypes.data, lambda: None,
                              lambda a: a, lambda a: np.array(a, copy=True), "emu")
    return _BACKEND

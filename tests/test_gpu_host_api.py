"""Runs the C++ drop-in header tests (kokkos-kernels_amd/host/tests/test_drop_in.cpp) on the GPU: the public
KokkosSparse::spmv / spgemm_* / SPMVHandle / KokkosKernelsHandle surface over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kokkos-kernels_amd", "host", "tests", "test_drop_in")


@pytest.mark.gpu
def test_cpp_drop_in_headers():
    assert os.path.exists(EXE), "build it with __graft_entry__.build() (make -C kokkos-kernels_amd/host)"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout


def test_cpp_drop_in_headers_build():
    """CPU-side: the headers compile and link against libkkamd.so (no GPU needed to build)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "kokkos-kernels_amd", "host"), "-s"])
    assert os.path.exists(EXE)

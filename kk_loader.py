"""Imports the package directory `kokkos-kernels_amd/` (the hyphen is the repo's mandated name and is not
a valid Python identifier) under the module name `kokkos_kernels_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "kokkos-kernels_amd")


def load():
    if "kokkos_kernels_amd" in sys.modules:
        return sys.modules["kokkos_kernels_amd"]
    spec = importlib.util.spec_from_file_location("kokkos_kernels_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["kokkos_kernels_amd"] = mod
    spec.loader.exec_module(mod)
    return mod

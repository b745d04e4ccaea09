"""The reference-side binding (kokkos-kernels_amd/host/kokkos_tpl/*.append.hpp) compiled against the
REFERENCE's own headers: the integration step is performed on copies of the reference's sparse/tpls files in a temporary
directory, and `g++ -fsyntax-only` parses the reference's sparse/impl/*_spec.hpp with them over a declarations-only stand-in
for Kokkos core (tests/kokkos_mock/).  The translation units assert that every claimed type tuple -- built the way the public
KokkosSparse::spmv / spgemm_* build their internal types -- selects the KKAMD specialisation.
Needs /root/reference (present in the build container, absent on the GPU box): skipped without it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
TPL = os.path.join(ROOT, "kokkos-kernels_amd", "host", "kokkos_tpl")
MOCK = os.path.join(ROOT, "tests", "kokkos_mock")
GENERATED = ["KokkosSparse_spmv_eti_spec_avail.hpp", "KokkosSparse_spmv_mv_eti_spec_avail.hpp", "KokkosSparse_spmv_eti_spec_decl.hpp",
             "KokkosSparse_spmv_mv_eti_spec_decl.hpp", "KokkosSparse_spgemm_symbolic_eti_spec_avail.hpp",
             "KokkosSparse_spgemm_symbolic_eti_spec_decl.hpp", "KokkosSparse_spgemm_numeric_eti_spec_avail.hpp",
             "KokkosSparse_spgemm_numeric_eti_spec_decl.hpp"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sparse", "tpls")), reason="the reference checkout is not on this machine")


def integrate(tmp):
    """INTEGRATION.md section 2 on copies: every <name>.append.hpp is appended to the reference's sparse/tpls/<name>.hpp, the
    patch adds the sub-handle member to SPGEMMHandle; cmake's generated (empty: no ETI in this check) headers are created."""
    tpls = os.path.join(tmp, "tpls"); os.makedirs(tpls, exist_ok=True)
    for f in sorted(os.listdir(TPL)):
        if f.endswith(".append.hpp"):
            name = f[:-len(".append.hpp")] + ".hpp"
            with open(os.path.join(tpls, name), "w") as out:
                out.write(open(os.path.join(REF, "sparse", "tpls", name)).read())
                out.write("\n" + open(os.path.join(TPL, f)).read())
    gen = os.path.join(tmp, "generated_specializations_hpp"); os.makedirs(gen, exist_ok=True)
    for g in GENERATED:
        open(os.path.join(gen, g), "w").close()
    for f in os.listdir(TPL):
        if f.endswith(".patch"):
            target = f[:-len(".patch")]
            src = os.path.join(tmp, "src"); os.makedirs(src, exist_ok=True)
            shutil.copy(os.path.join(REF, "sparse", "src", target), os.path.join(src, target))
            subprocess.check_call(["patch", "-s", os.path.join(src, target), os.path.join(TPL, f)])
            # quote-includes look next to the including file first: the reference headers that include the patched one move along
            for includer in ("KokkosKernels_Handle.hpp",):
                shutil.copy(os.path.join(REF, "sparse", "src", includer), os.path.join(src, includer))
    return tmp


def syntax_check(tmp, tu, extra=()):
    inc = [tmp, os.path.join(tmp, "tpls"), os.path.join(tmp, "src"), MOCK, os.path.join(ROOT, "include")] + \
          [os.path.join(REF, d) for d in ("sparse/src", "sparse/impl", "sparse/tpls", "common/src", "common/impl", "graph/src", "graph/impl",
                                          "blas/src", "blas/impl", "blas/tpls", "batched", "batched/dense/src", "batched/dense/impl")]
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", "-fmax-errors=20", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + list(extra) + ["-I" + i for i in inc] + [os.path.join(MOCK, tu)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-6000:]


def test_spmv_specialisations_against_reference_headers(tmp_path):
    syntax_check(integrate(str(tmp_path)), "check_spmv_binding.cpp")


def test_spmv_specialisations_coexist_with_rocsparse_guard(tmp_path):
    """with KOKKOSKERNELS_ENABLE_TPL_ROCSPARSE defined the int-offset tuples (rocSPARSE claims them) are left alone: the
    reference's rocSPARSE wrappers (parsed against the real rocsparse.h of this ROCm) and the KKAMD ones live in one translation
    unit without a redefinition, and the 64-bit-offset tuples stay bound to KKAMD"""
    syntax_check(integrate(str(tmp_path)), "check_spmv_binding.cpp", extra=["-DKOKKOSKERNELS_ENABLE_TPL_ROCSPARSE"])


def test_spgemm_specialisations_against_reference_headers(tmp_path):
    """SPGEMM_SYMBOLIC / SPGEMM_NUMERIC over the reference's real KokkosKernelsHandle and (patched) SPGEMMHandle"""
    syntax_check(integrate(str(tmp_path)), "check_spgemm_binding.cpp")


def test_spgemm_specialisations_coexist_with_rocsparse(tmp_path):
    syntax_check(integrate(str(tmp_path)), "check_spgemm_binding.cpp", extra=["-DKOKKOSKERNELS_ENABLE_TPL_ROCSPARSE"])


def test_append_files_are_inert_without_the_tpl_macro(tmp_path):
    """with KOKKOSKERNELS_ENABLE_TPL_KKAMD undefined the appended text must vanish: the unmodified build is untouched"""
    tmp = integrate(str(tmp_path))
    tu = os.path.join(tmp, "inert.cpp")
    with open(tu, "w") as f:
        f.write("#include <KokkosSparse_spmv_spec.hpp>\n#include <KokkosSparse_spgemm_symbolic_spec.hpp>\n#include <KokkosSparse_spgemm_numeric_spec.hpp>\n"
                "#ifdef KKAMD_OK\n#error kkamd.h was included although the TPL is off\n#endif\nint main() { return 0; }\n")
    cfg = os.path.join(tmp, "KokkosKernels_config.h")          # found first: the same configuration without the KKAMD TPL
    with open(cfg, "w") as f:
        f.write(open(os.path.join(MOCK, "KokkosKernels_config.h")).read().replace("#define KOKKOSKERNELS_ENABLE_TPL_KKAMD", ""))
    shutil.copy(tu, os.path.join(MOCK, "_inert_tmp.cpp"))
    try:
        syntax_check(tmp, "_inert_tmp.cpp")
    finally:
        os.remove(os.path.join(MOCK, "_inert_tmp.cpp"))

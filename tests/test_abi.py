"""The C-ABI library loads and exports every symbol include/thermite.h declares
(no compute calls: there is no GPU in the CPU test run)."""
import ctypes
import os
import re

import pytest

from thermite_amd import capi


def test_library_exports_every_declared_symbol():  # also compiles the C++ mirror header
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "thermite.h")).read()
    declared = set(re.findall(r"\b(thm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.ABI_SYMBOLS)
    L = ctypes.CDLL(capi.SO_PATH)
    for s in sorted(declared):
        assert hasattr(L, s), "missing export: " + s


def test_struct_layouts_match_header():
    assert capi.ALN_DT.itemsize == 112
    assert ctypes.sizeof(capi.Opts) == 32
    from oracle import pyoracle as orc
    assert orc.ALN_DT == capi.ALN_DT and orc.MEM_DT == capi.MEM_DT and orc.SWG_DT == capi.SWG_DT


def test_no_cpu_fallback(data_dir):
    """Without a GPU the product path must fail loudly, never fall back."""
    from thermite_amd import refdata
    if capi.lib().thm_device_count() > 0:
        pytest.skip("a GPU is visible")
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    ix = capi.Index(t)
    with pytest.raises(capi.ThermiteError) as e:
        capi.Aligner(ix, capi.DEFAULT_OPTS)
    assert e.value.code == capi.ERR_NO_DEVICE


def test_cpp_mirror_header_compiles(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text('#include "thermite.hpp"\nint main() { thermite::AlignOpts o; return (int)o.min_seed_len - 20; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(root, "include"), "-fsyntax-only", str(src)])


def test_c_headers_are_plain_c(tmp_path):
    """the boundary is a C ABI: both headers must compile as C99, not only as C++"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "thermite_io.h"\nint main(void) { thm_run_stats s; thm_aln a; (void)s; (void)a; return THM_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                           "-fsyntax-only", str(src)])


def test_tools_and_bench_compile():
    """the scripts under tools/ run only on the GPU box: at least keep them syntactically alive here"""
    import glob
    import py_compile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]:
        py_compile.compile(f, doraise=True)

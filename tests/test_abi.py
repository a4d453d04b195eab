"""CPU tier: the C-ABI library loads and exports exactly what include/s2k.h declares; host-only entry
points behave; nothing computes without a GPU (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from s2k_loader import import_package

pkg = import_package()


def _header_functions():
    src = open(os.path.join(ROOT, "include", "s2k.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(s2k_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = pkg.load_library()
    declared = _header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), "include/s2k.h declares %s but libs2k.so does not export it" % name
    assert sorted(pkg.ABI_SYMBOLS) == declared
    assert lib.s2k_abi_version() == pkg.ABI_VERSION == 2


def test_hash_bound_host_function(oracle):
    for d in (0.01, 0.001, 1e-4, 0.1, 0.5, 1.0, 3.0, 0.0, -2.0, float("nan"), 0.2, 1e-9, 0.999999):
        assert pkg.hash_bound(d) == oracle.hash_bound(d)


def test_strerror():
    lib = pkg.load_library()
    assert lib.s2k_strerror(0) == b"ok"
    assert b"no GPU" in lib.s2k_strerror(8)


def test_no_cpu_fallback_without_gpu():
    lib = pkg.load_library()
    if lib.s2k_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.S2kError) as e:
        pkg.Engine(0)
    assert e.value.status == 8  # S2K_ERR_NO_DEVICE


def test_struct_layouts_match_header(tmp_path):
    """A C translation unit that includes include/s2k.h is compiled and run; its sizeof/offsetof of every struct field
    must equal the ctypes mirrors the tests and bench.py call the library through (and the Rust #[repr(C)] structs of
    INTEGRATION.md are written against the same table)."""
    import subprocess

    exe = str(tmp_path / "abi_layout")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "abi_layout.c"), "-o", exe])
    facts = {}
    for line in subprocess.check_output([exe], text=True).splitlines():
        k, *v = line.split()
        facts[k] = tuple(int(x) for x in v)
    mirrors = {"s2k_params": pkg.Params, "s2k_counts": pkg.Counts, "s2k_result": pkg.Result, "s2k_device_out": pkg.DeviceOut}
    for cname, T in mirrors.items():
        assert facts["sizeof." + cname] == (ctypes.sizeof(T),), cname
        declared = [k.split(".")[1] for k in facts if k.startswith(cname + ".")]
        assert declared == [f for f, _ in T._fields_], (cname, declared)  # same fields, same order
        for f, _ in T._fields_:
            d = getattr(T, f)
            assert facts["%s.%s" % (cname, f)] == (d.offset, d.size), (cname, f)
    HM = pkg.HashMode
    assert (facts["enum.S2K_MODE_REGULAR"], facts["enum.S2K_MODE_HPC"], facts["enum.S2K_MODE_SIMD"], facts["enum.S2K_MODE_HPCSIMD"]) == \
        ((int(HM.Regular),), (int(HM.Hpc),), (int(HM.Simd),), (int(HM.HpcSimd),))
    assert facts["enum.S2K_FLAG_WANT_MINIMIZERS"] == (pkg.FLAG_WANT_MINIMIZERS,) and facts["enum.S2K_FLAG_FORCE_SERIAL"] == (pkg.FLAG_FORCE_SERIAL,)
    assert facts["enum.S2K_HPC_RLE_ALPHABET"] == (pkg.HPC_RLE_ALPHABET,) and facts["enum.S2K_FLAG_NO_PACK2"] == (pkg.FLAG_NO_PACK2,)
    assert facts["enum.S2K_ABI_VERSION"] == (pkg.load_library().s2k_abi_version(),)
    assert facts["enum.S2K_ERR_NO_DEVICE"] == (8,) and facts["enum.S2K_ERR_CAPACITY"] == (7,)
    assert facts["enum.S2K_FLAG_LEGACY_PATH"] == (pkg.FLAG_LEGACY_PATH,)


def test_kminmerhash_semantics():
    a = pkg.KminmerHash.new_from_hash(5, 1, 2, 0, False)
    b = pkg.KminmerHash(5, 9, 9, 3, True)
    assert a == b and not (a < b)  # equality by hash only, src/kminmer.rs:181-185
    assert a.get_hash() == 5
    assert "rev: false" in repr(a)

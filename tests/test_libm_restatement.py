"""The product's restatement of glibc expf/logf/log (ulc-codec_amd/csrc/ulcx_libm.h),
compiled for the host, against the live libm of the machine running the tests.
(Full 2^32 sweep: run cmp_expf/cmp_logf with stride 1 — 0 mismatches on glibc 2.35.)"""
import ctypes as C
import os
import subprocess
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "helpers", "libm_check.cpp")
    so = os.path.join(HERE, "helpers", "libm_check.so")
    hdr = os.path.join(HERE, "..", "ulc-codec_amd", "csrc", "ulcx_libm.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src, "-lm"])
    l = C.CDLL(so)
    for f in (l.cmp_expf, l.cmp_logf):
        f.restype = C.c_longlong
        f.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    l.cmp_log.restype = C.c_longlong
    l.cmp_log.argtypes = [C.c_uint64, C.c_longlong, C.POINTER(C.c_uint64)]
    return l


@pytest.mark.parametrize("fn", ["cmp_expf", "cmp_logf"])
def test_f32_functions_bit_exact_on_strided_sweep(lib, fn):
    bad = C.c_uint32(0)
    # every 61st bit pattern of the whole binary32 space (~70 M points incl. NaN/inf/subnormals)
    n = getattr(lib, fn)(0, 1 << 32, 61, C.byref(bad))
    assert n == 0, f"{n} mismatches, first at bit pattern {bad.value:#010x}"


def test_f32_functions_dense_near_hot_ranges(lib):
    bad = C.c_uint32(0)
    # expf arguments in the codec are mostly in [-60, 5]; logf arguments are positive ratios
    assert lib.cmp_expf(0xC0000000, 0xC2800000, 1, C.byref(bad)) == 0   # [-2, -64]
    assert lib.cmp_logf(0x3F000000, 0x40000000, 1, C.byref(bad)) == 0   # [0.5, 2)


def test_f64_log_bit_exact_on_random_sample(lib):
    bad = C.c_uint64(0)
    n = lib.cmp_log(0xC0FFEE, 4_000_000, C.byref(bad))
    assert n == 0, f"{n} mismatches, first at {bad.value:#018x}"

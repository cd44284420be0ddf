"""include/svo_libm.h is shared by the CPU oracle (oracle/cv_prims.c) and the HIP kernels
(csrc/svo_device.hpp): a defect there would move checker and product together. This test is the
independent check: the header is compiled on its own (gcc, the oracle's flags) and compared with
correctly rounded results computed in extended precision (numpy longdouble: 64-bit mantissa on
x86-64), over the range a rotation vector's angle can take."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHIM = r"""
#include "svo_libm.h"
void sincos_arr(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; i++) svo_sincos(x[i], &s[i], &c[i]); }
void hypot_arr(const double* a, const double* b, double* h, long n) { for (long i = 0; i < n; i++) h[i] = svo_hypot(a[i], b[i]); }
"""


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    d = tmp_path_factory.mktemp("libm")
    src = d / "shim.c"
    src.write_text(SHIM)
    so = d / "libshim.so"
    subprocess.check_call(["gcc", "-O3", "-mavx2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "include"), "-o", str(so), str(src), "-lm"])
    return C.CDLL(str(so))


def _ulps(got, exact_ld):
    """|got - exact| in units of the last place of the correctly rounded double"""
    ref = exact_ld.astype(np.float64)
    ulp = np.spacing(np.abs(ref))
    return np.abs((got.astype(np.longdouble) - exact_ld) / ulp.astype(np.longdouble)).astype(np.float64), ref


@pytest.mark.skipif(np.finfo(np.longdouble).nmant < 63, reason="needs an extended-precision long double")
def test_sincos_within_one_ulp_of_the_correctly_rounded_value(shim):
    rng = np.random.RandomState(5)
    n = 1_000_000
    # rotation-vector angles: mostly small (inter-frame motion), the full turn range, and a few multiples
    x = np.concatenate([rng.uniform(-0.5, 0.5, n // 2), rng.uniform(-2 * np.pi, 2 * np.pi, n // 2 - 1000),
                        rng.uniform(-100, 100, 1000)])
    s, c = np.empty_like(x), np.empty_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    shim.sincos_arr(p(x), p(s), p(c), C.c_long(len(x)))
    xl = x.astype(np.longdouble)
    us, rs = _ulps(s, np.sin(xl))
    uc, rc = _ulps(c, np.cos(xl))
    assert us.max() <= 1.0 and uc.max() <= 1.0, (us.max(), uc.max())
    last_bit = float(np.mean(s != rs)), float(np.mean(c != rc))
    # fdlibm's kernels: < 1 ulp, and correctly rounded for the large majority of arguments
    assert last_bit[0] < 0.2 and last_bit[1] < 0.2, last_bit
    # glibc's sin / cos (what the reference binary would call) against the same truth, for the record
    g = float(np.mean(np.sin(x) != rs)), float(np.mean(np.cos(x) != rc))
    print(f"last-bit differences from the correctly rounded value: svo sin {last_bit[0]:.4f} cos {last_bit[1]:.4f}; "
          f"numpy/glibc sin {g[0]:.4f} cos {g[1]:.4f}; svo vs numpy sin {np.mean(s != np.sin(x)):.4f}")


@pytest.mark.skipif(np.finfo(np.longdouble).nmant < 63, reason="needs an extended-precision long double")
def test_hypot_formula_of_opencv_within_two_ulps(shim):
    """|a| sqrt(1 + (b/a)^2) (OpenCV lapack.cpp) is not the correctly rounded hypot: three roundings,
    at most ~2 ulp. Checked against the extended-precision value; zero and equal arguments exactly."""
    rng = np.random.RandomState(6)
    n = 1_000_000
    a = rng.normal(0, 1, n) * 10.0 ** rng.uniform(-6, 6, n)
    b = rng.normal(0, 1, n) * 10.0 ** rng.uniform(-6, 6, n)
    a[:4] = [0.0, 3.0, -3.0, 0.0]
    b[:4] = [0.0, 4.0, 4.0, -2.5]
    h = np.empty_like(a)
    p = lambda v: v.ctypes.data_as(C.c_void_p)
    shim.hypot_arr(p(a), p(b), p(h), C.c_long(n))
    assert list(h[:4]) == [0.0, 5.0, 5.0, 2.5]
    al, bl = a.astype(np.longdouble), b.astype(np.longdouble)
    u, _ = _ulps(h, np.sqrt(al * al + bl * bl))
    assert u.max() <= 2.0, u.max()

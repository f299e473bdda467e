"""The device's exp/log (g-phocs_amd/csrc/gph_math.h) must be bit-identical to the glibc libm the
reference links.  CPU part: the same header compiled for the host against glibc on millions of
inputs.  GPU part (-m gpu): the functions evaluated ON THE DEVICE plus the device's native
sqrt / divide / floor against the host's."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

SRC = r'''
#include <stdio.h>
#include <math.h>
#include <string.h>
#include "%s/g-phocs_amd/csrc/gph_math.h"
static uint64_t s = 88172645463325252ull;
static uint64_t xr() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double u01() { return (xr() >> 11) * (1.0 / 9007199254740992.0); }
int main() {
  long bad = 0;
  for (long i = 0; i < 3000000; i++) {
    double x, a, b;
    switch (i %% 6) { case 0: x = -u01() * 50; break; case 1: x = (u01() - 0.5) * 0.2; break;
      case 2: x = -u01() * 1e-4; break; case 3: x = (u01() - 0.5) * 1500; break;
      case 4: x = -u01() * 1e-18; break; default: { uint64_t q = xr(); memcpy(&x, &q, 8); if (x != x) x = 1; } }
    a = exp(x); b = gph_exp(x);
    if (memcmp(&a, &b, 8) && !(a != a && b != b)) bad++;
    switch (i %% 6) { case 0: x = u01(); break; case 1: x = 1 + (u01() - 0.5) * 0.2; break;
      case 2: x = u01() * 1e5; break; case 3: x = 1 + (u01() - 0.5) * 1e-6; break;
      case 4: x = u01() * 1e-310; break; default: { uint64_t q = xr(); memcpy(&x, &q, 8); if (x != x) x = 1; } }
    a = log(x); b = gph_log(x);
    if (memcmp(&a, &b, 8) && !(a != a && b != b)) bad++;
  }
  printf("%%ld\n", bad);
  return bad != 0;
}
'''


def test_host_compile_matches_glibc(tmp_path):
    src = tmp_path / "m.cpp"
    src.write_text(SRC % REPO)
    exe = tmp_path / "m"
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout


@pytest.mark.gpu
def test_device_math_bit_identical():
    import gphocs_amd as G
    G.build()
    lib = G.load_library()
    rng = np.random.default_rng(7)
    n = 1 << 18
    x = np.concatenate([rng.uniform(0, 1, n // 4), 1 + rng.uniform(-0.1, 0.1, n // 4),
                        rng.uniform(-50, 50, n // 4), rng.uniform(0, 1e-4, n // 4)])
    y = rng.uniform(0.5, 30323.0, n)
    out = np.zeros(5 * n)
    dp = C.POINTER(C.c_double)
    rc = lib.gph_debug_math(x.ctypes.data_as(dp), y.ctypes.data_as(dp), n, out.ctypes.data_as(dp), 0)
    assert rc == 0
    ex, lg, sq, dv, fl = out.reshape(5, n)
    # reference = the C library's exp/log (numpy's own SIMD exp/log are NOT glibc's)
    libm = C.CDLL("libm.so.6")
    libm.exp.restype = libm.log.restype = C.c_double
    libm.exp.argtypes = libm.log.argtypes = [C.c_double]
    with np.errstate(all="ignore"):
        ref_exp = np.array([libm.exp(v) for v in x])
        assert np.array_equal(ex.view(np.uint64), ref_exp.view(np.uint64)), "device exp != libm exp"
        ref_log = np.array([libm.log(v) for v in x])
        ok = (lg.view(np.uint64) == ref_log.view(np.uint64)) | (np.isnan(lg) & np.isnan(ref_log))
        assert ok.all(), "device log != libm log"
        assert np.array_equal(sq.view(np.uint64), np.sqrt(np.abs(x)).view(np.uint64)), "device sqrt not IEEE"
        assert np.array_equal(dv.view(np.uint64), (x / y).view(np.uint64)), "device divide not IEEE"
        assert np.array_equal(fl, np.floor(x))


def test_all_missing_son_needs_no_select():
    """gph_locus.h: child_factor4 / child_inplace4 drop the reference's `probSum >= CODE_SIZE -> skip`
    (LocusDataLikelihood.c:1660-1663): for a son of N only (four exact 1.0) the general factor
    fl(fl(4 * pe) + fl(1.0 * fl(1 - 4 * pe))) is exactly 1.0 for every edge probability pe in [0, 1/4]"""
    import numpy as np
    rng = np.random.default_rng(7)
    for scale in (0.25, 1e-3, 1e-8, 1e-16, 1e-300):
        pe = rng.random(2_000_000) * scale
        assert np.all(4.0 * pe + 1.0 * (1.0 - 4.0 * pe) == 1.0), scale
    for e in range(-70, -1):            # every binade, random significands
        m = rng.integers(0, 2 ** 52, 100_000, dtype=np.uint64)
        pe = np.ldexp(1.0 + m / 2.0 ** 52, e)
        pe = pe[pe <= 0.25]
        assert np.all(4.0 * pe + (1.0 - 4.0 * pe) == 1.0), e
    pe = np.nextafter(0.25, 0) - np.arange(0, 4096) * 2.0 ** -55      # the last values below 1/4
    assert np.all(4.0 * pe + (1.0 - 4.0 * pe) == 1.0)
    for pe in (0.0, 0.25, 2.0 ** -1074, 2.0 ** -1022, 0.125, 0.1875):
        assert 4.0 * pe + (1.0 - 4.0 * pe) == 1.0

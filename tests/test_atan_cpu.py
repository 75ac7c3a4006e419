"""The arctangent both sides of the tracker parity rest on (TaylorCamera::Project, src/TaylorCamera.cc:243).

Template bytes depend on the last ulp of atan (CVD::transform truncates), so device and oracle both take the
CORRECTLY ROUNDED value, by two independent routes: double-double arithmetic (mcptam_amd/csrc/atan_cr.h, the code the HIP
kernels run, compiled here for the host) and binary128 atanq rounded once (oracle/ba_oracle.c::orc_atan)."""
import ctypes
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_atan_algorithm_equals_rounded_binary128(tmp_path):
    """Both forms: with the fast path (one division, error < 2^-80, accepted only when rounding is unambiguous at 2^-75 -- Ziv's
    strategy) and the double-double evaluation alone.  The fast path must also be what camera arguments actually take."""
    import re
    for flags in ([], ["-DMCP_ATAN_NO_FAST"]):
        exe = str(tmp_path / ("atan_cr_check" + ("_slow" if flags else "")))
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off"] + flags + ["-o", exe, os.path.join(ROOT, "tests", "cpp", "atan_cr_check.cpp"), "-lquadmath"])
        out = subprocess.run([exe, "3000000"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout[-2000:]
        assert "mismatches 0 of 3000000" in out.stdout
        if not flags:
            slow, n = map(int, re.search(r"angle-uniform slow (\d+) of (\d+)", out.stdout).groups())
            assert n == 300000 and slow <= 3, (slow, n)


def test_oracle_atan_is_within_one_ulp_of_libm():
    import oracle
    L = oracle.lib()
    L.orc_atan.restype = ctypes.c_double
    L.orc_atan.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(7)
    xs = np.concatenate([np.tan((rng.random(200000) - 0.5) * 3.0), rng.normal(size=100000) * 10.0 ** rng.integers(-8, 8, 100000)])
    diff = 0
    for x in xs:
        a, b = L.orc_atan(float(x)), math.atan(float(x))
        if a != b:
            diff += 1
            assert abs(a - b) <= math.ulp(b), (x, a, b)
    # glibc 2.35's atan is not correctly rounded, but close: the two agree on all but a fraction of a percent of arguments
    assert diff < 0.005 * len(xs)
    for x, want in ((0.0, 0.0), (1.0, math.pi / 4), (float("inf"), math.pi / 2), (-1.0, -math.pi / 4)):
        assert L.orc_atan(x) == want


def test_oracle_libm_atan_switch_is_off_by_default_and_is_libm():
    """The oracle-only switch behind the 'libm atan' rows of profiles/r05/oracle_sensitivity.md: what a real MCPTAM build calls
    (/root/reference/src/TaylorCamera.cc:216-217) instead of the correctly rounded value; default = correctly rounded."""
    import oracle
    L = oracle.lib()
    L.orc_atan.restype = ctypes.c_double
    L.orc_atan.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(11)
    xs = np.tan((rng.random(100000) - 0.5) * 3.0)
    cr = [L.orc_atan(float(x)) for x in xs]
    L.orc_set_atan_libm(1)
    try:
        lm = [L.orc_atan(float(x)) for x in xs]
    finally:
        L.orc_set_atan_libm(0)
    assert lm == [math.atan(float(x)) for x in xs]
    assert [L.orc_atan(float(x)) for x in xs[:1000]] == cr[:1000]          # switched back
    assert 0 < sum(a != b for a, b in zip(cr, lm)) < 0.005 * len(xs)       # the two definitions do differ (in the last ulp, rarely)

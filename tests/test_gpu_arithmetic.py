"""The arithmetic short cuts of the pair kernels (csrc/sph_fastdiv.hpp: one refined reciprocal per denominator, sqrt
without range scaling) against the IEEE instruction sequences they replace, on the device, over the whole operand window."""
import ctypes as C

import pytest

from sphfluidsimulation_amd import _native as N

pytestmark = pytest.mark.gpu


def test_shared_reciprocal_quotients_and_unscaled_sqrt_are_ieee():
    out = (C.c_uint64 * 4)()
    for seed in (1, 2024):
        N.check(N.lib().sph_selftest_arithmetic(0, 200_000_000, seed, out))
        assert list(out) == [0, 0, 0, 0], f"mismatches (quotient, doubled denominator, sqrt, window test) = {list(out)}"


def _build_and_run(tmp_path, source, args=()):
    """hipcc (same flags as the library: no contraction, correctly rounded divide / sqrt) -> run -> stdout"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / os.path.splitext(os.path.basename(source))[0])
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                           "-I", os.path.join(root, "sphfluidsimulation_amd", "csrc"), "-o", exe, os.path.join(root, source)],
                          stderr=subprocess.DEVNULL)
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600, check=False)


def test_one_step_quotient_is_ieee_on_every_denominator_significand(tmp_path):
    """sph_fastdiv.hpp, SPH_DIV_ONE_STEP: one correction step per numerator.  Every one of the 2^23 denominator significands
    against 2^14 numerator significands spread over the range (1.4e11 pairs; the committed run over all 2^23 x 2^23 pairs is
    profiles/r03b_div_exhaustive.txt), and the refined reciprocal against RN(1 / b) for every significand."""
    r = _build_and_run(tmp_path, "scripts/ubench/div_exhaustive.hip", (14,))
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("pairs per b")][0]
    assert "V1 (rcp + 1 Newton, 1 step) 0 in 0 b's" in line and "div_shared (2 steps) 0 in 0 b's" in line, line
    assert "one Newton step 0," in r.stdout
    sanity = [l for l in r.stdout.splitlines() if l.startswith("sanity")][0]
    assert float(sanity.rsplit("(", 1)[1].split("%")[0]) > 10.0, sanity      # the checker does see a sequence without a correction step fail


def test_residual_step_sqrt_is_ieee_on_every_float_of_its_window(tmp_path):
    """sph_fastdiv.hpp, sqrt_unscaled: every float of [2^-96, 2^101) against sqrtf, and [0, 2^-96) against the callers' bound."""
    r = _build_and_run(tmp_path, "scripts/ubench/sqrt_exhaustive.hip")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "): 0 of 1652555776" in r.stdout and "[0, 2^-96): 0" in r.stdout, r.stdout

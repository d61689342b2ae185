"""The reprojection gather (csrc/reproject.hip) divides by the camera count and by 255 with a
multiply and two FMAs instead of an IEEE division.  That form must give the IEEE quotient bit for
bit wherever the quotient is a normal number: checked here on a sample of all float bit patterns
(tools/div_const_check.c; the exhaustive runs are recorded in its header)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_fma_division_by_constant_equals_ieee_division(tmp_path):
    exe = str(tmp_path / "divc")
    src = os.path.join(ROOT, "tools", "div_const_check.c")
    for flags in (["-mfma"], []):
        r = subprocess.run(["gcc", "-O2", "-ffp-contract=off", *flags, src, "-lm", "-o", exe], capture_output=True)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stderr.decode()
    for c in list(range(2, 17)) + [18, 20, 24, 30, 32, 255]:
        out = subprocess.run([exe, str(c), "8191"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout

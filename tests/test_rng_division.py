"""CPU: the 3-instruction replacement for `(double)r / RAND_MAX` used on the device equals the IEEE quotient for
every one of the 2^31 possible inputs (oracle/check_div_rand_max.c, full sweep, ~2 s on 8 threads)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_div_rand_max_exhaustive(tmp_path):
    exe = str(tmp_path / "divcheck")
    src = os.path.join(ROOT, "oracle", "check_div_rand_max.c")
    # -ffp-contract=off: only the explicit fma() calls fuse; fma() is exact whether it maps to hardware or libm
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", src, "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0 " in out.stdout

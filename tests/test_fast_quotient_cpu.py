"""k_reduce's table quotient (float)((double)a * (1.0 / b)) against the IEEE binary32 division a / b, bit for bit, in C on the
host (tests/cpp/test_fast_quotient.c): random operands, every divisor of the table, numerators as close to a rounding boundary
as binary32 allows, specials.  The kernel itself is compared with the oracle by the GPU parity tests."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_quotient_equals_ieee_division(tmp_path):
    exe = str(tmp_path / "test_fast_quotient")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-ffp-contract=off", "-msse2", "-mfpmath=sse",
                           os.path.join(ROOT, "tests", "cpp", "test_fast_quotient.c"), "-o", exe, "-lm"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout
    assert " 0 mismatches" in p.stdout

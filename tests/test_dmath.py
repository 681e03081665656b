"""csrc/dmath.h (the policy heads' exp / log / tanh in double) against libm, on the CPU: the header is host-compilable, so the
code the kernels run is the code checked here.  Built with g++ from tests/native/dmath_check.cpp."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_dmath_matches_libm(tmp_path):
    exe = str(tmp_path / "dmath_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "native", "dmath_check.cpp")], check=True)
    out = subprocess.run([exe, "600000"], check=True, capture_output=True, text=True).stdout
    rows = re.findall(r"^(\w+) max_rel_err (\S+) at \S+ float_mismatches (\d+) of (\d+)$", out, re.M)
    assert len(rows) == 10, out
    assert re.search(r"^nonfinite_failures 0$", out, re.M), out   # NaN / +-inf / out-of-range arguments behave like libm
    for name, err, mism, n in rows:
        # a few double ulps; rounded to float (what every caller keeps) the results are libm's
        assert float(err) < 5e-15, (name, err)
        assert int(mism) == 0, (name, mism, n)

"""CPU-only: the table-driven atan2 routines of the AFC phase detector, built for the host from the same headers + tables the
kernels use, against glibc atan2: csrc/opv_atan2.h (the product's two) and tests/atan/opv_atan2_q3.h (the (k, h) form the
shipped 1025-row table was derived from)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
INC = ["-I", str(ROOT / "opv-cxx-demod_amd" / "csrc"), "-I", str(ROOT / "tests" / "atan")]
SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "opv_atan2.h"
#include "opv_atan2_q3.h"
int main() {
    double maxabs = 0, maxrel = 0;
    srand48(7);
    for (long i = 0; i < 4000000; i++) {
        double s = exp(drand48() * 60 - 30);
        double y = (drand48() * 2 - 1) * s, x = (drand48() * 2 - 1) * s;
        if (i % 5 == 0) y *= 1e-6;
        if (i % 7 == 0) x *= 1e-4;
        if (i % 11 == 0) { double t = x; x = y; y = t; }
        if (x == 0 && y == 0) continue;
        double a = OPV_FN(y, x), b = atan2(y, x), e = fabs(a - b);
        if (e > maxabs) maxabs = e;
        if (b != 0 && e / fabs(b) > maxrel) maxrel = e / fabs(b);
    }
    /* axes and diagonals */
    double ax[][2] = {{0,1},{0,-1},{1,0},{-1,0},{1,1},{-1,1},{1,-1},{-1,-1},{1e-300,1},{1,1e-300}};
    for (unsigned k = 0; k < sizeof ax / sizeof ax[0]; k++) {
        double e = fabs(OPV_FN(ax[k][0], ax[k][1]) - atan2(ax[k][0], ax[k][1]));
        if (e > maxabs) maxabs = e;
    }
    printf("%.3e %.3e\n", maxabs, maxrel);
    return 0;
}
'''


SRC_Q = SRC.replace("OPV_FN(", "opv_atan2_q(")


def test_atan2_q_table_matches_libm(tmp_path):
    """opv_atan2_q (the four- and sixteen-streams-per-wave front-ends' angle: pi/4 + atan((|y| - |x|) / (|y| + |x|)), 257 rows,
    degree 5, no octant fix-up): ABSOLUTE accuracy like opv_atan2; the relative accuracy of tiny angles is that of an angle near
    pi/4 by construction (the AFC integrates the angle), so only the absolute error is asserted. Axis arguments exact."""
    c = tmp_path / "t.cpp"
    c.write_text(SRC_Q)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", *INC, str(c), "-o",
                    str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert float(out[0]) < 5e-16
    c.write_text(r"""
#include <math.h>
#include <stdio.h>
#include "opv_atan2.h"
int main() {
    const double pi = 3.14159265358979323846;
    int ok = opv_atan2_q(0.0, 1.0) == 0.0 && opv_atan2_q(0.0, -1.0) == pi && opv_atan2_q(1.0, 0.0) == pi / 2 &&
             opv_atan2_q(-1.0, 0.0) == -pi / 2 && opv_atan2_q(3.0, 3.0) == pi / 4 && opv_atan2_q(-2.0, -2.0) == -3 * pi / 4;
    printf("%d\n", ok);
    return 0;
}
""")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", *INC, str(c), "-o",
                    str(exe), "-lm"], check=True)
    assert subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip() == "1"


def test_table_is_reproducible(tmp_path):
    """The committed tables equal what tools/gen_atan_table.py generates (mpmath, 60 digits)."""
    import shutil
    pkg = ROOT / "opv-cxx-demod_amd"
    incs = [pkg / "csrc" / "opv_atan_table_q.inc", pkg / "csrc" / "opv_atan_table_q3r.inc",
            ROOT / "tests" / "atan" / "opv_atan_table_q3.inc"]
    before = [inc.read_text() for inc in incs]
    for k, inc in enumerate(incs):
        shutil.copy(inc, tmp_path / f"keep{k}.inc")
    try:
        subprocess.run(["python3", str(pkg / "tools" / "gen_atan_table.py")], check=True, capture_output=True)
        assert [inc.read_text() for inc in incs] == before
    finally:
        for k, inc in enumerate(incs):
            shutil.copy(tmp_path / f"keep{k}.inc", inc)


def test_atan2_q3_table_is_within_its_stated_error(tmp_path):
    """opv_atan2_q3 (1025 rows, cubic: the one-wave front-end's angle from round 3 on): absolute error against glibc below
    1e-13 rad over 4e6 random arguments, axes exact - the accuracy csrc/opv_atan2.h states and NOTEBOOK.md §3.1 prices (AFC steady
    state 3e-10 Hz, soft symbols 1e-17)."""
    c = tmp_path / "t.cpp"
    c.write_text(SRC.replace("OPV_FN(", "opv_atan2_q3("))
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", *INC, str(c), "-o",
                    str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert float(out[0]) < 1e-13
    chk = tmp_path / "a.cpp"
    chk.write_text('#include <stdio.h>\n#include "opv_atan2_q3.h"\nint main(){printf("%a %a\\n", opv_atan2_q3(0.0, 1.0), opv_atan2_q3(0.0, 5e7));return 0;}')
    subprocess.run(["g++", "-O2", "-ffp-contract=off", *INC, str(chk), "-o", str(tmp_path / "a"), "-lm"], check=True)
    assert subprocess.run([str(tmp_path / "a")], capture_output=True, text=True, check=True).stdout.split() == ["0x0p+0", "0x0p+0"]


def test_atan2_q3r_is_q3_written_in_the_argument(tmp_path):
    """opv_atan2_q3r (what the kernel evaluates: the same 1025 cubics re-expanded in the argument itself, no k / h): within
    1e-13 rad of glibc and within 1e-15 of opv_atan2_q3 over 4e6 random arguments."""
    src = SRC.replace("double a = OPV_FN(y, x), b = atan2(y, x), e = fabs(a - b);",
                      "double a = opv_atan2_q3r(y, x), b = atan2(y, x), e = fabs(a - b); if (fabs(a - opv_atan2_q3(y, x)) > 1e-15) e = 1.0;")
    src = src.replace("OPV_FN(ax[k][0], ax[k][1])", "opv_atan2_q3r(ax[k][0], ax[k][1])")
    assert "opv_atan2_q3r(y, x)" in src
    c = tmp_path / "t.cpp"
    c.write_text(src)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", *INC, str(c), "-o",
                    str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert float(out[0]) < 1e-13

"""Coefficients of pmx_exp_poly / pmx_exp2 (pharmsol_amd/csrc/pmx_structures.hpp): Chebyshev fits (near-minimax) of
e^r on |r| <= ln2/2 and 2^r on |r| <= 1/2 in 60-digit arithmetic, rounded to double, with the worst relative error of a
double-precision Horner evaluation over 40001 points.  usage: python tools/exp_poly_fit.py [degree]"""
import sys

import mpmath as mp

mp.mp.dps = 60


def fit(f, a, degree):
    c, err = mp.chebyfit(f, [-a, a], degree + 1, error=True)
    cd = [float(k) for k in c]
    worst = mp.mpf(0)
    for i in range(40001):
        r = -a + 2 * a * i / 40000
        p = 0.0
        for k in cd:
            p = p * r + k
        ex = f(mp.mpf(r))
        worst = max(worst, abs((mp.mpf(p) - ex) / ex))
    return cd, err, worst


if __name__ == "__main__":
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    for name, f, a in (("2^r, |r|<=1/2", lambda r: mp.mpf(2) ** r, 0.5), ("e^r, |r|<=0.347", lambda r: mp.e ** r, 0.3470)):
        cd, err, worst = fit(f, a, deg)
        print(f"{name}: degree {deg}, approximation error {mp.nstr(err, 4)}, double Horner worst rel {mp.nstr(worst, 4)}")
        for k in cd:
            print("   ", repr(k))

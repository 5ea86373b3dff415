#!/usr/bin/env python3
"""Prints the constant tables of dagsfm_amd/csrc/exact_trig.h (mathematical constants to ~200 bits, split into
doubles): pi/2 in four parts (33 + 33 + 33 + 53 bits: the first three times a small integer are exact), pi/2 as a
double-double, 1/n! and 1/(2n+1) as double-doubles, atan(k/8) as double-doubles.  Pure integer arithmetic."""
from fractions import Fraction

BITS = 400


def atan_inv(n):  # atan(1/n) * 2^BITS, integer series
    one = 1 << BITS
    x = one // n
    n2 = n * n
    s, term, k = 0, x, 0
    while term:
        s += term // (2 * k + 1) if k % 2 == 0 else -(term // (2 * k + 1))
        term //= n2
        k += 1
    return s


PI = Fraction(4 * (4 * atan_inv(5) - atan_inv(239)), 1 << BITS)   # Machin


def atan_frac(q):   # atan of a Fraction 0 <= q <= 1, via argument halving to a small argument + series
    if q == 0:
        return Fraction(0)
    if q == 1:
        return PI / 4
    # atan(q) = 2 atan(q / (1 + sqrt(1 + q^2))): do it in scaled integers, three times, then the series
    one = 1 << BITS
    x = (q.numerator * one) // q.denominator
    from math import isqrt
    for _ in range(3):
        x = (x * one) // (one + isqrt(one * one + x * x))
    s, term, k, x2 = 0, x, 0, (x * x) // one
    while term:
        s += term // (2 * k + 1) if k % 2 == 0 else -(term // (2 * k + 1))
        term = (term * x2) // one
        k += 1
    return Fraction(8 * s, one)


def dd(fr):
    hi = float(fr)
    lo = float(fr - Fraction(hi))
    return hi, lo


def part(fr, bits):   # the leading `bits` bits of fr as a double (truncated), and the rest
    from math import frexp
    m, e = frexp(float(fr))
    scale = Fraction(2) ** (bits - e)
    p = Fraction(int(fr * scale), 1) / scale
    return float(p), fr - p


def main():
    h = PI / 2
    p0, r = part(h, 33)
    p1, r = part(r, 33)
    p2, r = part(r, 33)
    p3 = float(r)
    print("static const double kPio2Parts[4] = {%s, %s, %s, %s};" % tuple(x.hex() for x in (p0, p1, p2, p3)))
    print("static const double kPio2DD[2] = {%s, %s};" % tuple(x.hex() for x in dd(h)))
    print("static const double kTwoOverPi = %s;" % float(2 / PI).hex())
    f = Fraction(1)
    rows = []
    for n in range(0, 30):
        if n:
            f /= n
        rows.append(dd(f))
    print("static const double kInvFact[30][2] = {")
    for hi, lo in rows:
        print("    {%s, %s}," % (hi.hex(), lo.hex()))
    print("};")
    print("static const double kInvOdd[16][2] = {   // 1 / (2n + 1)")
    for n in range(16):
        hi, lo = dd(Fraction(1, 2 * n + 1))
        print("    {%s, %s}," % (hi.hex(), lo.hex()))
    print("};")
    print("static const double kAtanEighths[9][2] = {   // atan(k / 8)")
    for k in range(9):
        hi, lo = dd(atan_frac(Fraction(k, 8)))
        print("    {%s, %s}," % (hi.hex(), lo.hex()))
    print("};")


if __name__ == "__main__":
    main()

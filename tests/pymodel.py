"""A second, independent statement of the arithmetic contract for SMALL cases: exact
rational arithmetic (fractions.Fraction) with an explicit round-to-nearest-even to
binary32 after every operation.  It shares no code with oracle/rtr_oracle.c and does not
depend on any compiler flag, so agreement pins the C oracle's fp32 behaviour (contraction,
division, rintf) -- the reference itself has no fixtures to pin it with."""
from fractions import Fraction

import numpy as np

_INF = float("inf")


def rnd(q):
    """Fraction -> nearest binary32 value (ties to even), returned as a Python float."""
    if q == 0:
        return 0.0
    sign = -1 if q < 0 else 1
    a = abs(q)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    n, r = divmod(a, ulp)
    n = int(n)
    if r * 2 > ulp or (r * 2 == ulp and (n & 1)):
        n += 1
    v = n * ulp
    if v >= Fraction(2) ** 128:
        return sign * _INF
    return sign * float(v)  # exact: v has <= 24 significant bits


def F(x):
    return Fraction(float(np.float32(x)))


def _special(*v):
    return any(x != x or x in (_INF, -_INF) for x in v)


def mul(a, b):
    if _special(a, b):
        with np.errstate(all="ignore"):
            return float(np.float32(a) * np.float32(b))  # inf / nan propagation only
    return rnd(F(a) * F(b))


def add(a, b):
    if _special(a, b):
        with np.errstate(all="ignore"):
            return float(np.float32(a) + np.float32(b))
    return rnd(F(a) + F(b))


def fma(a, b, c):
    if _special(a, b, c):
        with np.errstate(all="ignore"):
            return float(np.float32(np.float32(a) * np.float32(b)) + np.float32(c))  # exact enough for inf / nan
    return rnd(F(a) * F(b) + F(c))


def div(a, b):
    if _special(a, b):
        with np.errstate(all="ignore"):
            return float(np.float32(a) / np.float32(b))
    return rnd(F(a) / F(b))


def rint(a):
    a = Fraction(float(a))
    n = a.numerator // a.denominator
    r = a - n
    if r * 2 > 1 or (r * 2 == 1 and (n & 1)):
        n += 1
    return float(n)


def project_point(P, x, y, z, W, H):
    """render.cu:33-40, 62-70 under the contract of SURVEY.md 8c.  -> (pixel id | -1, depth)"""
    P = [float(np.float32(v)) for v in np.asarray(P).reshape(-1)]
    x, y, z = (float(np.float32(v)) for v in (x, y, z))
    r = []
    for k in range(3):
        t = mul(P[4 * k], x)
        t = fma(P[4 * k + 1], y, t)
        t = fma(P[4 * k + 2], z, t)
        r.append(add(t, P[4 * k + 3]))
    rx, ry, rz = r
    if not rz > 0.0:
        return -1, None
    inv = div(1.0, rz)
    qx, qy = mul(rx, inv), mul(ry, inv)
    if _special(qx, qy):  # +-inf / nan fail the range test below (render.cu:68)
        return -1, None
    fu, fv = rint(qx), rint(qy)
    if not (0.0 <= fu < W and 0.0 <= fv < H):
        return -1, None
    return int(fv) * W + int(fu), rz

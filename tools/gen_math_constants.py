#!/usr/bin/env python3
"""Derive the constants of the portable sin/cos/log used by BOTH the oracle
(oracle/portable_math.c) and the HIP kernels (grl_amd/csrc/grlx_math.h).

Everything is derived from first principles with exact rational arithmetic:
  * pi to 120 decimal digits (well-known expansion), split into three doubles
    P1+P2+P3 ~= pi/2 (each correctly rounded remainder of the previous ones);
  * Taylor coefficients 1/k! rounded to nearest double.
Run:  python tools/gen_math_constants.py
"""
from fractions import Fraction
from math import factorial

PI_STR = ("3.14159265358979323846264338327950288419716939937510"
          "58209749445923078164062862089986280348253421170679"
          "8214808651328230664709384460955058223172535940812848111745")
PI = Fraction(PI_STR)

def rn(fr: Fraction) -> float:
    """round-to-nearest-even double of an exact rational (python's int/int
    true division is correctly rounded)."""
    return fr.numerator / fr.denominator

def show(name, v):
    print(f"#define {name:<10} {float(v).hex():<26} /* {v!r} */")

pio2 = PI / 2
p1 = rn(pio2); p2 = rn(pio2 - Fraction(p1)); p3 = rn(pio2 - Fraction(p1) - Fraction(p2))
show("PM_PIO2_1", p1); show("PM_PIO2_2", p2); show("PM_PIO2_3", p3)
show("PM_INVPIO2", rn(2 / PI))
show("PM_PI", rn(PI)); show("PM_2PI", 2 * rn(PI))
print("/* sin: r + r^3*(S1 + z*(S2 + ... S8)),  Sk = (-1)^k/(2k+1)! */")
for k in range(1, 9):
    show(f"PM_S{k}", rn(Fraction((-1) ** k, factorial(2 * k + 1))))
print("/* cos: 1 - z/2 + z^2*(C1 + z*(C2 + ... C8)), Ck = (-1)^(k+1)/(2k+2)! */")
for k in range(1, 9):
    show(f"PM_C{k}", rn(Fraction((-1) ** (k + 1), factorial(2 * k + 2))))
# log constants: ln2 split (hi has 32 trailing zero bits so k*hi is exact for |k|<2^20)
import struct
LN2_STR = ("0.69314718055994530941723212145817656807550013436025"
           "52541206800094933936219696947156058633269964186875")
LN2 = Fraction(LN2_STR)
l = rn(LN2)
bits = struct.unpack("<Q", struct.pack("<d", l))[0] & ~((1 << 32) - 1)
ln2hi = struct.unpack("<d", struct.pack("<Q", bits))[0]
ln2lo = rn(LN2 - Fraction(ln2hi))
show("PM_LN2_HI", ln2hi); show("PM_LN2_LO", ln2lo)
show("PM_SQRT2", rn(Fraction("1.41421356237309504880168872420969807856967187537694807317667973799")))
print("/* log: atanh series 2*(s + s^3/3 + s^5/5 + ...), Lk = 2/(2k+1) */")
for k in range(1, 12):
    show(f"PM_L{k}", rn(Fraction(2, 2 * k + 1)))

"""Test helper: log2 of positive doubles, correctly rounded, computed independently of the device code: 80-bit
long-double log2 decides every value whose fractional position between two doubles is clear of a half-way point by
more than the long double's own error, and 50-digit decimal arithmetic decides the rest.  TEST INFRASTRUCTURE ONLY."""
from decimal import Decimal, getcontext

import numpy as np


def log2_correctly_rounded(t: np.ndarray) -> np.ndarray:
    shape = np.shape(t)
    t = np.asarray(t, dtype=np.float64).ravel()
    y = np.log2(t.astype(np.longdouble))
    d = y.astype(np.float64)  # candidate: the long double rounded to nearest
    # distance of y from the nearest half-way point between doubles, in units of the double's ulp
    up = np.nextafter(d, np.inf)
    ulp = (up - d).astype(np.longdouble)
    frac = np.abs((y - d.astype(np.longdouble)) / ulp)  # in [0, 0.5]: 0 = on the double, 0.5 = half-way
    unsure = np.flatnonzero((0.5 - frac) < 2.0 ** -8)   # the long double carries 11 more bits: 2^-8 is a wide margin
    if unsure.size:
        getcontext().prec = 60
        ln2 = Decimal(2).ln()
        for k in unsure:
            exact = Decimal(float(t[k])).ln() / ln2
            lo, hi = (d[k], np.nextafter(d[k], np.inf)) if Decimal(float(d[k])) <= exact else (np.nextafter(d[k], -np.inf), d[k])
            mid = (Decimal(float(lo)) + Decimal(float(hi))) / 2
            d[k] = hi if exact > mid else lo  # (log2 of a double is never exactly half-way unless it is exact)
    return d.reshape(shape)

#!/usr/bin/env python3
"""Exhaustive check behind csrc/atc_device.h:py_mod360 — for every float32 a with 2^-12 <= |a| < 2^25 the quotient estimate
floor(a * RN(1/360)) (fp32 multiply) is never SMALLER than floor(a / 360), and is larger (by one) only for a >= 1799.9999 just
below a multiple of 360 — the one case the function's fix-up handles.  ~1 minute of numpy."""
import numpy as np

c = np.float32(1.0 / 360.0)
too_big = too_small = 0
first_big = None
for sign in (1, -1):
    for e in range(-12, 25):
        lo = np.float32(2.0 ** e).view(np.uint32)
        t = np.arange(lo, lo + (1 << 23), dtype=np.uint32).view(np.float32) * np.float32(sign)
        q = np.floor(t * c)
        d = q.astype(np.float64) - np.floor(t.astype(np.float64) / 360.0)
        assert d.max() <= 1 and d.min() >= -1
        if (d > 0).any() and first_big is None:
            first_big = float(t[np.nonzero(d > 0)[0][0]])
        too_big += int((d > 0).sum())
        too_small += int((d < 0).sum())
print("quotient one too big: %d inputs (first %r); too small: %d" % (too_big, first_big, too_small))
assert too_small == 0 and first_big is not None and first_big > 1799.0

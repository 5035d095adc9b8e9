#!/usr/bin/env python3
"""Derives the float64 heading-kinematics polynomials of include/atc_step.h (fp32 specification, ABI 18).

Heading state is 32-bit fixed point (deg = 180 + fix * 2^-23).  k = rint(fix / (180 * 2^23)), t = fix - k * 180 * 2^23 is the
remainder in COUNTS (|t| <= 90 * 2^23), r = t * kappa its value in radians (kappa = pi / (180 * 2^23)), u = t * t:
    sin r = t * (S0 + u (S1 + u (S2 + u (S3 + u (S4 + u S5)))))        S_i = s_i kappa^(2 i + 1)
    cos r = 1 + u (C1 + u (C2 + u (C3 + u (C4 + u C5))))               C_i = c_i kappa^(2 i)
with (s_i), (c_i) the coefficients of near-minimax polynomials in r^2 on |r| <= pi/2 (Chebyshev interpolation of
sin(r)/r and cos(r) in the variable r^2, computed with 60 digits).  Prints the #defines and the measured maximum errors."""
import mpmath as mp

mp.mp.dps = 60
A = (mp.pi / 2) ** 2 * mp.mpf("1.0001")   # the interval in z = r^2, with a hair of slack


def cheb_fit(f, deg):
    n = deg + 1
    nodes = [A / 2 * (1 + mp.cos(mp.pi * (2 * j + 1) / (2 * n))) for j in range(n)]
    M = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, z in enumerate(nodes):
        for j in range(n):
            M[i, j] = z ** j
        b[i] = f(z)
    return list(mp.lu_solve(M, b))


def f_sin(z):
    r = mp.sqrt(z)
    return mp.sin(r) / r if z else mp.mpf(1)


s = cheb_fit(f_sin, 5)
c = cheb_fit(lambda z: mp.cos(mp.sqrt(z)), 5)
c[0] = mp.mpf(1)   # pinned: cos(0) = 1 exactly (the fit gives 1 - 2e-11)
kappa = mp.pi / (180 * 2 ** 23)
S = [s[i] * kappa ** (2 * i + 1) for i in range(6)]
C = [c[i] * kappa ** (2 * i) for i in range(6)]


def ev(co, u):
    acc = co[-1]
    for k in reversed(co[:-1]):
        acc = acc * u + k
    return acc


worst_s = worst_c = 0
for j in range(20001):
    t = mp.mpf(90 * 2 ** 23) * (mp.mpf(j) / 10000 - 1)
    u = t * t
    worst_s = max(worst_s, abs(t * ev([mp.mpf(float(v)) for v in S], u) - mp.sin(t * kappa)))
    worst_c = max(worst_c, abs(ev([mp.mpf(1)] + [mp.mpf(float(v)) for v in C[1:]], u) - mp.cos(t * kappa)))
print("/* max |error| with the float64-rounded coefficients in exact arithmetic: sin %.2e, cos %.2e */" % (worst_s, worst_c))
for i, v in enumerate(S):
    print("#define ATC_KIN_S%d (%s)" % (i, float(v).hex()))
for i, v in enumerate(C[1:], 1):
    print("#define ATC_KIN_C%d (%s)" % (i, float(v).hex()))
print("#define ATC_KIN_INV180 (%s)   /* 1 / (180 * 2^23) */" % float(mp.mpf(1) / (180 * 2 ** 23)).hex())

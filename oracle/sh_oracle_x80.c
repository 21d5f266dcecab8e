/*
 * oracle/sh_oracle_x80.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * sh_oracle.c (the restatement of the reference's spherical-harmonics solvers, fluxes.py:2675-3635) compiled in x87
 * extended precision: every `double` of that file -- arguments, work arrays, the banded LU -- becomes `long double`,
 * and <tgmath.h> turns its exp / sqrt / pow / fabs calls into the long-double ones.  The system headers are read first,
 * with the real `double`.  Callers hand over numpy longdouble arrays (oracle.py: get_reflected_SH(..., x80=True)).
 * What it is for: near-conservative scattering (w0 -> 1) makes the reference's own SH4 formulas ill-conditioned
 * (Q = (a0 a1 / lam^2 - 1) / 2 with a0 a1 / lam^2 -> 1, fluxes.py:3423-3425); |fp64 - x80| of this file is how far the
 * reference's fp64 rounding moves a result, which the tests allow a kernel on top of their 1e-9.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <tgmath.h>
#define double long double
#include "sh_oracle.c"

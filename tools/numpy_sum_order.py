#!/usr/bin/env python
"""The order in which numpy sums a contiguous float64 vector, restated in plain Python and held against np.sum: pairwise
blocks of <= 128 terms with eight partial sums (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum), applied to
chunks of 8 192 elements (the ufunc buffer size) whose sums are added to the running result in order.
csrc/integrals.hip sums in this order so that the spectrum-wide integrals computed on the device carry numpy's bits."""
import numpy as np


def leaf(a):
    n = len(a)
    if n < 8:
        r = np.float64(-0.0)
        for v in a:
            r = r + v
        return r
    r = [a[j] for j in range(8)]
    body = n - n % 8
    for i in range(8, body, 8):
        for j in range(8):
            r[j] = r[j] + a[i + j]
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    for i in range(body, n):
        res = res + a[i]
    return res


def pairwise(a):
    n = len(a)
    if n <= 128:
        return leaf(a)
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise(a[:n2]) + pairwise(a[n2:])


def numpy_sum(a, chunk=8192):
    a = [np.float64(v) for v in a]
    acc = pairwise(a[:chunk])
    for k in range(chunk, len(a), chunk):
        acc = acc + pairwise(a[k:k + chunk])
    return acc


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for m in [1, 7, 8, 100, 129, 1000, 4095, 8192, 8193, 9998, 20000, 99999, 262144]:
        a = rng.normal(size=m) * 10.0 ** rng.uniform(-3, 3, m)
        print(m, np.sum(a) == numpy_sum(a), "unchunked:", np.sum(a) == pairwise([np.float64(v) for v in a]))

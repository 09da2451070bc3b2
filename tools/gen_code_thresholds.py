#!/usr/bin/env python3
"""
Thresholds for the exact nearest-code search of quantize_4bit (csrc/quant_kernels.hip, nearest_code).

The reference picks argmin_i fl(|xn - code[i]|) in f32 with the first minimum (functional.py:242-243).  For a
sorted table that decision is monotone in xn, so it equals  #{i : xn > t_i}  for per-pair thresholds t_i = the
largest f32 x that still prefers code[i] over code[i+1] under the reference's own rounded comparison.  This script
finds every t_i by bisection over the f32 number line, checks the counting rule against the brute-force argmin on
a few million values (all values within 4096 ulps of each threshold and of each code, plus random ones), and prints
the constants.  FP4 is searched on |xn| (its table is symmetric); see nearest_code for the sign / zero rule.
"""
import numpy as np

NF4 = np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                0.7229568362236023, 1.0], dtype=np.float32)
FP4_POS = np.array([0.0, 0.0625, 0.125, 0.25, 0.375, 0.5, 0.75, 1.0], dtype=np.float32)
FP4 = np.concatenate([FP4_POS, -FP4_POS]).astype(np.float32)


def f2i(x):   # order-preserving integer key of an f32
    u = np.asarray(x, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.where(u < 0, -(u & 0x7FFFFFFF), u)


def i2f(k):
    k = np.asarray(k, dtype=np.int64)
    u = np.where(k < 0, (-k) | 0x80000000, k).astype(np.uint32)
    return u.view(np.float32)


def prefers_lower(x, a, b):
    """reference rule between adjacent codes a < b: a wins on <= (first minimum)."""
    x = np.float32(x)
    return np.abs(x - a) <= np.abs(x - b)


def threshold(a, b):
    lo, hi = int(f2i(a)), int(f2i(b))     # prefers a at lo, b at hi
    assert prefers_lower(i2f(lo), a, b) and not prefers_lower(i2f(hi), a, b)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if prefers_lower(i2f(mid), a, b):
            lo = mid
        else:
            hi = mid
    return i2f(lo)


def brute(x, table):
    d = np.abs(x[:, None].astype(np.float32) - table[None, :].astype(np.float32))
    return np.argmin(d, axis=1)


def check(table, thr, fp4):
    rng = np.random.default_rng(0)
    pts = [rng.uniform(-1, 1, 2_000_000).astype(np.float32), rng.standard_normal(500_000).astype(np.float32)]
    for c in np.concatenate([np.abs(table), thr, -thr if fp4 else thr]):
        k = int(f2i(np.float32(c)))
        pts.append(i2f(np.arange(k - 4096, k + 4097)))
    x = np.concatenate(pts).astype(np.float32)
    ref = brute(x, table)
    if fp4:
        mag = (np.abs(x)[:, None] > thr[None, :]).sum(axis=1)
        got = np.where((x < 0) & (mag > 0), 8 + mag, mag)
    else:
        got = (x[:, None] > thr[None, :]).sum(axis=1)
    bad = np.nonzero(got != ref)[0]
    assert bad.size == 0, (x[bad[:5]], got[bad[:5]], ref[bad[:5]])
    return x.size


if __name__ == "__main__":
    nf4_thr = np.array([threshold(NF4[i], NF4[i + 1]) for i in range(15)], dtype=np.float32)
    fp4_thr = np.array([threshold(FP4_POS[i], FP4_POS[i + 1]) for i in range(7)], dtype=np.float32)
    n1, n2 = check(NF4, nf4_thr, False), check(FP4, fp4_thr, True)
    print(f"// verified against the brute-force argmin on {n1} (NF4) / {n2} (FP4) f32 values")
    print("NF4:", ", ".join(f"{float(t):.9g}f /*0x{np.float32(t).view(np.uint32):08x}*/" for t in nf4_thr))
    print("FP4:", ", ".join(f"{float(t):.9g}f /*0x{np.float32(t).view(np.uint32):08x}*/" for t in fp4_thr))

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


def bin_table(thr, lo, scale):
    """256-bin first guess for the threshold count: bin b = int(x * scale + 256 * lo_off) covers [e_b, e_b+1).  The kernel
    computes the bin with one fma (error < 2^-24 in x near a bin edge), so idx_low[b] counts the thresholds safely below
    the bin (< e_b - 2^-23) and at most ONE further threshold may lie below e_{b+1} + 2^-23: the kernel settles it with one
    exact compare, idx = idx_low + (x > thr[idx_low]) (thr padded with +inf)."""
    thr = np.asarray(thr, dtype=np.float64)
    out = []
    for b in range(256):
        e0, e1 = lo + b / scale, lo + (b + 1) / scale
        low = int((thr < e0 - 2.0 ** -23).sum())
        assert int((thr < e1 + 2.0 ** -23).sum()) - low <= 1, (b, e0, e1)
        out.append(low)
    return np.array(out, dtype=np.uint8)


def check_bins(thr, table, lo, scale, absolute):
    rng = np.random.default_rng(1)
    pts = [rng.uniform(-1.001, 1.001, 2_000_000).astype(np.float32), np.float32([-1.0, 1.0, 0.0, -0.0, 2.0, -2.0])]
    for c in thr:
        k = int(f2i(np.float32(c)))
        pts.append(i2f(np.arange(k - 4096, k + 4097)))
    for b in range(257):   # every bin edge and its f32 neighbourhood
        k = int(f2i(np.float32(lo + b / scale)))
        pts.append(i2f(np.arange(k - 64, k + 65)))
    x = np.concatenate(pts).astype(np.float32)
    a = np.abs(x) if absolute else x
    want = (a[:, None] > thr[None, :]).sum(axis=1)
    # the kernel's bin: fl(a * scale + off) in f32 (one rounding, fma), truncated, clamped to 0..255
    y = (a.astype(np.float64) * scale - lo * scale).astype(np.float32)
    b = np.clip(np.trunc(y).astype(np.int64), 0, 255)
    low = table[b].astype(np.int64)
    pad = np.concatenate([thr, np.float32([np.inf])]).astype(np.float32)
    got = low + (a > pad[low])
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (x[bad[:5]], got[bad[:5]], want[bad[:5]])
    return x.size


def c_array(name, t):
    rows = [", ".join(str(int(v)) for v in t[i:i + 32]) for i in range(0, 256, 32)]
    return f"static __device__ const uint8_t {name}[256] = {{\n    " + ",\n    ".join(rows) + "};"


if __name__ == "__main__":
    nf4_thr = np.array([threshold(NF4[i], NF4[i + 1]) for i in range(15)], dtype=np.float32)
    fp4_thr = np.array([threshold(FP4_POS[i], FP4_POS[i + 1]) for i in range(7)], dtype=np.float32)
    n1, n2 = check(NF4, nf4_thr, False), check(FP4, fp4_thr, True)
    print(f"// verified against the brute-force argmin on {n1} (NF4) / {n2} (FP4) f32 values")
    print("NF4:", ", ".join(f"{float(t):.9g}f /*0x{np.float32(t).view(np.uint32):08x}*/" for t in nf4_thr))
    print("FP4:", ", ".join(f"{float(t):.9g}f /*0x{np.float32(t).view(np.uint32):08x}*/" for t in fp4_thr))
    # first-guess bin tables of nearest_code_lut: NF4 on xn in [-1, 1) (bin = xn * 128 + 128), FP4 on |xn| (bin = |xn| * 256)
    nf4_bins, fp4_bins = bin_table(nf4_thr, -1.0, 128.0), bin_table(fp4_thr, 0.0, 256.0)
    m1, m2 = check_bins(nf4_thr, nf4_bins, -1.0, 128.0, False), check_bins(fp4_thr, fp4_bins, 0.0, 256.0, True)
    import os
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mps_bitsandbytes_amd", "csrc", "code_bins.inc")
    with open(inc, "w") as f:
        f.write("// generated by tools/gen_code_thresholds.py -- first-guess bins of nearest_code_lut (quant_kernels.hip);\n"
                f"// the bin rule was checked against the threshold count on {m1} (NF4) / {m2} (FP4) f32 values\n")
        f.write(c_array("g_nf4_bins", nf4_bins) + "\n" + c_array("g_fp4_bins", fp4_bins) + "\n")
    print("wrote", os.path.normpath(inc))

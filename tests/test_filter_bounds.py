"""Soundness of the MFMA pre-filter's thresholds (csrc/mlf_filter_dev.hpp: filter_thresholds),
checked on the CPU with a numpy emulation of the device arithmetic: binary16 operands, exact
products, float32 accumulation -- once in a deliberately BAD order (reversed, one rounding per product) and once the
way v_mfma_f32_32x32x16_f16 was found to accumulate on gfx950 (scripts/probes/mfma16_acc_probe.hip: per instruction
two groups of 8 exact products, each group added to the accumulator with one rounding; also with a rounded tree
inside the groups) -- compared with the reference's sequential binary64 distance.

Property (for every pair): Dt <= T_lo  =>  s <= r2 ;  Dt > T_hi  =>  s > r2.
"""
import numpy as np
import pytest


def thresholds(sigma, namax, nbn2, r2, K):
    """float64 restatement of filter_thresholds() incl. the directed float32 rounding"""
    nbn = np.sqrt(nbn2)
    delta = 2.0**-11 * (1 + 2.0**-9) * (namax + nbn) + 2 * np.sqrt(K) * 2.0**-24 + 2.0**-40 * (namax + nbn)
    w = namax + nbn + 2.0**-8
    eacc = 2.0**-15 * w * w + 2.0**-22
    sr = sigma * np.sqrt(r2)
    lo = sr * (1 - 2.0**-30) - delta
    hi = sr * (1 + 2.0**-30) + delta
    # no certain hit possible: nothing compares <= -inf (Dt itself can be negative); a negative finite threshold becomes
    # -inf as well (the kernels' integer minima order negative values the wrong way round)
    t_lo = lo * lo - eacc if (lo > 0 and lo * lo - eacc >= 0) else -np.inf
    t_hi = hi * hi + eacc
    lo_f = np.float32(t_lo)
    if float(lo_f) > t_lo:
        lo_f = np.nextafter(lo_f, np.float32(-np.inf))
    hi_f = np.float32(t_hi)
    if float(hi_f) < t_hi:
        hi_f = np.nextafter(hi_f, np.float32(np.inf))
    return float(lo_f), float(hi_f), t_hi < 30000.0


def split3(v):
    p1 = np.float16(np.float32(v))
    r1 = v - float(p1)
    p2 = np.float16(np.float32(r1))
    r2 = r1 - float(p2)
    return [p1, p2, np.float16(np.float32(r2))]


def mfma_accumulate(prod, model):
    """Dt for every row of `prod` (n, K) float64 exact products"""
    n, K = prod.shape
    if model == "reversed":
        dt = np.zeros(n, dtype=np.float32)
        for k in reversed(range(K)):
            dt = dt + prod[:, k].astype(np.float32)
        return dt
    acc = np.zeros(n, dtype=np.float64)
    for s0 in range(0, K, 16):            # one matrix instruction per 16 columns
        for g0 in (s0, s0 + 8):
            grp = prod[:, g0:g0 + 8]
            if model == "groups":
                part = grp.sum(axis=1)      # exact inside the group (8 products of binary16 values fit binary64)
            else:                           # "tree": binary32 roundings inside the group as well
                v = grp
                while v.shape[1] > 1:
                    v = (v[:, 0::2] + v[:, 1::2]).astype(np.float32).astype(np.float64)
                part = v[:, 0]
            acc = (acc + part).astype(np.float32).astype(np.float64)
    return acc.astype(np.float32)


def seq_dist2(a, b):
    acc = np.zeros(len(a))
    for k in range(a.shape[1]):
        diff = a[:, k] - b[k]
        acc = acc + diff * diff
    return acc


@pytest.mark.parametrize("model", ["reversed", "groups", "tree"])
@pytest.mark.parametrize("d,scale,offset,spread", [(2, 1.0, 0.0, 1.0), (5, 1e-5, 0.5, 1.0), (20, 1.0, 0.0, 0.3),
                                                   (50, 1.0, 0.0, 1.0), (50, 3e3, -7e4, 1.0), (90, 1.0, 10.0, 2.0)])
def test_thresholds_are_sound(d, scale, offset, spread, model):
    rs = np.random.RandomState(d)
    n, nq = 400, 60
    a = offset + scale * rs.normal(size=(n, d))
    b = offset + scale * spread * rs.normal(size=(nq, d))
    b[::3] = a[rs.randint(n, size=len(b[::3]))] + scale * 0.3 * rs.normal(size=(len(b[::3]), d))
    c = a.mean(axis=0)
    amax = np.abs(a - c).max()
    sigma = 2.0 ** -np.frexp(amax)[1]
    xa = sigma * (a - c)
    assert np.abs(xa).max() <= 1.0
    namax = np.sqrt((xa**2).sum(axis=1)).max() * (1 + 1e-12)
    ah = xa.astype(np.float32).astype(np.float16)
    K = (d + 6 + 15) // 16 * 16
    A = np.zeros((n, K), dtype=np.float16)
    A[:, :d] = ah
    na = (ah.astype(np.float64) ** 2).sum(axis=1)
    for i in range(n):
        A[i, d:d + 3] = split3(na[i])
    A[:, d + 3:d + 6] = 1.0
    checked = band = 0
    for j in range(nq):
        xb = sigma * (b[j] - c)
        bh = xb.astype(np.float32).astype(np.float16)
        B = np.zeros(K, dtype=np.float16)
        B[:d] = (-2.0 * bh.astype(np.float32)).astype(np.float16)
        assert np.array_equal(B[:d].astype(np.float64), -2.0 * bh.astype(np.float64))   # exact
        B[d:d + 3] = 1.0
        B[d + 3:d + 6] = split3((bh.astype(np.float64) ** 2).sum())
        # "MFMA": exact f16 x f16 products, float32 accumulation in the order of `model`
        prod = A.astype(np.float32) * B.astype(np.float32)
        assert np.array_equal(prod.astype(np.float64), A.astype(np.float64) * B.astype(np.float64))
        dt = mfma_accumulate(prod.astype(np.float64), model)
        s = seq_dist2(a, b[j])
        for r2 in (np.median(s), np.sort(s)[3], np.sort(s)[0] * (1 + 1e-9), s[rs.randint(n)]):
            lo, hi, ok = thresholds(sigma, namax, float((xb**2).sum()), r2, K)
            if not ok:
                continue
            assert not (dt[(s > r2)] <= lo).any(), "certain-hit threshold admitted a miss"
            assert not (dt[(s <= r2)] > hi).any(), "certain-miss threshold rejected a hit"
            checked += n
            band += int(((dt > lo) & (dt <= hi)).sum())
    assert checked > 0
    # the band must stay a small minority, otherwise the filter would not pay (not a correctness matter)
    assert band / checked < 0.2, band / checked

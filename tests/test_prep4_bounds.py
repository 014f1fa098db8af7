"""Soundness of the bounded per-proposal stage (csrc/mlf_prep4.hip) on the CPU: a numpy restatement of the device
arithmetic -- binary32 FMA chains (what v_mfma_f32_32x32x2_f32 computes, bit for bit a k-ascending fmaf chain), the
binary32 decision with its outward factors, filter_thresholds4 -- checked against the reference's binary64 results.

Properties:
  ellipsoid   sure_in  => einsum-order q <= enlarge ;  sure_out => q > enlarge   (mlfriends.pyx:882-912)
  chain       |y^ - y|_2 <= g (|y0| + |L|_F |delta|)                             (the eta / zeta of the kernel header)
  filter      with a query operand built from ANY point within zeta of the exact whitened point:
              Dt <= T_lo => s <= r2 ;  Dt > T_hi => s > r2                       (s = the reference's sequential distance)
"""
import numpy as np
import pytest

f32 = np.float32
UP = f32(1.0) + f32(2.0**-18)
DN = f32(1.0) - f32(2.0**-18)


def fma32_chain(start, a, b):
    """fl32(a_n b_n + ... fl32(a_1 b_1 + start)) for rows: start (m,), a (m, n), b (n,) all float32"""
    acc = start.astype(np.float64)
    for k in range(a.shape[1]):
        acc = (a[:, k].astype(np.float64) * np.float64(b[k]) + acc).astype(np.float32).astype(np.float64)
    return acc.astype(np.float32)


def g_chain(dp):
    g = (dp + 4) * 2.0**-24 * (1 + 2.0**-10) + 2.0**-40
    return np.nextafter(f32(g), f32(np.inf))


def up32(x):
    v = f32(x)
    return v if float(v) >= x else np.nextafter(v, f32(np.inf))


def dn32(x):
    v = f32(x)
    return v if float(v) <= x else np.nextafter(v, f32(-np.inf))


def einsum_order_q(delta, A):
    """numpy's c_einsum order for 'ij,jk,ik->i' on one row: one accumulator, j outer, (d_j A_jk) d_k"""
    acc = 0.0
    for j in range(len(delta)):
        for k in range(len(delta)):
            acc += (delta[j] * A[j, k]) * delta[k]
    return acc


def ellipsoid_decision(x, c_lay, L, y0_f32, consts, enlarge):
    """restatement of the H3 part of k_prep4 for one proposal; returns (sure_in, sure_out)"""
    d = len(x)
    dt = (x - c_lay).astype(np.float32)
    Lt32 = L.T.astype(np.float32)                     # (L^T)[i][k] = L[k][i]
    y = fma32_chain(y0_f32, Lt32, dt)
    qs = f32(0)
    for v in y:                                       # the device sums per lane and adds two halves: any order of n + 1 roundings
        qs = f32(np.float64(v) * np.float64(v) + np.float64(qs))
    dn2 = f32(0)
    for v in dt:
        dn2 = f32(np.float64(v) * np.float64(v) + np.float64(dn2))
    finite = bool(qs < f32(3e38) and dn2 < f32(3e38))
    sq = np.sqrt(qs)
    dnorm = np.sqrt(dn2) * UP + f32(2.0**-100)
    eta = consts["g"] * (consts["y0n"] + consts["lf"] * dnorm) * UP
    de = dnorm + consts["s0n"]
    eps = consts["eps_scale"] * (de * de) * UP
    hi = sq * UP + eta
    qhi = ((hi * hi) * UP + eps) * UP
    lo = (sq * DN - eta) * DN
    qlo = ((lo * lo) * DN - eps * UP) * DN
    return finite and bool(qhi < dn32(enlarge)), finite and bool(lo > 0) and bool(qlo > up32(enlarge))


@pytest.mark.parametrize("d,cond,shift", [(2, 1.0, 0.0), (5, 30.0, 1e-3), (20, 3.0, 0.0), (50, 1.5, 2e-3), (50, 40.0, 0.0),
                                          (64, 5.0, 1e-2)])
def test_ellipsoid_decisions_are_sound_and_chain_error_is_bounded(d, cond, shift):
    rs = np.random.RandomState(d * 7 + int(cond))
    dp = d + (d & 1)
    # a covariance with the requested condition number, its inverse as the ellipsoid matrix
    Q, _ = np.linalg.qr(rs.normal(size=(d, d)))
    ev = np.geomspace(1.0, cond**2, d) * 1e-3
    A = np.linalg.inv((Q * ev) @ Q.T)
    A = 0.5 * (A + A.T)
    L = np.linalg.cholesky(A)
    c_ell = 0.5 + 0.01 * rs.normal(size=d)
    c_lay = c_ell + shift * rs.normal(size=d)
    s0 = c_lay - c_ell
    y0 = L.T @ s0
    g = g_chain(dp)
    consts = dict(g=g, y0n=up32(np.linalg.norm(y0) * (1 + 1e-12)), lf=up32(np.linalg.norm(L) * (1 + 1e-12)),
                  s0n=up32(np.linalg.norm(s0) * (1 + 1e-12)),
                  eps_scale=up32(2.0**-34 * np.linalg.norm(A) * (1 + 1e-12)))
    enlarge = float(d) * 1.3
    # proposals on and around the boundary q = enlarge (the only place where the decision is delicate)
    z = rs.normal(size=(400, d))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    radii = np.sqrt(enlarge) * np.concatenate((1 + 3e-6 * rs.normal(size=200), rs.uniform(0.3, 1.7, size=200)))
    x = c_ell + (z * radii[:, None]) @ np.linalg.inv(L)            # |L^T (x - c_e)| = radius
    nin = nout = nband = 0
    y0_f32 = y0.astype(np.float32)
    for row in x:
        delta_e = row - c_ell
        q_ref = einsum_order_q(delta_e, A)
        sure_in, sure_out = ellipsoid_decision(row, c_lay, L, y0_f32, consts, enlarge)
        assert not (sure_in and sure_out)
        if sure_in:
            assert q_ref <= enlarge
            nin += 1
        elif sure_out:
            assert q_ref > enlarge
            nout += 1
        else:
            nband += 1
        # the error model itself, in the 2-norm
        dlt = row - c_lay
        y_hat = fma32_chain(y0_f32, L.T.astype(np.float32), dlt.astype(np.float32)).astype(np.float64)
        y_true = L.T @ delta_e
        bound = float(g) * (np.linalg.norm(y0) + np.linalg.norm(L) * np.linalg.norm(dlt))
        assert np.linalg.norm(y_hat - y_true) <= bound
    assert nin > 50 and nout > 50
    # the band is what k_ell_exact has to decide: it must stay a small share away from the boundary cluster
    assert nband < 260, (nin, nout, nband)


def thresholds4(namax, nb, zeta, sqrt_k, sr_lo, sr_hi):
    """float32 restatement of filter_thresholds4 (csrc/mlf_filter_dev.hpp)"""
    namax, nb, zeta, sqrt_k, sr_lo, sr_hi = (f32(v) for v in (namax, nb, zeta, sqrt_k, sr_lo, sr_hi))
    nbh = np.sqrt(nb) * UP
    nbn = ((nbh + sqrt_k * f32(2.0**-25)) * (f32(1) + f32(2.0**-10)) + zeta) * UP
    w0 = namax + nbn
    delta = (f32(2.0**-11) * (f32(1) + f32(2.0**-9)) * w0 + f32(2) * sqrt_k * f32(2.0**-24) + f32(2.0**-40) * w0
             + (f32(1) + f32(2.0**-10)) * zeta) * UP
    w = w0 + f32(2.0**-8)
    eacc = (f32(2.0**-15) * w * w + f32(2.0**-22) + f32(2.0**-18) * nb) * UP
    lo = (sr_lo * DN - delta) * DN
    hi = (sr_hi * UP + delta) * UP
    t_lo = (lo * lo) * DN - eacc * UP if lo > 0 else f32(-np.inf)
    t_hi = ((hi * hi) * UP + eacc) * UP
    return float(t_lo), float(t_hi), bool(t_hi < 30000)


def split3(v):
    p1 = np.float16(np.float32(v))
    r1 = np.float32(v) - np.float32(p1)
    p2 = np.float16(r1)
    r2 = r1 - np.float32(p2)
    return [p1, p2, np.float16(r2)]


def seq_dist2(a, b):
    acc = np.zeros(len(a))
    for k in range(a.shape[1]):
        diff = a[:, k] - b[k]
        acc = acc + diff * diff
    return acc


@pytest.mark.parametrize("d,scale,offset,zeta_rel", [(2, 1.0, 0.0, 1e-5), (5, 1e-5, 0.5, 3e-5), (20, 1.0, 0.0, 1e-4),
                                                     (50, 1.0, 0.0, 2e-5), (50, 3e3, -7e4, 1e-4), (62, 1.0, 10.0, 3e-4)])
def test_thresholds4_are_sound_for_every_point_within_zeta(d, scale, offset, zeta_rel):
    rs = np.random.RandomState(d + 100)
    n, nq = 300, 40
    a = offset + scale * rs.normal(size=(n, d))
    b = offset + scale * rs.normal(size=(nq, d))
    b[::3] = a[rs.randint(n, size=len(b[::3]))] + scale * 0.3 * rs.normal(size=(len(b[::3]), d))
    c = a.mean(axis=0)
    amax = np.abs(a - c).max()
    sigma = 2.0 ** -np.frexp(amax)[1]
    xa = sigma * (a - c)
    namax = np.sqrt((xa**2).sum(axis=1)).max() * (1 + 1e-12)
    ah = xa.astype(np.float32).astype(np.float16)
    dp = d + (d & 1)
    K = (dp + 6 + 15) // 16 * 16
    A = np.zeros((n, K), dtype=np.float16)
    A[:, :d] = ah
    na = (ah.astype(np.float64) ** 2).sum(axis=1)
    for i in range(n):
        A[i, dp:dp + 3] = [np.float16(np.float32(p)) for p in _split3_f64(na[i])]
    A[:, dp + 3:dp + 6] = 1.0
    checked = band = 0
    for j in range(nq):
        xb = sigma * (b[j] - c)                                  # exact scaled whitened point
        zeta = zeta_rel * (np.linalg.norm(xb) + namax)
        # the kernel's approximate point: anywhere within zeta -- pushed towards / away from a live point and random
        target = xa[rs.randint(n)]
        for direction in (target - xb, xb - target, rs.normal(size=d)):
            bq = (xb + 0.999 * zeta * direction / max(np.linalg.norm(direction), 1e-300)).astype(np.float32)
            m2 = (f32(-2) * bq).astype(np.float16)               # the chain delivers -2 bq; the operand is its binary16 rounding
            nb = f32(0)
            for v in m2.astype(np.float32):
                nb = f32(np.float64(v) * np.float64(v) + np.float64(nb))
            nb = f32(0.25) * nb
            B = np.zeros(K, dtype=np.float16)
            B[:d] = m2
            B[dp:dp + 3] = 1.0
            B[dp + 3:dp + 6] = split3(nb)
            prod = A.astype(np.float32) * B.astype(np.float32)
            dt = np.zeros(n, dtype=np.float32)
            for k in reversed(range(K)):
                dt = dt + prod[:, k]
            s = seq_dist2(a, b[j])
            for r2 in (np.median(s), np.sort(s)[3], np.sort(s)[0] * (1 + 1e-9), s[rs.randint(n)]):
                sr = sigma * np.sqrt(r2)
                lo, hi, ok = thresholds4(up32(namax), nb, up32(zeta), np.sqrt(f32(K)), dn32(sr * (1 - 2.0**-30)) * DN,
                                         up32(sr * (1 + 2.0**-30)) * UP)
                if not ok:
                    continue
                assert not (dt[s > r2] <= lo).any(), "certain-hit threshold admitted a miss"
                assert not (dt[s <= r2] > hi).any(), "certain-miss threshold rejected a hit"
                checked += n
                band += int(((dt > lo) & (dt <= hi)).sum())
    assert checked > 0
    assert band / checked < 0.35, band / checked


def _split3_f64(v):
    p1 = np.float16(np.float32(v))
    r1 = v - float(p1)
    p2 = np.float16(np.float32(r1))
    r2 = r1 - float(p2)
    return [p1, p2, np.float16(np.float32(r2))]


def test_whitening_chain_error_model():
    """|bq - sigma (T^T delta - c_s)|_2 <= g (|sigma c_s| + sigma |T|_F |delta|) for the binary32 chain, also with the
    -2 folded into the matrix (exact scaling)"""
    rs = np.random.RandomState(11)
    for d, cond in ((6, 2.0), (50, 1.3), (50, 15.0), (64, 6.0)):
        dp = d + (d & 1)
        Q, _ = np.linalg.qr(rs.normal(size=(d, d)))
        T = Q * np.geomspace(1.0, cond, d) * 20.0
        sigma = 2.0**-3
        cs = 0.01 * rs.normal(size=d)
        g = float(g_chain(dp))
        M32 = (T.T.astype(np.float32) * f32(-2 * sigma))          # rows = outputs
        start = (f32(2) * (sigma * cs).astype(np.float32))
        for _ in range(30):
            delta = 0.05 * rs.normal(size=d) * rs.uniform(0.1, 3)
            out = fma32_chain(start, M32, delta.astype(np.float32)).astype(np.float64) / -2.0
            exact = sigma * (T.T @ delta - cs)
            bound = g * (np.linalg.norm(sigma * cs) + sigma * np.linalg.norm(T) * np.linalg.norm(delta))
            assert np.linalg.norm(out - exact) <= bound

"""Soundness of the bounded per-proposal stage (csrc/mlf_prep4.hip) on the CPU: a numpy restatement of the device
arithmetic -- split-binary16 matrix chains with binary32 accumulation (v_mfma_f32_32x32x16_f16 as probed on gfx950:
two groups of 8 exact products per instruction, each added with one rounding; a rounded tree inside the groups is
covered as well), the binary32 decision with its outward factors, filter_thresholds4 -- checked against the
reference's binary64 results.

Properties:
  ellipsoid   sure_in  => einsum-order q <= enlarge ;  sure_out => q > enlarge   (mlfriends.pyx:882-912)
  chain       |y^ - y|_2 <= eta,  |bq - sigma (T^T delta - c_s)|_2 <= zeta        (the eta / zeta of the kernel header)
  filter      with a query operand built from ANY point within zeta of the exact whitened point:
              Dt <= T_lo => s <= r2 ;  Dt > T_hi => s > r2                       (s = the reference's sequential distance)
"""
import numpy as np
import pytest

f32 = np.float32
UP = f32(1.0) + f32(2.0**-18)
DN = f32(1.0) - f32(2.0**-18)


def pow2_scale(v, e):
    """power of two s with s v in [2^(e-1), 2^e)"""
    return 2.0 ** (e - np.frexp(v)[1])


def split16(x):
    """two binary16 pieces of binary32 values: hi = fl16(x), lo = fl16(x - hi) (the subtraction is exact)"""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def split16_matrix(m, scale):
    """host side: pieces of scale * m computed in binary64, and the representation error"""
    x = m * scale
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float64)).astype(np.float16)
    return hi, lo, x - hi.astype(np.float64) - lo.astype(np.float64)


def _group_sum(prod, tree):
    """sum of 8 exact products per row: exact (one rounding when added to C), or a depth-3 tree of binary32 additions"""
    if not tree:
        return prod.sum(axis=1)
    v = prod
    while v.shape[1] > 1:
        v = (v[:, 0::2] + v[:, 1::2]).astype(np.float32).astype(np.float64)
    return v[:, 0]


def mfma16_chain(start, mh, ml, hi, lo, tree=False, k_from=None):
    """v_mfma_f32_32x32x16_f16 chain as probed on gfx950 (scripts/probes/mfma16_acc_probe.hip): per instruction two
    groups of 8 exact products, added to C one after the other with a binary32 rounding each; per k-step the three
    partial products Mh hi, Mh lo, Ml hi.  start (m,) float32; mh, ml (m, K) float16; hi, lo (K,) float16."""
    acc = start.astype(np.float64)
    K = mh.shape[1]
    mh = mh.astype(np.float64)
    ml = ml.astype(np.float64)
    hi = hi.astype(np.float64)
    lo = lo.astype(np.float64)
    for s in range(0, K, 16):
        for a, b in ((mh, hi), (mh, lo), (ml, hi)):
            for g0 in (s, s + 8):
                part = _group_sum(a[:, g0:g0 + 8] * b[g0:g0 + 8], tree)
                acc = (acc + part).astype(np.float32).astype(np.float64)
    return acc.astype(np.float32)


def g_chain(dp):
    nsteps = 3 * ((dp + 15) // 16)
    g = (4.0 * nsteps + 8.0) * 2.0**-24 * (1 + 2.0**-8) + 2.0**-21.6 + 2.0**-21.9
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


def pad16(v, K):
    out = np.zeros(v.shape[:-1] + (K,), dtype=v.dtype)
    out[..., :v.shape[-1]] = v
    return out


class EllSetup:
    """host constants of region_prep4_setup / region_prep4_centres (csrc/mlf_api.hip) for the ellipsoid chain"""

    def __init__(self, L, A, c_ell, c_lay, amax):
        d = L.shape[0]
        dp = d + (d & 1)
        self.K = K = 16 * ((dp + 15) // 16)
        self.sx = pow2_scale(amax, 5)
        self.sl = pow2_scale(np.abs(L).max(), 8)
        lh, ll, err = split16_matrix(L.T, self.sl)
        self.lh, self.ll = pad16(lh, K), pad16(ll, K)
        s0 = c_lay - c_ell
        y0 = L.T @ s0
        self.y0 = y0
        self.y0f = (y0 * self.sl * self.sx).astype(np.float32)
        lf = np.linalg.norm(L)
        self.g = g_chain(dp)
        self.consts = dict(g=self.g, y0n=up32(np.linalg.norm(y0) * (1 + 1e-12)), lf=up32(lf * (1 + 1e-12)),
                           el=up32(np.linalg.norm(err) / self.sl * (1 + 1e-12)),
                           l_abs=up32(lf * np.sqrt(K) * 2.0**-25 / self.sx * (1 + 1e-12)),
                           s0n=up32(np.linalg.norm(s0) * (1 + 1e-12)),
                           eps_scale=up32(2.0**-34 * np.linalg.norm(A) * (1 + 1e-12)))
        self.c_lay = c_lay

    def operands(self, x):
        xp = x * self.sx - self.sx * self.c_lay          # binary64, one rounding (the scalings are exact)
        x32 = pad16(xp.astype(np.float32), self.K)
        hi, lo = split16(x32)
        dn2 = f32(0)
        for v in x32:
            dn2 = f32(np.float64(v) * np.float64(v) + np.float64(dn2))
        return hi, lo, f32(dn2 * f32(1.0 / self.sx**2))


def ellipsoid_decision(x, es, enlarge, tree):
    """restatement of the H3 part of k_prep4 for one proposal; returns (sure_in, sure_out, y^)"""
    c = es.consts
    hi, lo, dn2 = es.operands(x)
    y = mfma16_chain(es.y0f, es.lh, es.ll, hi, lo, tree)
    qs = f32(0)
    for v in y:                                       # the device sums per lane and adds two halves: any order of n + 1 roundings
        qs = f32(np.float64(v) * np.float64(v) + np.float64(qs))
    qs = f32(qs * f32(1.0 / (es.sl * es.sx) ** 2))
    finite = bool(qs < f32(3e38) and dn2 < f32(3e38))
    sq = np.sqrt(qs)
    dnorm = np.sqrt(dn2) * UP + f32(2.0**-100)
    eta = (c["g"] * (c["y0n"] + c["lf"] * dnorm) + c["el"] * dnorm + c["l_abs"]) * UP
    de = dnorm + c["s0n"]
    eps = c["eps_scale"] * (de * de) * UP
    hi_ = sq * UP + eta
    qhi = ((hi_ * hi_) * UP + eps) * UP
    lo_ = (sq * DN - eta) * DN
    qlo = ((lo_ * lo_) * DN - eps * UP) * DN
    return (finite and bool(qhi < dn32(enlarge)), finite and bool(lo_ > 0) and bool(qlo > up32(enlarge)),
            y.astype(np.float64) / (es.sl * es.sx), float(eta))


@pytest.mark.parametrize("d,cond,shift", [(2, 1.0, 0.0), (5, 30.0, 1e-3), (20, 3.0, 0.0), (50, 1.5, 2e-3), (50, 40.0, 0.0),
                                          (64, 5.0, 1e-2)])
def test_ellipsoid_decisions_are_sound_and_chain_error_is_bounded(d, cond, shift):
    rs = np.random.RandomState(d * 7 + int(cond))
    # a covariance with the requested condition number, its inverse as the ellipsoid matrix
    Q, _ = np.linalg.qr(rs.normal(size=(d, d)))
    ev = np.geomspace(1.0, cond**2, d) * 1e-3
    A = np.linalg.inv((Q * ev) @ Q.T)
    A = 0.5 * (A + A.T)
    L = np.linalg.cholesky(A)
    c_ell = 0.5 + 0.01 * rs.normal(size=d)
    c_lay = c_ell + shift * rs.normal(size=d)
    enlarge = float(d) * 1.3
    # proposals on and around the boundary q = enlarge (the only place where the decision is delicate)
    z = rs.normal(size=(300, d))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    radii = np.sqrt(enlarge) * np.concatenate((1 + 3e-5 * rs.normal(size=150), rs.uniform(0.3, 1.7, size=150)))
    x = c_ell + (z * radii[:, None]) @ np.linalg.inv(L)            # |L^T (x - c_e)| = radius
    es = EllSetup(L, A, c_ell, c_lay, np.abs(x - c_lay).max() * 0.7)
    nin = nout = nband = 0
    for i, row in enumerate(x):
        delta_e = row - c_ell
        q_ref = einsum_order_q(delta_e, A)
        sure_in, sure_out, y_hat, eta = ellipsoid_decision(row, es, enlarge, tree=bool(i & 1))
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
        assert np.linalg.norm(y_hat - L.T @ delta_e) <= eta
    assert nin > 40 and nout > 40
    # the band is what the binary64 test has to decide: it must stay a small share away from the boundary cluster
    assert nband < 200, (nin, nout, nband)


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
    if t_lo < 0:
        t_lo = f32(-np.inf)      # never negative and finite (filter_thresholds4)
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
    """|bq - sigma (T^T delta - c_s)|_2 <= zeta = sigma [g (|c_s| + |T|_F |delta|) + |E_T|_F |delta| / s_T + |T|_F sqrt(K) 2^-25 / s_x]
    for the split-binary16 chain; the epilogue v = fma(acc, -2 sigma / (s_T s_x), 2 sigma c_s) adds one rounding of a
    value the binary16 conversion behind it rounds 2^13 times coarser (covered by Delta's binary16 term)"""
    rs = np.random.RandomState(11)
    for d, cond in ((6, 2.0), (50, 1.3), (50, 15.0), (64, 6.0)):
        dp = d + (d & 1)
        K = 16 * ((dp + 15) // 16)
        Q, _ = np.linalg.qr(rs.normal(size=(d, d)))
        T = Q * np.geomspace(1.0, cond, d) * 20.0
        sigma = 2.0**-3
        cs = 0.01 * rs.normal(size=d)
        g = float(g_chain(dp))
        st = pow2_scale(np.abs(T).max(), 8)
        sx = pow2_scale(0.12, 5)
        th, tl, err = split16_matrix(T.T, st)              # rows = outputs
        th, tl = pad16(th, K), pad16(tl, K)
        tf = np.linalg.norm(T)
        zt = g * tf + np.linalg.norm(err) / st
        zt_abs = tf * np.sqrt(K) * 2.0**-25 / sx
        c_lay = 0.5 + 0.01 * rs.normal(size=d)
        for it in range(30):
            delta = 0.05 * rs.normal(size=d) * rs.uniform(0.1, 3)
            x = c_lay + delta
            xp = x * sx - sx * c_lay
            x32 = pad16(xp.astype(np.float32), K)
            hi, lo = split16(x32)
            acc = mfma16_chain(np.zeros(d, dtype=np.float32), th, tl, hi, lo, tree=bool(it & 1)).astype(np.float64)
            kappa = -2.0 * sigma / (st * sx)
            v = (acc * kappa + 2.0 * sigma * cs).astype(np.float32).astype(np.float64)
            out = v / -2.0
            dtrue = x - c_lay
            exact = sigma * (T.T @ dtrue - cs)
            dnorm = np.linalg.norm(x32.astype(np.float64)) / sx * (1 + 2.0**-18)
            zeta = sigma * (zt * dnorm + g * np.linalg.norm(cs) + zt_abs)
            assert np.linalg.norm(out - exact) <= zeta * (1 + 2.0**-20), (d, cond, it)


class SameFormSetup:
    """host constants of the "same quadratic form" variant (region_prep4_setup, csrc/mlf_api.hip: p4c_same): the ellipsoid
    test reads delta^T A delta = |T^T delta|^2 + delta^T E delta off the whitening chain; None when the residue is refused"""

    def __init__(self, T, A, c, amax):
        d = T.shape[0]
        dp = d + (d & 1)
        self.K = K = 16 * ((dp + 15) // 16)
        self.sx = pow2_scale(amax, 5)
        self.st = pow2_scale(np.abs(T).max(), 8)
        th, tl, err = split16_matrix(T.T, self.st)         # rows = outputs
        self.th, self.tl = pad16(th, K), pad16(tl, K)
        tf = np.linalg.norm(T)
        afro = np.linalg.norm(A)
        E = 0.5 * (A + A.T) - T @ T.T
        e_bound = (np.linalg.norm(E) + (d + 4.0) * 2.0**-52 * (tf * tf + afro)) * (1 + 1e-12)
        self.accepted = bool(e_bound <= 2.0**-34 * afro)
        self.g = g_chain(dp)
        self.consts = dict(g=self.g, y0n=f32(0), lf=up32(tf * (1 + 1e-12)), el=up32(np.linalg.norm(err) / self.st * (1 + 1e-12)),
                           l_abs=up32(tf * np.sqrt(K) * 2.0**-25 / self.sx * (1 + 1e-12)), s0n=f32(0),
                           eps_scale=up32((2.0**-34 * afro + e_bound) * (1 + 1e-12)))
        self.c_lay = c
        self.sl = self.st                                    # (inv_sl_sx = inv_st_sx)
        self.lh, self.ll = self.th, self.tl
        self.y0f = np.zeros(d, dtype=np.float32)

    operands = EllSetup.operands


@pytest.mark.parametrize("d,cond,perturb", [(2, 1.0, 0.0), (5, 30.0, 0.0), (20, 3.0, 2.0**-37), (50, 1.5, 0.0), (50, 40.0, 2.0**-36),
                                            (50, 1.5, 2.0**-33), (56, 5.0, 0.0)])
def test_same_form_ellipsoid_decisions_are_sound(d, cond, perturb):
    """AffineLayer with one cluster: the layer's T = eigvec * eigval^-0.5 and the ellipsoid's A = inv(cov) come from the same
    covariance (mlfriends.pyx:684-706, 447-452 + :1040-1048), so A = T T^T up to LAPACK rounding -- optionally with a residue
    of the size the acceptance rule still admits.  ellipsoid_decision() restates the kernel's decision with the whitening
    chain in the ellipsoid chain's place (k_prep_sweep<.., SQ = true>)."""
    rs = np.random.RandomState(d * 11 + int(cond))
    Q, _ = np.linalg.qr(rs.normal(size=(d, d)))
    ev = np.geomspace(1.0, cond**2, d) * 1e-3
    cov = (Q * ev) @ Q.T
    cov = 0.5 * (cov + cov.T)
    eigval, eigvec = np.linalg.eigh(cov)
    T = eigvec * eigval**-0.5
    A = np.linalg.inv(cov)
    if perturb:
        P = rs.normal(size=(d, d))
        A = A + perturb * np.linalg.norm(A) * (P + P.T) / np.linalg.norm(P + P.T)
    c = 0.5 + 0.01 * rs.normal(size=d)
    enlarge = float(d) * 1.3
    z = rs.normal(size=(300, d))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    radii = np.sqrt(enlarge) * np.concatenate((1 + 3e-5 * rs.normal(size=150), rs.uniform(0.3, 1.7, size=150)))
    x = c + (z * radii[:, None]) @ np.linalg.inv(T)                 # |T^T (x - c)| = radius
    es = SameFormSetup(T, A, c, np.abs(x - c).max() * 0.7)
    assert es.accepted == (perturb < 2.0**-34)
    if not es.accepted:
        return
    nin = nout = nband = 0
    for i, row in enumerate(x):
        delta = row - c
        q_ref = einsum_order_q(delta, A)
        sure_in, sure_out, t_hat, eta = ellipsoid_decision(row, es, enlarge, tree=bool(i & 1))
        assert not (sure_in and sure_out)
        if sure_in:
            assert q_ref <= enlarge
            nin += 1
        elif sure_out:
            assert q_ref > enlarge
            nout += 1
        else:
            nband += 1
        assert np.linalg.norm(t_hat - T.T @ delta) <= eta
    assert nin > 40 and nout > 40
    assert nband < 200, (nin, nout, nband)

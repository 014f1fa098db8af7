#!/usr/bin/env python3
"""Ground-truth log-evidence of the eggbox problem (reference examples/testeggbox.py:9-14):

    loglike(z) = (2 + prod_i cos(z_i / 2))**5,   z = 10 pi x,   x ~ U[0, 1]^d
    Z = E_x[exp(loglike)]

for the dimensionalities the configurations use (BASELINE.json C3: d = 10; d = 2 as a cross-check against the literature
value 235.88), written to tests/golden/g15_eggbox_logz.json.  No reference code is involved: this is numerics on the
likelihood's DEFINITION (numpy only, any machine).

Uniform Monte Carlo is hopeless at d = 10 (the peaks fill ~5e-15 of the cube), so the integral is first FOLDED, exactly:
with u = z / 2 in [0, 5 pi] every axis splits into 10 cells of length pi / 2 on which |cos u| = cos w, w in [0, pi / 2]
(a measure-preserving reflection / shift), five of them with cos u > 0 and five with cos u < 0.  The sign of the product is
+ for exactly half of the 10^d cell combinations ((5 - 5)^d = 0), hence

    Z = 1/2 * (E_plus + E_minus),   E_pm = E_{w ~ U[0, pi/2]^d} exp((2 +- prod_i cos w_i)**5).

E_minus <= exp(32) is negligible next to E_plus ~ exp(210) but is estimated anyway (plain Monte Carlo).  E_plus has ONE peak,
at w = 0, close to a Gaussian of variance 1/405 per axis ((2 + c)^5 ~ 243 - 405 (1 - c), 1 - c ~ sum w^2 / 2): importance
sampling with a defensive mixture q = 0.9 * prod half-normal(sigma = 1.25 / sqrt(405), truncated to [0, pi/2]) + 0.1 * uniform,
whose weights are bounded, gives an unbiased estimate with a relative standard error of a few 1e-4 from 2e7 draws.
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HALF_PI = 0.5 * math.pi


def log_e_plus(d, n, rs, chunk=1000000):
    sigma = 1.25 / math.sqrt(405.0)
    mix = 0.9
    # normalisation of the truncated half-normal on [0, pi/2]
    from math import erf, sqrt
    znorm = sigma * sqrt(HALF_PI) * erf(HALF_PI / (sigma * sqrt(2.0)))        # integral of exp(-w^2 / (2 sigma^2)) over [0, pi/2]
    log_unif = -d * math.log(HALF_PI)
    shift = 243.0                      # exp(loglike - 243): the integrand's maximum is 1
    s1 = s2 = 0.0
    done = 0
    wmax = 0.0
    while done < n:
        m = min(chunk, n - done)
        from_gauss = rs.uniform(size=m) < mix
        w = np.abs(rs.normal(scale=sigma, size=(m, d)))
        bad = w > HALF_PI                      # truncation by rejection (never happens at sigma = 0.06, kept for exactness)
        while bad.any():
            w[bad] = np.abs(rs.normal(scale=sigma, size=int(bad.sum())))
            bad = w > HALF_PI
        wu = rs.uniform(0.0, HALF_PI, size=(m, d))
        w = np.where(from_gauss[:, None], w, wu)
        logq_g = -0.5 * (w * w).sum(axis=1) / sigma**2 - d * math.log(znorm)
        logq = np.logaddexp(math.log(mix) + logq_g, math.log(1.0 - mix) + log_unif)
        logf = (2.0 + np.cos(w).prod(axis=1)) ** 5 - shift + log_unif      # integrand times the uniform density of w
        wt = np.exp(logf - logq)
        s1 += float(wt.sum())
        s2 += float((wt * wt).sum())
        wmax = max(wmax, float(wt.max()))
        done += m
    mean = s1 / n
    var = max(s2 / n - mean * mean, 0.0) / n
    ess = s1 * s1 / s2
    return shift + math.log(mean), math.sqrt(var) / mean, ess, wmax / mean


def log_e_minus(d, n, rs):
    w = rs.uniform(0.0, HALF_PI, size=(n, d))
    v = (2.0 - np.cos(w).prod(axis=1)) ** 5
    m = float(v.max())
    return m + math.log(float(np.exp(v - m).mean()))


def laplace(d):
    """Gaussian approximation around the single folded peak: Z ~ 1/2 * e^243 * (2 / pi)^d * (sqrt(2 pi / 405) / 2)^d."""
    return math.log(0.5) + 243.0 + d * (math.log(2.0 / math.pi) + 0.5 * math.log(2.0 * math.pi / 405.0) - math.log(2.0))


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
    out = {"what": "log E_x exp((2 + prod cos(5 pi x_i))**5), x ~ U[0,1]^d: folded onto [0, pi/2]^d (exact), importance sampling with a "
                   "defensive Gaussian + uniform mixture (tests/golden/make_eggbox_logz.py)", "draws": n, "cases": {}}
    for d in (2, 10):
        rs = np.random.RandomState(1000 + d)
        lp, rel, ess, wrel = log_e_plus(d, n, rs)
        lm = log_e_minus(d, 1000000, rs)
        logz = math.log(0.5) + np.logaddexp(lp, lm)
        out["cases"][str(d)] = {"logz": float(logz), "stderr": float(rel), "effective_sample_size": ess,
                                "largest_weight_over_mean": wrel, "log_E_plus": lp, "log_E_minus": lm, "laplace_logz": laplace(d)}
        print(d, out["cases"][str(d)], flush=True)
    with open(os.path.join(HERE, "g15_eggbox_logz.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()

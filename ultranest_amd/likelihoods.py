"""Vectorized-likelihood batch callbacks backed by HIP kernels.

These plug into the reference's ``ReactiveNestedSampler(..., vectorized=True)`` callback surface
(reference integrator.py:1205-1209, 1270-1283, 1789-1804): ``loglike(params[n, num_params]) ->
logl[n]`` and ``transform(u[n, x_dim]) -> p[n, num_params]``, float64, input left unmodified.
The kernels follow the (params, d, n, like) convention of reference languages/c/mylib.c:33.

Definitions restated from the reference's example scripts:
  Gaussian    docs/gauss.py:25-27 (centres docs/gauss.py:21; examples/testgauss.py:10,56)
  eggbox      examples/testeggbox.py:9-14
  eggbox (2)  examples/test_PopSliceSampler.py:69-74
  Rosenbrock  examples/testrosenbrock.py:10-16
"""
import numpy as np

from . import _lib
from ._lib import check, f64, ptr


def _batch(params):
    p = f64(params)
    if p.ndim != 2:
        raise ValueError("vectorized likelihoods take a 2-d (n, num_params) array")
    return p


class GaussLikelihood(object):
    """``-0.5*sum(((theta-centers)/sigma)**2, axis=1) - 0.5*log(2*pi*sigma**2)*ndim``."""

    def __init__(self, centers, sigma, ndim):
        self.ndim = int(ndim)
        self.centers = f64(np.broadcast_to(centers, (self.ndim,)))
        self.sigma = float(sigma)

    @classmethod
    def docs_gauss(cls, ndim, sigma=0.1):
        """centres of reference docs/gauss.py:18-21"""
        width = max(0, 1 - 5 * sigma)
        return cls((np.sin(np.arange(ndim) / 2.) * width + 1.) / 2., sigma, ndim)

    @property
    def device_spec(self):
        """(kind, aux, sigma) of mlf_loglike_dev: lets device-resident callers evaluate in place"""
        return 0, self.centers, self.sigma

    def __call__(self, theta):
        p = _batch(theta)
        if p.shape[1] != self.ndim:
            raise ValueError("expected %d parameters" % self.ndim)
        out = np.empty(p.shape[0])
        check(_lib.lib().mlf_loglike_gauss(ptr(p), p.shape[1], p.shape[0], ptr(self.centers), self.sigma, ptr(out)))
        return out


def eggbox_loglike(z):
    """``(2 + prod(cos(z/2), axis=1))**5``"""
    p = _batch(z)
    out = np.empty(p.shape[0])
    check(_lib.lib().mlf_loglike_eggbox(ptr(p), p.shape[1], p.shape[0], ptr(out)))
    return out


def eggbox_transform(x):
    """``x * 10 * pi`` (elementwise; stays on the host, it is one multiply per value)"""
    return np.asarray(x) * 10 * np.pi


def eggbox2_loglike(theta):
    """``prod(cos(theta), axis=1)**2``"""
    p = _batch(theta)
    out = np.empty(p.shape[0])
    check(_lib.lib().mlf_loglike_eggbox2(ptr(p), p.shape[1], p.shape[0], ptr(out)))
    return out


def rosenbrock_loglike(theta):
    """``-2 * sum(100*(b - a**2)**2 + (1 - a)**2, axis=1)`` over consecutive pairs (a, b)."""
    p = _batch(theta)
    out = np.empty(p.shape[0])
    check(_lib.lib().mlf_loglike_rosenbrock(ptr(p), p.shape[1], p.shape[0], ptr(out)))
    return out


def rosenbrock_transform(u):
    """``u * 20 - 10``"""
    return np.asarray(u) * 20 - 10


def identity_transform(u):
    """``p = u``"""
    return np.array(u, dtype=float)


# (kind, aux, sigma) of mlf_loglike_dev and (tkind, a, b) of mlf_walkers_finish_dev: samplers that
# keep their points on the device (ultranest_amd.popstepsampler) evaluate these in place
eggbox_loglike.device_spec = (1, None, 0.0)
eggbox2_loglike.device_spec = (2, None, 0.0)
rosenbrock_loglike.device_spec = (3, None, 0.0)
identity_transform.device_spec = (0, 0.0, 0.0)
eggbox_transform.device_spec = (2, 10.0, np.pi)       # (x * 10) * pi
rosenbrock_transform.device_spec = (1, 20.0, -10.0)   # x * 20 - 10

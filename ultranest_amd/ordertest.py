"""Insertion-order test of a nested sampling run: are the ranks at which new live points enter uniformly
distributed?  Names and behaviour of the reference's ``ultranest.ordertest`` (reference ultranest/ordertest.py:
``infinite_U_zscore`` :31-47, ``UniformOrderAccumulator`` :49-104; Buchner 2023, section 4.5.2).

A rank r out of N contributes (r + 1/2) / N, which under uniform insertion has mean 1/2 and variance 1/12 (for
large N); the z-score is the standardised sum.  ``netiter.MultiCounter`` keeps the same two numbers inside the
compiled counter (csrc/mlf_netiter.hip); this module is the stand-alone form the driver uses for its rolling
test (integrator.py:2609).  Host code, no GPU.
"""
import numpy as np

__all__ = ['infinite_U_zscore', 'UniformOrderAccumulator']

_UNIT_VARIANCE = 1.0 / 12.0


def _standardise(total, count):
    """z-score of `total` = sum of `count` terms with mean 1/2 and variance 1/12 each."""
    if count == 0:
        return 0.0
    return (total - 0.5 * count) / (count * _UNIT_VARIANCE) ** 0.5


def infinite_U_zscore(sample, B):
    """z-score of the Mann-Whitney-Wilcoxon U statistic of the integer `sample` against the uniform distribution
    on 0 .. `B`."""
    sample = np.asarray(sample)
    return _standardise((sample + 0.5).sum() / B, len(sample))


class UniformOrderAccumulator(object):
    """Running form of the same statistic: ``add(order, N)`` accumulates one rank, ``zscore`` is the current
    value, ``len()`` the number of ranks since the last ``reset()``.  Attributes ``N`` (count) and ``U`` (sum)
    as in the reference."""

    __slots__ = ("N", "U")

    def __init__(self):
        self.reset()

    def reset(self):
        self.N, self.U = 0, 0.0

    def __len__(self):
        return self.N

    def add(self, order, N):
        if order < 0 or order > N:
            raise ValueError("order %d out of %d invalid" % (order, N))
        self.N += 1
        self.U += (order + 0.5) / N

    @property
    def zscore(self):
        return _standardise(self.U, self.N)

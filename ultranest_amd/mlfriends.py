"""Import-compatible namespace: every name the reference's driver, step samplers and tests import
from ``ultranest.mlfriends`` (reference integrator.py:28-30 and SURVEY.md 8b) is available here
with the same meaning, so ``from ultranest_amd.mlfriends import MLFriends, AffineLayer, ...`` is a
drop-in for ``from ultranest.mlfriends import ...`` on the region path."""
from .kernels import (compute_mean_pair_distance, count_nearby, find_nearby, int_dtype,  # noqa: F401
                      subtract_nearby)
from .layers import (AffineLayer, LocalAffineLayer, MaxPrincipleGapAffineLayer, ScalingLayer,  # noqa: F401
                     update_clusters)
from .regions import (MLFriends, RobustEllipsoidRegion, SimpleRegion, WrappingEllipsoid,  # noqa: F401
                      _inside_ellipsoid, bounding_ellipsoid, make_eigvals_positive, vol_prefactor)

__all__ = [
    "ScalingLayer", "AffineLayer", "MaxPrincipleGapAffineLayer", "LocalAffineLayer",
    "MLFriends", "RobustEllipsoidRegion", "SimpleRegion", "WrappingEllipsoid",
    "find_nearby", "subtract_nearby", "update_clusters", "compute_mean_pair_distance",
    "bounding_ellipsoid", "make_eigvals_positive", "vol_prefactor", "_inside_ellipsoid", "int_dtype",
]

"""ultranest_amd -- MI355X (gfx950) implementation of UltraNest's MLFriends hot path.

Drop-in for ``ultranest.mlfriends`` on that path only: region construction
(bootstrapped ``compute_maxradiussq``), the membership test (``find_nearby`` / ``inside``) and
the vectorized-likelihood batch call.  Compute runs in hand-written HIP kernels behind the C ABI
of ``include/mlfriends_hip.h``; there is no CPU fallback.
"""
__version__ = "0.1.0"

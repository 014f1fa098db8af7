"""TEST INFRASTRUCTURE -- numpy restatement of the device proposal generator's random stream.

Philox-4x32-10 (Salmon, Moraes, Dror & Shaw, "Parallel random numbers: as easy as 1, 2, 3",
SC'11; the Random123 reference constants) plus the word -> double mapping and the counter layout
of ultranest_amd/csrc/mlf_sample.hip.  Pinned by the Random123 known-answer vectors in
tests/test_philox.py.  Only tests may import this module.

The reference (UltraNest) draws proposals from numpy's global MT19937 stream
(mlfriends.pyx:1096-1160); the device generator is an opt-in replacement with the same
distributions, so parity for it is (a) bit-exact against this restatement and (b) statistical
against the reference's distributions.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (n, 4) uint32 counters, key: (2,) uint32 -> (n, 4) uint32 output blocks."""
    c = [np.asarray(ctr)[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n1 = p1 & MASK32
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        n3 = p0 & MASK32
        c = [n0, n1, n2, n3]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def blocks(seed, stream, counters):
    """Output blocks for 64-bit `counters` on `stream` (counter words = (lo, hi, stream, 0))."""
    counters = np.asarray(counters, dtype=np.uint64)
    ctr = np.zeros((len(counters), 4), dtype=np.uint32)
    ctr[:, 0] = (counters & MASK32).astype(np.uint32)
    ctr[:, 1] = (counters >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = stream
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))


def u01(hi, lo):
    """53-bit uniform strictly inside (0, 1): ((hi>>5) * 2^26 + (lo>>6) + 0.5) * 2^-53."""
    m = ((hi.astype(np.uint64) >> np.uint64(5)) << np.uint64(26)) | (lo.astype(np.uint64) >> np.uint64(6))
    return (m.astype(np.float64) + 0.5) * 2.0**-53


def cube_points(seed, offset, nsamples, ndim):
    """The (nsamples, ndim) unit-cube batch of method 0 and the next counter offset."""
    nelem = nsamples * ndim
    nb = (nelem + 1) // 2
    w = blocks(seed, 0, np.uint64(offset) + np.arange(nb, dtype=np.uint64))
    flat = np.empty(2 * nb)
    flat[0::2] = u01(w[:, 0], w[:, 1])
    flat[1::2] = u01(w[:, 2], w[:, 3])
    return flat[:nelem].reshape(nsamples, ndim), offset + nb


def ball_points(seed, offset, nsamples, ndim, enlarge):
    """Method 1's draw before the axes rotation: uniform in the ball of radius sqrt(enlarge)
    (Box-Muller pairs + one radial uniform per point).  libm on the host and the device's
    log/sin/cos/pow differ by a few ULP, so this is compared with a tolerance."""
    npairs = (ndim + 1) // 2
    per = npairs + 1
    base = np.uint64(offset) + np.arange(nsamples, dtype=np.uint64) * np.uint64(per)
    z = np.empty((nsamples, 2 * npairs))
    for j in range(npairs):
        w = blocks(seed, 1, base + np.uint64(j))
        r = np.sqrt(-2.0 * np.log(u01(w[:, 0], w[:, 1])))
        ang = 2.0 * np.pi * u01(w[:, 2], w[:, 3])
        z[:, 2 * j] = r * np.cos(ang)
        z[:, 2 * j + 1] = r * np.sin(ang)
    z = z[:, :ndim]
    w = blocks(seed, 1, base + np.uint64(npairs))
    scale = np.sqrt(enlarge) * u01(w[:, 0], w[:, 1])**(1.0 / ndim) / np.sqrt((z**2).sum(axis=1))
    return z * scale[:, None], offset + nsamples * per


def tbox_points(seed, offset, nsamples, ndim, lo, hi, pad):
    """Method 2's whitened-space batch: low + (high - low) * U with low = lo - pad, high = hi + pad
    (stream 5, two doubles per block) and the next counter offset."""
    nelem = nsamples * ndim
    nb = (nelem + 1) // 2
    w = blocks(seed, 5, np.uint64(offset) + np.arange(nb, dtype=np.uint64))
    flat = np.empty(2 * nb)
    flat[0::2] = u01(w[:, 0], w[:, 1])
    flat[1::2] = u01(w[:, 2], w[:, 3])
    U = flat[:nelem].reshape(nsamples, ndim)
    low, high = np.asarray(lo) - pad, np.asarray(hi) + pad
    return low + (high - low) * U, offset + nb


def around_points(seed, offset, nsamples, ndim, unormed, r2):
    """Method 3's draw: (t-space proposals, thinning uniforms, chosen live indices, next offset);
    stream 6, (npairs + 2) blocks per proposal.  libm tolerance like ball_points."""
    npairs = (ndim + 1) // 2
    per = npairs + 2
    base = np.uint64(offset) + np.arange(nsamples, dtype=np.uint64) * np.uint64(per)
    w0 = blocks(seed, 6, base)
    which = ((w0[:, 0].astype(np.uint64) * np.uint64(len(unormed))) >> np.uint64(32)).astype(np.int64)
    radial = u01(w0[:, 2], w0[:, 3])
    w1 = blocks(seed, 6, base + np.uint64(1))
    thin = u01(w1[:, 0], w1[:, 1])
    z = np.empty((nsamples, 2 * npairs))
    for j in range(npairs):
        w = blocks(seed, 6, base + np.uint64(2 + j))
        r = np.sqrt(-2.0 * np.log(u01(w[:, 0], w[:, 1])))
        ang = 2.0 * np.pi * u01(w[:, 2], w[:, 3])
        z[:, 2 * j] = r * np.cos(ang)
        z[:, 2 * j + 1] = r * np.sin(ang)
    z = z[:, :ndim]
    f = radial**(1.0 / ndim) / np.sqrt((z**2).sum(axis=1)) * np.sqrt(r2)
    return unormed[which] + z * f[:, None], thin, which, offset + nsamples * per

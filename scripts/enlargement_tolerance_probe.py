import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import test_device_rebuild as T
for n, d, ln, blobs in [(400, 5, "LocalAffineLayer", 1), (1000, 10, "AffineLayer", 1), (600, 4, "AffineLayer", 2), (2000, 20, "LocalAffineLayer", 1), (4000, 50, "LocalAffineLayer", 1)]:
    u2, ((a, *_), (b, *_)) = T._pair(n, d, 3, ln, blobs)
    ra, rb = a.region, b.region
    print(n, d, ln, 'enlarge rel diff', abs(ra.enlarge - rb.enlarge) / ra.enlarge, 'r', ra.maxradiussq, rb.maxradiussq)
import ultranest_amd.mlfriends as M
from ultranest_amd import device_rebuild
g = np.load('tests/golden/g456_region.npz')
u = np.ascontiguousarray(g["g5_u"]); n, d = u.shape
for lname, cls in (("affine", M.AffineLayer), ("local", M.LocalAffineLayer)):
    layer = cls(); layer.optimize(u, u)
    rb = device_rebuild.DeviceRebuild()
    key = "g5_%s1_" % lname
    np.random.seed(77 + 2)
    nxt, region, contains = rb.next_region(u, layer, float(g[key + "r_f"][0]), 30)
    r_next, f_next = (float(v) for v in g["g5_%s2_r_f" % lname])
    print('golden', lname, 'enlarge rel diff vs reference', abs(region.enlarge - f_next) / f_next)

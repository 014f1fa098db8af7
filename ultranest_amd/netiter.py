"""Tree exploration and bootstrapped evidence counters with the API of the reference's
``ultranest.netiter`` (reference ultranest/netiter.py: TreeNode :34-60, BreadthFirstIterator :63-162,
MultiCounter :571-855).

The driver runs ``explorer.next_node()`` -> ``main_iterator.passing_node(...)`` ->
``explorer.expand_children_of(...)`` once per nested-sampling iteration (integrator.py:2650-2832).
``MultiCounter.passing_node`` -- the evidence / information / remainder update of all bootstrap
counters and the insertion-order U test -- is compiled host code here (csrc/mlf_netiter.hip, one
call instead of ~40 small numpy operations); the counter's attributes keep the reference's names.
No GPU is involved in this module.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, ptr


class TreeNode(object):
    """Tree node: ordering value (log-likelihood), point id, children (reference :34-60)."""

    def __init__(self, value=None, id=None, children=None):
        self.value = value
        self.id = id
        self.children = children or []

    def __str__(self, indent=0):
        return ' ' * indent + '- Node: %s\n' % self.value + '\n'.join(
            [c.__str__(indent=indent + 2) for c in self.children])

    def __lt__(self, other):
        return self.value < other.value


class BreadthFirstIterator(object):
    """Explores the tree in order of node value; the nodes crossed "in parallel" are the live
    points (reference :63-162).  Same attributes: ``active_nodes``, ``active_root_ids``,
    ``active_node_values``, ``active_node_ids``."""

    def __init__(self, roots):
        self.roots = roots
        self.reset()

    def reset(self):
        self.active_nodes = list(self.roots)
        self.active_root_ids = np.arange(len(self.active_nodes))
        self.active_node_values = np.array([n.value for n in self.active_nodes])
        self.active_node_ids = np.array([n.id for n in self.active_nodes])

    def next_node(self):
        """``(rootid, node, (active_nodes, active_root_ids, active_node_values, active_node_ids))``
        of the lowest active node (left in the active set), or None when the tree is exhausted."""
        if self.active_nodes == []:
            return None
        self.next_index = i = np.argmin(self.active_node_values)
        return self.active_root_ids[i], self.active_nodes[i], (
            self.active_nodes, self.active_root_ids, self.active_node_values, self.active_node_ids)

    def _remove_current(self):
        i = self.next_index
        self.active_nodes.pop(i)
        self.active_node_values = np.delete(self.active_node_values, i)
        self.active_root_ids = np.delete(self.active_root_ids, i)
        self.active_node_ids = np.delete(self.active_node_ids, i)

    def drop_next_node(self):
        """Forget the current node (reference :110-120)."""
        self._remove_current()

    def expand_children_of(self, rootid, node):
        """Replace the current node by its children: an only child takes its slot, several are
        appended at the end (reference :122-161)."""
        kids = node.children
        if len(kids) == 1:
            i = self.next_index
            self.active_nodes[i] = kids[0]
            self.active_node_values[i] = kids[0].value
            self.active_root_ids[i] = rootid
            self.active_node_ids[i] = kids[0].id
            return
        self._remove_current()
        if kids:
            self.active_nodes += kids
            self.active_node_values = np.concatenate((self.active_node_values, [c.value for c in kids]))
            self.active_root_ids = np.concatenate((self.active_root_ids, [rootid for c in kids]))
            self.active_node_ids = np.concatenate((self.active_node_ids, [c.id for c in kids]))


class _OrderAccumulatorView(object):
    """Read-only view of the counter's U-test accumulator (reference ordertest.py:49-104)."""

    def __init__(self, counter):
        self._counter = counter

    def __len__(self):
        return int(self._counter._scalars()[7])

    @property
    def N(self):
        return len(self)

    @property
    def U(self):
        return float(self._counter._scalars()[8])

    @property
    def zscore(self):
        N = len(self)
        if N == 0:
            return 0.0
        return (self.U - N * 0.5) / (N / 12.0)**0.5


class MultiCounter(object):
    """Bootstrap-capable evidence integrator (reference netiter.py:571-855).

    Attributes as in the reference: ``logZ``, ``logZerr``, ``logVolremaining``, ``logZremain``,
    ``logZremainMax``, ``remainder_ratio``, ``remainder_fraction``, ``all_H``, ``all_logZ``,
    ``all_logVolremaining``, ``all_logZremain``, ``logweights`` / ``istail`` (one entry per
    iteration), ``rootids``, ``ncounters``, ``logZ_bs``, ``logZerr_bs``,
    ``insertion_order_runs`` / ``_runlength`` / ``_converged``."""

    def __init__(self, nroots, nbootstraps=10, random=False, check_insertion_order=False):
        everything = np.ones(nroots, dtype=bool)
        rows = [everything]
        for _ in range(nbootstraps):       # same np.random consumption as the reference (:611-619)
            mask = ~everything
            mask[np.unique(np.random.randint(nroots, size=nroots))] = True
            rows.append(mask)
        self.rootids = np.array(rows)
        self.random = random
        self.ncounters = len(self.rootids)
        self.check_insertion_order = check_insertion_order
        self.insertion_order_threshold = 4
        handle = ctypes.c_void_p()
        member = np.ascontiguousarray(self.rootids, dtype=np.uint8)
        check(_lib.lib().mlf_counter_create(ctypes.byref(handle), nroots, self.ncounters, ptr(member), int(bool(random)),
                                            int(bool(check_insertion_order))))
        self._h = handle
        self._passing_node = _lib.lib().mlf_counter_passing_node
        self._kids = np.zeros(8)
        self.insertion_order_accumulator = _OrderAccumulatorView(self)
        self.reset(self.ncounters)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().mlf_counter_destroy(h)
            except Exception:
                pass

    def reset(self, nentries):
        assert nentries == self.ncounters
        self.logweights = []
        self.istail = []
        self.Lmax = -np.inf
        check(_lib.lib().mlf_counter_reset(self._h))
        self._cache = None

    # ---- state held by the compiled counter ---------------------------------------------------
    def _state(self):
        if self._cache is None:
            n = self.ncounters
            scalars = np.empty(10)
            arrays = [np.empty(n) for _ in range(4)]
            nruns = ctypes.c_size_t(0)
            cap = 64
            while True:
                runs = np.empty(cap, dtype=np.int64)
                check(_lib.lib().mlf_counter_state(self._h, ptr(scalars), ptr(arrays[0]), ptr(arrays[1]), ptr(arrays[2]),
                                                   ptr(arrays[3]), ptr(runs), cap, ctypes.byref(nruns)))
                if nruns.value <= cap:
                    break
                cap = nruns.value
            self._cache = (scalars, arrays, [int(r) for r in runs[:nruns.value]])
        return self._cache

    def _scalars(self):
        return self._state()[0]

    logZ = property(lambda self: float(self._scalars()[0]))
    logZerr = property(lambda self: float(self._scalars()[1]))
    logVolremaining = property(lambda self: float(self._scalars()[2]))
    logZremainMax = property(lambda self: float(self._scalars()[3]))
    logZremain = property(lambda self: float(self._scalars()[4]))
    remainder_ratio = property(lambda self: float(self._scalars()[5]))
    remainder_fraction = property(lambda self: float(self._scalars()[6]))
    all_H = property(lambda self: self._state()[1][0])
    all_logZ = property(lambda self: self._state()[1][1])
    all_logVolremaining = property(lambda self: self._state()[1][2])
    all_logZremain = property(lambda self: self._state()[1][3])
    insertion_order_runs = property(lambda self: self._state()[2])

    @property
    def logZ_bs(self):
        """logZ of the bootstrap ensemble"""
        return self.all_logZ[1:].mean()

    @property
    def logZerr_bs(self):
        """logZ scatter of the bootstrap ensemble"""
        return self.all_logZ[1:].std()

    @property
    def insertion_order_runlength(self):
        """Shortest run of the insertion-order test so far (infinity if it never triggered)."""
        runs = self.insertion_order_runs
        return np.inf if len(runs) == 0 else min(runs)

    @property
    def insertion_order_converged(self):
        """Whether the number of U-test resets is compatible with an unbiased run (4 sigma runs
        are expected to last 10^5.5 iterations)."""
        expected_number = max(1, int(np.ceil(len(self.logweights) / 10**(5.5))))
        return len(self.insertion_order_runs) <= expected_number

    def passing_node(self, rootid, node, rootids, parallel_values):
        """Accumulate `node` (from root `rootid`), crossed in parallel by the active nodes with root
        ids `rootids` and values `parallel_values` (reference :721-855)."""
        assert not isinstance(rootid, float)
        children = node.children
        nchildren = len(children)
        if not (isinstance(rootids, np.ndarray) and rootids.dtype == np.int64 and rootids.flags.c_contiguous):
            rootids = np.ascontiguousarray(rootids, dtype=np.int64)
        values = parallel_values
        if not (isinstance(values, np.ndarray) and values.dtype == np.float64 and values.flags.c_contiguous):
            values = np.ascontiguousarray(values, dtype=np.float64)
        if nchildren <= len(self._kids):
            kids = self._kids
            for k in range(nchildren):
                kids[k] = children[k].value
        else:
            kids = np.array([c.value for c in children], dtype=np.float64)
        beta_ptr = None
        if self.random and nchildren >= 1:
            nlive = self.rootids[:, rootids].sum(axis=1)
            beta = np.random.beta(1, nlive, size=self.ncounters)
            beta_ptr = beta.ctypes.data
        logwidth = np.empty(self.ncounters)
        rc = self._passing_node(self._h, int(rootid), float(node.value), nchildren, kids.ctypes.data, rootids.ctypes.data,
                                values.ctypes.data, len(values), beta_ptr, logwidth.ctypes.data)
        if rc:
            check(rc)
        self._cache = None
        self.logweights.append(logwidth)
        self.istail.append(nchildren == 0)


# ---- tree statistics, point pile and run summaries (what the result files hold) ----

def _walk(roots):
    """Yield (explorer, rootid, node, active_rootids) in the driver's breadth-first order; the consumer
    decides whether to expand, drop or stop."""
    explorer = BreadthFirstIterator(roots)
    while True:
        nxt = explorer.next_node()
        if nxt is None:
            return
        rootid, node, (_, active_rootids, _, _) = nxt
        yield explorer, rootid, node, active_rootids


def count_tree(roots):
    """(number of nodes, largest number of parallel edges) of a tree (reference netiter.py:259-285)."""
    nnodes = maxwidth = 0
    for explorer, rootid, node, active_rootids in _walk(roots):
        nnodes += 1
        maxwidth = max(maxwidth, len(active_rootids))
        explorer.expand_children_of(rootid, node)
    return nnodes, maxwidth


def count_tree_between(roots, lo, hi):
    """As :func:`count_tree`, restricted to nodes with lo <= value <= hi (reference :288-330)."""
    nnodes = maxwidth = 0
    for explorer, rootid, node, active_rootids in _walk(roots):
        if node.value > hi:
            break
        if node.value >= lo:
            nnodes += 1
            maxwidth = max(maxwidth, len(active_rootids))
        explorer.expand_children_of(rootid, node)
    return nnodes, maxwidth


def find_nodes_before(root, value):
    """Nodes that have a child at or above `value`, and for each the product of the fork counts on its
    path from the root's children (reference :333-383).  If a root child itself is at or above `value`
    the answer ends with `root` (weight 1)."""
    parents, parent_weights = [], []
    nforks = dict((n.id, 1.) for n in root.children)
    for explorer, rootid, node, _ in _walk(root.children):
        mine = nforks.pop(node.id)
        if node.value >= value:
            parents.append(root)
            parent_weights.append(1)
            break
        if any(child.value >= value for child in node.children):
            parents.append(node)
            parent_weights.append(mine)
            explorer.drop_next_node()
            continue
        explorer.expand_children_of(rootid, node)
        for child in node.children:
            nforks[child.id] = mine * len(node.children)
    return parents, parent_weights


class PointPile(object):
    """Growing table of the coordinates of every tree node: row ``node.id`` holds the unit-cube point and
    the transformed point (reference netiter.py:386-465; same attribute names ``us``, ``ps``, ``nrows``)."""

    def __init__(self, udim, pdim, chunksize=1000):
        self.nrows = 0
        self.chunksize = chunksize
        self.udim = udim
        self.pdim = pdim
        self.us = np.zeros((chunksize, udim))
        self.ps = np.zeros((chunksize, pdim))

    def add(self, newpointu, newpointp):
        """Append one point; returns its row index."""
        assert len(newpointu) == self.udim, (newpointu, self.us.shape)
        assert len(newpointp) == self.pdim, (newpointp, self.ps.shape)
        if self.nrows == len(self.us):
            self.us = np.concatenate((self.us, np.zeros((self.chunksize, self.udim))))
            self.ps = np.concatenate((self.ps, np.zeros((self.chunksize, self.pdim))))
        row = self.nrows
        self.us[row] = newpointu
        self.ps[row] = newpointp
        self.nrows = row + 1
        return row

    def getu(self, i):
        return self.us[i]

    def getp(self, i):
        return self.ps[i]

    def make_node(self, value, u, p):
        """Store the point and return the tree node that refers to it."""
        return TreeNode(value=value, id=self.add(u, p))


def combine_results(saved_logl, saved_nodeids, pointpile, main_iterator, mpi_comm=None):
    """Summary dictionary of a finished exploration (reference netiter.py:858-972): evidence with its
    bootstrap and tail uncertainties, effective sample size, information, weighted and equally weighted
    posterior samples, posterior summaries, best fit.  Keys and value types follow the reference, which is
    what ``info/results.json`` and the ``chains/`` files are written from.  `mpi_comm` is accepted for
    signature compatibility; this build shards over torch.distributed, not MPI (must be None)."""
    assert mpi_comm is None, "MPI exchange is not part of this build"
    from .utils import resample_equal
    saved_logl = np.array(saved_logl)
    logwt = np.array(main_iterator.logweights)
    all_logZ = np.asarray(main_iterator.all_logZ)
    assert logwt.shape == (len(saved_logl), len(all_logZ)), (logwt.shape, saved_logl.shape, all_logZ.shape)
    saved_u = pointpile.getu(saved_nodeids)
    saved_v = pointpile.getp(saved_nodeids)
    logwt0, logwt_bs = logwt[:, 0], logwt[:, 1:]
    logZ_bs = all_logZ[1:]
    logZ = main_iterator.logZ

    wt_bs = np.exp(logwt_bs + saved_logl.reshape((-1, 1)) - logZ_bs)
    wt0 = np.exp(logwt0 + saved_logl - all_logZ[0])
    w = wt0 / wt0.sum()
    assert np.isclose(w.sum() - 1, 0), w.sum()
    n = len(w)
    ess = n / (1.0 + ((n * w - 1)**2).sum() / n)
    tail_fraction = w[np.asarray(main_iterator.istail)].sum()
    logzerr_tail = 0
    if tail_fraction != 0:
        logzerr_tail = np.logaddexp(np.log(tail_fraction) + logZ, logZ) - logZ
    logzerr_bs = (logZ_bs - logZ).max()
    samples = resample_equal(saved_v, w)

    # per-axis information gain from a 39-bin weighted histogram of the cube coordinates
    edges = np.linspace(0, 1, 40)
    information_gain_bits = []
    for column in saved_u.T:
        density, _ = np.histogram(column, weights=wt0, density=True, bins=edges)
        information_gain_bits.append(float((np.log2(1 / ((density + 0.001) * 40)) / 40).sum()))

    best = saved_logl.argmax()
    all_H = np.asarray(main_iterator.all_H)
    results = dict(
        niter=len(saved_logl),
        logz=logZ, logzerr=(logzerr_tail**2 + logzerr_bs**2)**0.5,
        logz_bs=logZ_bs.mean(),
        logz_single=logZ,
        logzerr_tail=logzerr_tail,
        logzerr_bs=logzerr_bs,
        ess=ess,
        H=all_H[0], Herr=all_H.std(),
        posterior=dict(
            mean=samples.mean(axis=0).tolist(),
            stdev=samples.std(axis=0).tolist(),
            median=np.percentile(samples, 50, axis=0).tolist(),
            errlo=np.percentile(samples, 15.8655, axis=0).tolist(),
            errup=np.percentile(samples, 84.1345, axis=0).tolist(),
            information_gain_bits=information_gain_bits,
        ),
        weighted_samples=dict(
            upoints=saved_u, points=saved_v, weights=wt0, logw=logwt0,
            bootstrapped_weights=wt_bs, logl=saved_logl),
        samples=samples,
        maximum_likelihood=dict(
            logl=saved_logl[best],
            point=saved_v[best, :].tolist(),
            point_untransformed=saved_u[best, :].tolist(),
        ),
    )
    if getattr(main_iterator, 'check_insertion_order', False):
        results['insertion_order_MWW_test'] = dict(
            independent_iterations=main_iterator.insertion_order_runlength,
            converged=main_iterator.insertion_order_converged,
        )
    return results


def logz_sequence(root, pointpile, nbootstraps=12, random=True, onNode=None, verbose=False,
                  check_insertion_order=True):
    """Replay the whole tree under `root` through a fresh :class:`MultiCounter` and record, per
    iteration, evidence, its bootstrap scatter, remaining volume, live count and the insertion rank
    of the replacement (reference netiter.py:975-1095).  Returns ``(sequence, results)`` like the
    reference (its docstring states the opposite order)."""
    import sys
    roots = root.children
    explorer = BreadthFirstIterator(roots)
    counter = MultiCounter(nroots=len(roots), nbootstraps=max(1, nbootstraps), random=random,
                           check_insertion_order=check_insertion_order)
    counter.Lmax = max(-np.inf, max(n.value for n in roots))
    logz, logzerr, nlive, logvol, insert_order = [], [], [], [], []
    saved_nodeids, saved_logl = [], []
    while True:
        nxt = explorer.next_node()
        if nxt is None:
            break
        rootid, node, (_, active_rootids, active_values, _) = nxt
        if onNode:
            onNode(node, counter)
        logz.append(counter.logZ)
        with np.errstate(invalid='ignore'):
            logzerr.append(counter.logZerr_bs)
        nactive = len(active_values)
        # rank of the first child among the live values, only defined without ties
        if node.children and len(np.unique(active_values)) == nactive:
            rank = (active_values > node.children[0].value).sum()
            insert_order.append(2 * (rank + 1.) / nactive)
        else:
            insert_order.append(np.nan)
        nlive.append(nactive)
        logvol.append(counter.logVolremaining)
        if verbose:
            sys.stderr.write("%d...\r" % (len(saved_logl) + 1))
        saved_logl.append(node.value)
        saved_nodeids.append(node.id)
        counter.passing_node(rootid, node, active_rootids, active_values)
        explorer.expand_children_of(rootid, node)

    logwt = np.asarray(saved_logl) + np.asarray(counter.logweights)[:, 0]
    logvol[-1] = logvol[-2]
    results = combine_results(saved_logl, saved_nodeids, pointpile, counter)
    sequence = dict(
        logz=np.asarray(logz),
        logzerr=np.asarray(logzerr),
        logvol=np.asarray(logvol),
        samples_n=np.asarray(nlive),
        nlive=np.asarray(nlive),
        insert_order=np.asarray(insert_order),
        logwt=logwt,
        niter=len(saved_logl),
        logl=saved_logl,
        weights=results['weighted_samples']['weights'],
        samples=results['weighted_samples']['points'],
    )
    return sequence, results

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
    """One point of the nested-sampling tree: `value` orders the nodes (log-likelihood), `id` is the row of the
    point in the :class:`PointPile`, `children` the points that replaced it (reference netiter.py:34-60)."""

    __slots__ = ("value", "id", "children")

    def __init__(self, value=None, id=None, children=None):
        self.value, self.id = value, id
        self.children = [] if not children else children

    def __lt__(self, other):
        return self.value < other.value

    def __str__(self, indent=0):
        lines = ["%s- Node: %s" % (" " * indent, self.value)]
        lines += [child.__str__(indent=indent + 2) for child in self.children]
        return "\n".join(lines) if len(lines) > 1 else lines[0] + "\n"


class BreadthFirstIterator(object):
    """Walks the tree in ascending order of node value.  The nodes the walk is currently "passing in parallel" are the
    live points (reference netiter.py:63-162).  The live set is kept as four parallel columns -- node objects, root
    index, value, point id -- in preallocated buffers with a fill count (the driver advances this structure once per
    iteration; growing numpy arrays by concatenation, as a literal reading of the reference would, costs a
    reallocation each time a node forks).  The public attributes are views of the filled part and keep the reference's
    names: ``active_nodes``, ``active_root_ids``, ``active_node_values``, ``active_node_ids``; ordering rules that are
    observable downstream are kept: an only child inherits its parent's slot, the children of a fork go to the end."""

    def __init__(self, roots):
        self.roots = roots
        self.reset()

    def reset(self):
        n = len(self.roots)
        self._cap = max(16, 2 * n)
        self._n = n
        self._nodes = list(self.roots)
        self._root = np.empty(self._cap, dtype=np.int64)
        self._val = np.empty(self._cap, dtype=np.float64)
        self._ids = np.empty(self._cap, dtype=np.int64)
        self._root[:n] = np.arange(n)
        if n:
            self._val[:n] = [node.value for node in self.roots]
            self._ids[:n] = [node.id for node in self.roots]
        self.next_index = None

    active_nodes = property(lambda self: self._nodes)
    active_root_ids = property(lambda self: self._root[:self._n])
    active_node_values = property(lambda self: self._val[:self._n])
    active_node_ids = property(lambda self: self._ids[:self._n])

    def next_node(self):
        """``(rootid, node, (active_nodes, active_root_ids, active_node_values, active_node_ids))`` of the live node
        with the lowest value -- it stays in the live set until dropped or expanded -- or None when nothing is left."""
        step = self._next_index()
        if step is None:
            return None
        i = self.next_index
        # the caller gets arrays of its own: the buffers behind the public attributes are shifted in place when a slot
        # closes, and a caller that keeps the tuple across expand_children_of / drop_next_node (the reference hands out
        # fresh arrays on a fork or drop, netiter.py:110-161) must not see that
        return self._root[i], self._nodes[i], (list(self._nodes), self.active_root_ids.copy(),
                                               self.active_node_values.copy(), self.active_node_ids.copy())

    def _next_index(self):
        """``(rootid, node)`` of the lowest live node, nothing copied: for the walks of this module that look at the live set
        only until their next expand / drop (the public attributes are views of the buffers and valid exactly that long)."""
        if self._n == 0:
            return None
        i = self.next_index = int(np.argmin(self._val[:self._n]))
        return self._root[i], self._nodes[i]

    def _close_gap(self, i):
        n = self._n
        del self._nodes[i]
        for col in (self._root, self._val, self._ids):
            col[i:n - 1] = col[i + 1:n]
        self._n = n - 1

    def _grow_for(self, extra):
        need = self._n + extra
        if need <= self._cap:
            return
        self._cap = max(need, 2 * self._cap)
        for name in ("_root", "_val", "_ids"):
            old = getattr(self, name)
            new = np.empty(self._cap, dtype=old.dtype)
            new[:self._n] = old[:self._n]
            setattr(self, name, new)

    def drop_next_node(self):
        """Take the current node out of the live set without following its children (reference :110-120)."""
        self._close_gap(self.next_index)

    def expand_children_of(self, rootid, node):
        """The current node dies: one child -> it takes over the slot; a fork -> the slot closes and the children
        join at the end; a leaf -> the slot just closes (reference :122-161)."""
        kids = node.children
        i = self.next_index
        if len(kids) == 1:
            only = kids[0]
            self._nodes[i] = only
            self._root[i], self._val[i], self._ids[i] = rootid, only.value, only.id
            return
        self._close_gap(i)
        if kids:
            self._grow_for(len(kids))
            lo, hi = self._n, self._n + len(kids)
            self._nodes.extend(kids)
            self._root[lo:hi] = rootid
            self._val[lo:hi] = [k.value for k in kids]
            self._ids[lo:hi] = [k.id for k in kids]
            self._n = hi


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
        self.random, self.check_insertion_order = random, check_insertion_order
        self.ncounters, self.insertion_order_threshold = len(self.rootids), 4
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
        self.logweights, self.istail, self.Lmax = [], [], -np.inf
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

    # mean and scatter of ln Z over the bootstrap counters (everything but counter 0)
    logZ_bs = property(lambda self: np.mean(self.all_logZ[1:]))
    logZerr_bs = property(lambda self: np.std(self.all_logZ[1:]))

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
    walk = BreadthFirstIterator(roots)
    step = walk._next_index()
    while step is not None:
        yield walk, step[0], step[1], walk.active_root_ids     # a view: read before the consumer expands or drops
        step = walk._next_index()


def _tree_extent(roots, lo=-np.inf, hi=np.inf):
    """nodes with lo <= value <= hi, and the widest live set seen while one of them was the lowest node"""
    size = width = 0
    for walk, rootid, node, live_roots in _walk(roots):
        if node.value > hi:
            break
        if node.value >= lo:
            size, width = size + 1, max(width, len(live_roots))
        walk.expand_children_of(rootid, node)
    return size, width


def count_tree(roots):
    """(number of nodes, largest number of parallel edges) of a tree (reference netiter.py:259-285)."""
    return _tree_extent(roots)


def count_tree_between(roots, lo, hi):
    """As :func:`count_tree`, restricted to nodes with lo <= value <= hi (reference :288-330)."""
    return _tree_extent(roots, lo, hi)


def find_nodes_before(root, value):
    """The places of the tree where the likelihood threshold `value` is crossed (reference netiter.py:333-383):
    every node below `value` that has a child at or above it, each with the number of siblings-at-every-fork along
    its path (product of the fork widths from the root's children down); the subtree of such a node is not looked at
    any further.  If one of the root's children is itself at or above `value`, `root` closes the list with weight 1.
    The answer is in the order the breadth-first walk meets the nodes: ascending node value, equal values (likelihood
    plateaus) in the order of their live-set slots."""
    parents, parent_weights = [], []
    weight_of = {child.id: 1. for child in root.children}
    walk = BreadthFirstIterator(root.children)     # the live-set walk itself: equal values are met in ITS slot order
    while True:
        step = walk._next_index()
        if step is None:
            break
        rootid, node = step
        if node.value >= value:          # only a child of the root can get here: everything still live lies above too
            parents.append(root)
            parent_weights.append(1)
            break
        kids = node.children
        if any(k.value >= value for k in kids):
            parents.append(node)
            parent_weights.append(weight_of[node.id])
            walk.drop_next_node()
        else:
            for k in kids:
                weight_of[k.id] = weight_of[node.id] * len(kids)
            walk.expand_children_of(rootid, node)
        del weight_of[node.id]
    return parents, parent_weights


def dump_tree(filename, roots, pointpile):
    """Write the tree and its points to an HDF5 file (reference netiter.py:220-256, called by
    ``ReactiveNestedSampler.store_tree``, integrator.py:2995-2999): datasets ``unit_points`` / ``points`` (the first
    ``pointpile.nrows`` rows) and the edge list ``nodes_parent_id`` / ``nodes_child_id`` / ``nodes_child_logl`` in the
    order the breadth-first walk meets the parents; all gzip-compressed with the shuffle filter.  h5py is imported here,
    as in the reference, and is not part of this image (an ImportError without it)."""
    import h5py

    edges = [(node.id, kid.id, kid.value) for node in _expanded(roots) for kid in node.children]
    columns = {
        'unit_points': pointpile.us[:pointpile.nrows, :],
        'points': pointpile.ps[:pointpile.nrows, :],
        'nodes_parent_id': [e[0] for e in edges],
        'nodes_child_id': [e[1] for e in edges],
        'nodes_child_logl': [e[2] for e in edges],
    }
    with h5py.File(filename, 'w') as out:
        for name, data in columns.items():
            out.create_dataset(name, data=data, compression='gzip', shuffle=True)


def _expanded(roots):
    """every node of the tree, in the driver's breadth-first order"""
    for walk, rootid, node, _ in _walk(roots):
        yield node
        walk.expand_children_of(rootid, node)


class PointPile(object):
    """Append-only table of the points behind the tree nodes: row ``node.id`` holds the unit-cube coordinates and the
    transformed parameters (reference netiter.py:386-465; attributes ``us``, ``ps``, ``nrows``, ``udim``, ``pdim``,
    ``chunksize``).  One buffer with both blocks side by side, grown geometrically; ``us`` / ``ps`` are views of it
    (at least `chunksize` rows are allocated from the start, as callers slice them with ``[:nrows]``)."""

    def __init__(self, udim, pdim, chunksize=1000):
        self.udim, self.pdim, self.chunksize = udim, pdim, chunksize
        self.nrows = 0
        self._table = np.zeros((chunksize, udim + pdim))

    us = property(lambda self: self._table[:, :self.udim])
    ps = property(lambda self: self._table[:, self.udim:])

    def add(self, newpointu, newpointp):
        """Store one point, return its row."""
        newpointu, newpointp = np.asarray(newpointu), np.asarray(newpointp)
        if newpointu.shape != (self.udim,) or newpointp.shape != (self.pdim,):
            raise AssertionError((newpointu.shape, newpointp.shape, self.udim, self.pdim))
        row = self.nrows
        if row == len(self._table):
            bigger = np.zeros((max(row + self.chunksize, 2 * row), self.udim + self.pdim))
            bigger[:row] = self._table
            self._table = bigger
        self._table[row, :self.udim] = newpointu
        self._table[row, self.udim:] = newpointp
        self.nrows = row + 1
        return row

    def getu(self, i):
        return self._table[i, :self.udim]

    def getp(self, i):
        return self._table[i, self.udim:]

    def make_node(self, value, u, p):
        """Store the point and hand back the tree node that points at it."""
        return TreeNode(value=value, id=self.add(u, p))


def _kish_ess(w):
    """effective sample size of normalised weights in the reference's form N / (1 + mean((N w - 1)^2))"""
    n = len(w)
    return n / (1.0 + np.mean((n * w - 1) ** 2))


def _axis_information_bits(upoints, weights, nbins=40):
    """Per unit-cube axis: how far the weighted marginal departs from uniform, in bits (reference netiter.py:915-922):
    histogram density on nbins - 1 equal bins of [0, 1], then sum(log2(1 / ((density + 0.001) nbins)) / nbins)."""
    edges = np.linspace(0, 1, nbins)
    bits = []
    for column in upoints.T:
        density = np.histogram(column, bins=edges, weights=weights, density=True)[0]
        bits.append(float(np.sum(np.log2(1.0 / ((density + 0.001) * nbins)) / nbins)))
    return bits


def _posterior_summary(samples):
    lo, mid, hi = np.percentile(samples, [15.8655, 50, 84.1345], axis=0)
    return dict(mean=samples.mean(axis=0).tolist(), stdev=samples.std(axis=0).tolist(), median=mid.tolist(),
                errlo=lo.tolist(), errup=hi.tolist())


def combine_results(saved_logl, saved_nodeids, pointpile, main_iterator, mpi_comm=None):
    """What a finished walk over the tree says (reference netiter.py:858-972): evidence of the main counter, its
    scatter over the bootstrap counters and the share of the still-live tail, information, effective sample size,
    weighted and equally weighted posterior samples with their summaries, the best fit.  The dictionary layout is the
    reference's (``info/results.json`` and the ``chains/`` files are written from it).  Under MPI the reference
    gathers the bootstrap columns of all ranks; this build shards over torch.distributed only, so an MPI communicator
    is refused up front."""
    if mpi_comm is not None:
        raise NotImplementedError("combine_results: MPI gathering is not part of this build (pass mpi_comm=None)")
    from .utils import resample_equal
    logl = np.asarray(saved_logl, dtype=float)
    logwidths = np.asarray(main_iterator.logweights)             # (iterations, counters): column 0 = all roots
    logz_all = np.asarray(main_iterator.all_logZ)
    if logwidths.shape != (len(logl), len(logz_all)):
        raise AssertionError((logwidths.shape, logl.shape, logz_all.shape))
    logz = main_iterator.logZ
    upoints, points = pointpile.getu(saved_nodeids), pointpile.getp(saved_nodeids)

    # posterior weights of every counter at once; the main counter's column, normalised, drives everything below
    weights_all = np.exp(logwidths + logl[:, None] - logz_all[None, :])
    main = weights_all[:, 0]
    w = main / main.sum()
    if not np.isclose(w.sum(), 1):
        raise AssertionError(w.sum())
    tail_share = w[np.asarray(main_iterator.istail, dtype=bool)].sum()
    logzerr_tail = 0 if tail_share == 0 else np.logaddexp(np.log(tail_share) + logz, logz) - logz
    logz_boot = logz_all[1:]
    logzerr_bs = (logz_boot - logz).max()
    samples = resample_equal(points, w)
    posterior = _posterior_summary(samples)
    posterior["information_gain_bits"] = _axis_information_bits(upoints, main)
    info_all = np.asarray(main_iterator.all_H)
    top = int(np.argmax(logl))
    results = dict(
        niter=len(logl),
        logz=logz, logzerr=float((logzerr_tail**2 + logzerr_bs**2)**0.5),   # the reference's expression (netiter.py:927): hypot may differ in the last place
        logz_bs=logz_boot.mean(), logz_single=logz,
        logzerr_tail=logzerr_tail, logzerr_bs=logzerr_bs,
        ess=_kish_ess(w),
        H=info_all[0], Herr=info_all.std(),
        posterior=posterior,
        weighted_samples=dict(upoints=upoints, points=points, weights=main, logw=logwidths[:, 0],
                              bootstrapped_weights=weights_all[:, 1:], logl=logl),
        samples=samples,
        maximum_likelihood=dict(logl=logl[top], point=points[top, :].tolist(),
                                point_untransformed=upoints[top, :].tolist()),
    )
    if getattr(main_iterator, 'check_insertion_order', False):
        results['insertion_order_MWW_test'] = dict(independent_iterations=main_iterator.insertion_order_runlength,
                                                   converged=main_iterator.insertion_order_converged)
    return results


def logz_sequence(root, pointpile, nbootstraps=12, random=True, onNode=None, verbose=False,
                  check_insertion_order=True):
    """Replay the tree below `root` through a fresh :class:`MultiCounter` and keep, for every iteration, what the
    run looked like just before the node died: evidence and its bootstrap scatter, remaining volume, number of live
    points, and where the replacement landed among the live likelihoods (reference netiter.py:975-1095).  Returns
    ``(sequence, results)``, the order the reference implements (its docstring says the opposite)."""
    import sys
    roots = root.children
    walk = BreadthFirstIterator(roots)
    counter = MultiCounter(nroots=len(roots), nbootstraps=max(1, nbootstraps), random=random,
                           check_insertion_order=check_insertion_order)
    counter.Lmax = max([-np.inf] + [n.value for n in roots])
    track = dict(logz=[], logzerr=[], logvol=[], nlive=[], insert_order=[], logl=[], nodeid=[])
    step = walk.next_node()
    while step is not None:
        rootid, node, (_, live_roots, live_values, _) = step
        if onNode:
            onNode(node, counter)
        nlive = len(live_values)
        with np.errstate(invalid='ignore'):
            scatter = counter.logZerr_bs
        # insertion rank of the first replacement among the live values; undefined when values tie
        order = np.nan
        if node.children and len(np.unique(live_values)) == nlive:
            order = 2 * (np.count_nonzero(live_values > node.children[0].value) + 1.) / nlive
        for key, item in (("logz", counter.logZ), ("logzerr", scatter), ("logvol", counter.logVolremaining),
                          ("nlive", nlive), ("insert_order", order), ("logl", node.value), ("nodeid", node.id)):
            track[key].append(item)
        if verbose:
            sys.stderr.write("%d...\r" % len(track["logl"]))
        counter.passing_node(rootid, node, live_roots, live_values)
        walk.expand_children_of(rootid, node)
        step = walk.next_node()

    track["logvol"][-1] = track["logvol"][-2]
    results = combine_results(track["logl"], track["nodeid"], pointpile, counter)
    nlive = np.asarray(track["nlive"])
    sequence = dict(
        logz=np.asarray(track["logz"]), logzerr=np.asarray(track["logzerr"]), logvol=np.asarray(track["logvol"]),
        samples_n=nlive, nlive=nlive, insert_order=np.asarray(track["insert_order"]),
        logwt=np.asarray(track["logl"]) + np.asarray(counter.logweights)[:, 0],
        niter=len(track["logl"]), logl=track["logl"],
        weights=results['weighted_samples']['weights'], samples=results['weighted_samples']['points'],
    )
    return sequence, results

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

"""Integrator bookkeeping: ultranest_amd.netiter (compiled MultiCounter,
BreadthFirstIterator) against traces recorded from the reference's netiter over seeded trees
(tests/golden/make_golden.py g10).  Host code only -- these tests need no GPU.  Integers (order of
exploration, live counts, tail flags, U-test run lengths, bootstrap membership) are exact; the
floating-point sums go through libm instead of numpy's vector exp/log: 1e-12."""
import numpy as np
import pytest

from golden import inputs



@pytest.fixture(params=["cpu", pytest.param("gpu-box", marks=pytest.mark.gpu)], autouse=True)
def where(request):
    """The bookkeeping is host code (compiled into libmlfriends_hip.so), but harness.StaticNestedSampler uses it on the GPU box:
    every test of this module runs in BOTH suites (`-m "not gpu"` here, `-m gpu` on the MI355X box), VERDICT r5."""
    return request.param


CASES = [(1001, 40, 900, 10, False), (1002, 400, 4000, 30, False), (1003, 25, 600, 5, True), (1004, 1, 50, 3, False)]


def explore(seed, nroots, nnodes, nbootstraps, random):
    import ultranest_amd.netiter as netiter
    values, children = inputs.random_tree(seed, nroots, nnodes)
    roots = inputs.build_nodes(netiter.TreeNode, values, children, nroots)
    np.random.seed(seed)
    explorer = netiter.BreadthFirstIterator(roots)
    counter = netiter.MultiCounter(nroots=nroots, nbootstraps=nbootstraps, random=random, check_insertion_order=True)
    trace = []
    while True:
        nxt = explorer.next_node()
        if nxt is None:
            break
        rootid, node, (_, active_rootids, active_values, active_node_ids) = nxt
        counter.passing_node(rootid, node, active_rootids, active_values)
        trace.append([node.id, rootid, len(active_values), counter.logZ, counter.logZerr, counter.logVolremaining,
                      counter.logZremain, counter.logZremainMax, counter.remainder_ratio, counter.remainder_fraction,
                      len(counter.insertion_order_accumulator), counter.insertion_order_accumulator.zscore])
        explorer.expand_children_of(rootid, node)
    return counter, np.array(trace, dtype=float)


@pytest.mark.parametrize("seed,nroots,nnodes,nboot,random", CASES)
def test_counter_trace_equals_reference(golden, seed, nroots, nnodes, nboot, random):
    g = golden("g10_netiter")
    with np.errstate(all="ignore"):
        counter, trace = explore(seed, nroots, nnodes, nboot, random)
    k = "c%d_" % seed
    step = 16 if len(trace) > 1000 else 1
    want = g[k + "trace"]
    got = trace[::step]
    assert got.shape == want.shape
    assert np.array_equal(got[:, :3], want[:, :3])                    # exploration order, root ids, live counts
    assert np.array_equal(got[:, 10], want[:, 10])                    # U-test accumulator length
    np.testing.assert_allclose(got[:, 3:10], want[:, 3:10], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got[:, 11], want[:, 11], rtol=1e-10, atol=1e-12)
    assert np.array_equal(counter.rootids, g[k + "rootids"])
    assert np.random.uniform() == float(g[k + "next_random"])          # same numpy stream consumption
    assert np.array_equal(np.array(counter.istail), g[k + "istail"])
    assert np.array_equal(np.array(counter.insertion_order_runs, dtype=np.int64), g[k + "runs"])
    lw = np.array(counter.logweights)[::step]
    assert np.array_equal(np.isfinite(lw), np.isfinite(g[k + "logweights"]))
    fin = np.isfinite(lw)
    np.testing.assert_allclose(lw[fin], g[k + "logweights"][fin], rtol=1e-12, atol=1e-12)
    for name in ("all_logZ", "all_H", "all_logVolremaining"):
        np.testing.assert_allclose(getattr(counter, name), g[k + name], rtol=1e-11, atol=1e-12, equal_nan=True)
    assert abs(counter.logZ_bs - float(g[k + "logZ_bs"])) < 1e-11
    assert abs(counter.logZerr_bs - float(g[k + "logZerr_bs"])) < 1e-11
    assert counter.insertion_order_converged in (True, False)
    assert counter.insertion_order_runlength == (min(g[k + "runs"]) if len(g[k + "runs"]) else np.inf)


def test_counter_reset_and_errors():
    import ultranest_amd.netiter as netiter
    np.random.seed(1)
    c = netiter.MultiCounter(nroots=4, nbootstraps=2)
    leaf = netiter.TreeNode(value=-1.0, id=0)
    c.passing_node(0, leaf, np.arange(4), np.array([-1.0, 0.0, 0.5, 1.0]))
    assert np.isfinite(c.logZ) and c.istail == [True] and len(c.logweights) == 1
    c.reset(c.ncounters)
    assert c.logZ == -np.inf and c.logZerr == np.inf and c.logweights == [] and c.remainder_ratio == 1.0
    with pytest.raises(ValueError):
        c.passing_node(7, leaf, np.arange(4), np.zeros(4))            # unknown root


def test_breadth_first_iterator_drop():
    import ultranest_amd.netiter as netiter
    roots = [netiter.TreeNode(value=v, id=i) for i, v in enumerate([3.0, 1.0, 2.0])]
    it = netiter.BreadthFirstIterator(roots)
    rootid, node, (nodes, rids, vals, ids) = it.next_node()
    assert rootid == 1 and node.id == 1 and len(nodes) == 3
    it.drop_next_node()
    assert [n.id for n in it.active_nodes] == [0, 2] and list(it.active_root_ids) == [0, 2]
    roots[2].children = [netiter.TreeNode(value=5.0, id=3), netiter.TreeNode(value=6.0, id=4)]
    rootid, node, _ = it.next_node()
    it.expand_children_of(rootid, node)
    assert [n.id for n in it.active_nodes] == [0, 3, 4] and list(it.active_root_ids) == [0, 2, 2]
    assert list(it.active_node_values) == [3.0, 5.0, 6.0]


def test_find_nodes_before_orders_ties_by_live_slot():
    """likelihood plateau: equal-valued nodes are reported in the order of their slots in the breadth-first walk's live
    set (reference netiter.py:356-383 walks with BreadthFirstIterator; np.argmin takes the first of equal values)"""
    from ultranest_amd.netiter import TreeNode, find_nodes_before
    root = TreeNode(id=-1, value=-np.inf)
    kids = []
    for i in range(4):
        low = TreeNode(id=i, value=1.0)              # four roots on one plateau
        low.children.append(TreeNode(id=10 + i, value=5.0))
        root.children.append(low)
        kids.append(low)
    # a fork whose children tie with a later root: children of a fork go to the END of the live set
    forked = TreeNode(id=4, value=0.5)
    for j in range(2):
        c = TreeNode(id=20 + j, value=1.0)
        c.children.append(TreeNode(id=30 + j, value=7.0))
        forked.children.append(c)
    root.children.append(forked)
    parents, weights = find_nodes_before(root, 3.0)
    assert [p.id for p in parents] == [0, 1, 2, 3, 20, 21]
    assert weights == [1., 1., 1., 1., 2., 2.]

"""ultranest_amd.ordertest with the recipes of the reference's tests/test_ordertest.py, plus agreement with the
U-test state kept inside the compiled MultiCounter.  Host code only."""
import numpy as np
import pytest


# reference tests/test_ordertest.py:6-16
def test_invalid_order_and_growing_live_set():
    from ultranest_amd.ordertest import UniformOrderAccumulator
    acc = UniformOrderAccumulator()
    acc.add(2, 3)
    with pytest.raises(ValueError):
        acc.add(4, 3)
    with pytest.raises(ValueError):
        acc.add(-1, 3)
    acc = UniformOrderAccumulator()
    acc.add(1, 3)
    acc.add(4, 5)
    acc.add(5, 6)
    assert len(acc) == 3 and acc.N == 3


# reference tests/test_ordertest.py:18-46
def test_order_correctness():
    from ultranest_amd.ordertest import UniformOrderAccumulator, infinite_U_zscore
    np.random.seed(1)
    nlive, n = 400, 1000
    nruns = []
    for frac in 1, 0.9:
        acc = UniformOrderAccumulator()
        runlength, samples = [], []
        for i in range(n):
            order = np.random.randint(0, nlive * frac)
            acc.add(order, nlive)
            samples.append(order)
            zscore = acc.zscore
            assert np.isclose(zscore, infinite_U_zscore(np.asarray(samples), nlive)), (zscore, samples)
            if abs(zscore) > 3:
                runlength.append(len(acc))
                acc.reset()
                samples = []
        nruns.append(len(runlength))
    assert nruns[0] == 0 and nruns[1] > 0, nruns
    assert UniformOrderAccumulator().zscore == 0.0


def test_same_statistic_as_the_compiled_counter():
    """The accumulator inside MultiCounter (csrc/mlf_netiter.hip) and this class see the same ranks when the
    counter runs over a chain of single-child nodes."""
    import ultranest_amd.netiter as netiter
    from ultranest_amd.ordertest import UniformOrderAccumulator
    rs = np.random.RandomState(5)
    nroots = 50
    values = list(np.sort(rs.normal(size=nroots)))
    roots = [netiter.TreeNode(value=float(v), id=i) for i, v in enumerate(values)]
    np.random.seed(3)
    explorer = netiter.BreadthFirstIterator(roots)
    counter = netiter.MultiCounter(nroots=nroots, nbootstraps=3, random=False, check_insertion_order=True)
    mine = UniformOrderAccumulator()
    runs = []
    nid = nroots
    for step in range(300):
        rootid, node, (_, active_rootids, active_values, _) = explorer.next_node()
        child = netiter.TreeNode(value=float(node.value + rs.exponential(0.3) + 1e-9), id=nid)
        nid += 1
        node.children.append(child)
        if len(np.unique(active_values)) == len(active_values):
            mine.add(int((active_values < child.value).sum()), len(active_values))
            if abs(mine.zscore) > counter.insertion_order_threshold:      # the counter's rolling reset (netiter.py:838-846)
                runs.append(len(mine))
                mine.reset()
        counter.passing_node(rootid, node, active_rootids, active_values)
        explorer.expand_children_of(rootid, node)
    assert len(counter.insertion_order_accumulator) == len(mine)
    assert abs(counter.insertion_order_accumulator.zscore - mine.zscore) < 1e-9
    assert list(counter.insertion_order_runs) == runs

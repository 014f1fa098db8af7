"""Run summaries and result files: ultranest_amd.netiter.combine_results /
logz_sequence / tree statistics, ultranest_amd.utils and ultranest_amd.results against what the
reference's driver (`ReactiveNestedSampler._update_results`, integrator.py:2933-2995) produced on
seeded trees (tests/golden/make_golden.py g12).  Host code only.

Exact: exploration order, live counts, ranks, file layout, numpy stream consumption, and -- when the
writer is fed the reference's own arrays -- every byte of the six files.  Floating-point summaries
computed through the compiled counter: 1e-10 (libm instead of numpy's vector exp/log)."""
import hashlib
import json
import os
import sys
import types

import numpy as np
import pytest

from golden import inputs

FILES = (("chains", "equal_weighted_post.txt"), ("chains", "weighted_post.txt"),
         ("chains", "weighted_post_untransformed.txt"), ("chains", "run.txt"),
         ("info", "results.json"), ("info", "post_summary.csv"))


def finish_run(seed, nroots, nnodes, nboot, log_dir):
    """What the driver does after the last iteration, with this package's modules."""
    import ultranest_amd.netiter as netiter
    from ultranest_amd import results as R
    from ultranest_amd.utils import make_run_dir
    values, children = inputs.random_tree(seed, nroots, nnodes)
    us, ps = inputs.tree_points(seed, len(values))
    pile = netiter.PointPile(us.shape[1], ps.shape[1], chunksize=37)       # forces growth
    for urow, prow in zip(us, ps):
        pile.add(urow, prow)
    roots = inputs.build_nodes(netiter.TreeNode, values, children, nroots)
    root = netiter.TreeNode(id=-1, value=-np.inf, children=roots)
    np.random.seed(seed)
    explorer = netiter.BreadthFirstIterator(roots)
    counter = netiter.MultiCounter(nroots=nroots, nbootstraps=nboot, random=False, check_insertion_order=False)
    saved_logl, saved_nodeids = [], []
    while True:
        nxt = explorer.next_node()
        if nxt is None:
            break
        rootid, node, (_, active_rootids, active_values, _) = nxt
        saved_logl.append(node.value)
        saved_nodeids.append(node.id)
        counter.passing_node(rootid, node, active_rootids, active_values)
        explorer.expand_children_of(rootid, node)
    names, derived = inputs.RESULT_PARAMNAMES
    res = netiter.combine_results(saved_logl, saved_nodeids, pile, counter)
    sequence, res2 = netiter.logz_sequence(root, pile, random=True, check_insertion_order=True)
    R.finish_results(res, res2, 7 * len(values), names + derived, (counter.all_H[0] / nroots)**0.5)
    logs = make_run_dir(log_dir, run_num=1)
    R.write_results(logs, res, sequence, names + derived)
    return types.SimpleNamespace(results=res, sequence=sequence, logs=logs, root=root), np.random.uniform()


def _close(a, b, tol=1e-10):
    np.testing.assert_allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=tol, atol=tol, equal_nan=True)


def _json_close(got, want, path=""):
    assert type(got) is type(want) or (isinstance(got, (int, float)) and isinstance(want, (int, float))), (path, got, want)
    if isinstance(want, dict):
        assert list(got) == list(want), (path, list(got), list(want))       # same keys, same order
        for k in want:
            _json_close(got[k], want[k], path + "/" + k)
    elif isinstance(want, list):
        assert len(got) == len(want), path
        for i, (g, w) in enumerate(zip(got, want)):
            _json_close(g, w, "%s[%d]" % (path, i))
    elif isinstance(want, float):
        _close(got, want, 1e-9)
    else:
        assert got == want, (path, got, want)


@pytest.mark.parametrize("seed,nroots,nnodes,nboot", inputs.RESULT_TREES)
def test_run_summary_and_files_equal_reference(golden, tmp_path, seed, nroots, nnodes, nboot):
    g = golden("g12_results")
    k = "t%d_" % seed
    with np.errstate(all="ignore"):
        run, next_random = finish_run(seed, nroots, nnodes, nboot, str(tmp_path))
    assert next_random == float(g[k + "next_random"])          # same np.random consumption end to end
    ws = run.results["weighted_samples"]
    assert np.array_equal(ws["upoints"], g[k + "ws_upoints"])   # same dead-point order
    assert np.array_equal(ws["points"], g[k + "ws_points"])
    assert np.array_equal(ws["logl"], g[k + "ws_logl"])
    for name in ("weights", "logw", "bootstrapped_weights"):
        fin = np.isfinite(g[k + "ws_" + name])
        assert np.array_equal(np.isfinite(ws[name]), fin)
        _close(np.asarray(ws[name])[fin], g[k + "ws_" + name][fin])
    assert np.array_equal(run.results["samples"], g[k + "samples"])       # resampled rows are copies: exact
    seq = run.sequence
    assert np.array_equal(seq["nlive"], g[k + "seq_nlive"])
    assert np.array_equal(np.asarray(seq["insert_order"]), g[k + "seq_insert_order"], equal_nan=True)
    assert np.array_equal(np.asarray(seq["logl"]), g[k + "seq_logl"])
    for name in ("logz", "logzerr", "logvol", "logwt", "weights"):
        fin = np.isfinite(g[k + "seq_" + name])
        assert np.array_equal(np.isfinite(seq[name]), fin), name
        _close(np.asarray(seq[name])[fin], g[k + "seq_" + name][fin])
    # results.json: same keys in the same order, same types, numbers to 1e-9
    got = json.load(open(os.path.join(run.logs["info"], "results.json")))
    want = json.loads(str(g[k + "results_json"]))
    _json_close(got, want)
    # every file: same header line and the same number of lines / columns
    for sub, fn in FILES:
        text = open(os.path.join(run.logs[sub], fn)).read()
        head = str(g[k + "head_" + fn])
        assert text.split("\n")[0] == head.split("\n")[0], fn
        if fn.endswith(".txt"):
            first = text.split("\n")[1].split()
            assert len(first) == len(head.split("\n")[1].split()), fn
    assert len(open(os.path.join(run.logs["chains"], "run.txt")).read().splitlines()) == run.results["niter"] + 1


@pytest.mark.parametrize("seed,nroots,nnodes,nboot", inputs.RESULT_TREES)
def test_writer_is_byte_identical_on_reference_arrays(golden, tmp_path, seed, nroots, nnodes, nboot):
    """Fed the arrays the reference wrote from, the writer reproduces its six files byte for byte."""
    from ultranest_amd import results as R
    from ultranest_amd.utils import make_run_dir
    g = golden("g12_results")
    k = "t%d_" % seed
    res = json.loads(str(g[k + "results_json"]))
    res["weighted_samples"] = dict((name, g[k + "ws_" + name]) for name in
                                   ("upoints", "points", "weights", "logw", "bootstrapped_weights", "logl"))
    res["samples"] = g[k + "samples"]
    sequence = dict((name, g[k + "seq_" + name]) for name in
                    ("logz", "logzerr", "logvol", "nlive", "insert_order", "logwt", "logl", "weights"))
    names, derived = inputs.RESULT_PARAMNAMES
    logs = make_run_dir(str(tmp_path), run_num=1)
    R.write_results(logs, res, sequence, names + derived)
    for sub, fn in FILES:
        data = open(os.path.join(logs[sub], fn), "rb").read()
        assert hashlib.sha256(data).hexdigest() == str(g[k + "sha256_" + fn]), fn


@pytest.mark.parametrize("seed,nroots,nnodes,nboot", inputs.RESULT_TREES)
def test_tree_statistics_equal_reference(golden, seed, nroots, nnodes, nboot):
    import ultranest_amd.netiter as netiter
    g = golden("g12_results")
    k = "t%d_" % seed
    values, children = inputs.random_tree(seed, nroots, nnodes)
    roots = inputs.build_nodes(netiter.TreeNode, values, children, nroots)
    root = netiter.TreeNode(id=-1, value=-np.inf, children=roots)
    assert list(netiter.count_tree(roots)) == list(g[k + "count_tree"])
    lo, hi, n, width = g[k + "count_between"]
    assert list(netiter.count_tree_between(roots, lo, hi)) == [n, width]
    for tag in ("mid", "high", "first"):
        want = g[k + "before_" + tag]
        thresh = want[0, 0] if len(want) else float(np.max(values)) + 1.0
        parents, weights = netiter.find_nodes_before(root, thresh)
        assert [p.id for p in parents] == [int(i) for i in want[:, 1]], tag
        assert list(map(float, weights)) == list(want[:, 2]), tag


def test_resample_equal_and_run_dir(golden, tmp_path):
    from ultranest_amd import utils
    g = golden("g12_results")
    out = utils.resample_equal(g["resample_x"], g["resample_w"], rstate=np.random.RandomState(99))
    assert np.array_equal(out, g["resample_out"])
    with pytest.raises(ValueError, match="do not sum to 1"):
        utils.resample_equal(g["resample_x"], g["resample_w"] * 0.9)
    # every row appears floor(w N) or ceil(w N) times
    w = g["resample_w"]
    counts = np.array([(out == row).all(axis=1).sum() for row in g["resample_x"]])
    assert np.all(counts >= np.floor(w * len(w) - 1e-9)) and np.all(counts <= np.ceil(w * len(w) + 1e-9))
    base = str(tmp_path / "logs")
    first = utils.make_run_dir(base)
    second = utils.make_run_dir(base)
    assert first["run_dir"].endswith("run1") and second["run_dir"].endswith("run2")
    assert sorted(os.listdir(first["run_dir"])) == ["chains", "extra", "info", "plots", "results"]
    flat = utils.make_run_dir(str(tmp_path / "flat"), append_run_num=False)
    assert flat["run_dir"] == str(tmp_path / "flat") and os.path.isdir(flat["chains"])


def test_make_run_dir_numbering_limit(tmp_path):
    from ultranest_amd.utils import make_run_dir
    base = str(tmp_path)
    make_run_dir(base, max_run_num=3)
    assert os.path.exists(os.path.join(base, 'run1'))
    make_run_dir(base, max_run_num=3)
    assert os.path.exists(os.path.join(base, 'run2'))
    with pytest.raises(ValueError):
        make_run_dir(base, max_run_num=3)


# ---- dump_tree (reference netiter.py:220-256) ----------------------------------------------------
@pytest.mark.parametrize("seed,nroots,nnodes,nboot", inputs.RESULT_TREES)
def test_dump_tree_hands_h5py_what_the_reference_does(golden, seed, nroots, nnodes, nboot):
    """h5py is not in the image: the reference's dump_tree and ours both write through the recording stand-in of
    tests/golden/inputs.py (g14); same file mode, same datasets in the same order, same arrays bit for bit, same options."""
    import json
    import ultranest_amd.netiter as netiter
    g = golden("g14_tree_dump")
    k = "t%d_" % seed
    w = inputs.recorded_tree_dump(netiter, seed, nroots, nnodes)
    assert w["mode"] == str(g[k + "mode"]) and w["filename"] == "tree_%d.hdf5" % seed
    assert list(w["datasets"]) == list(g[k + "names"])
    for name, (data, options) in w["datasets"].items():
        want = g[k + name]
        assert data.dtype == want.dtype and np.array_equal(data, want), name
        assert json.dumps(options, sort_keys=True) == str(g[k + name + "_options"]), name


def test_dump_tree_without_h5py_raises_importerror(monkeypatch):
    import ultranest_amd.netiter as netiter
    monkeypatch.setitem(sys.modules, "h5py", None)
    with pytest.raises(ImportError):
        netiter.dump_tree("nowhere.hdf5", [], netiter.PointPile(2, 2))
